"""The fused FIR -> biquad -> gain kernel (chain_fused.hip): one pass over float32 buffers,
tile-to-tile biquad state through decoupled look-back.

Contract (the same as the overlap-save FIR and the time-segmented biquad it replaces in a
chain): the float64 value differs from the oracle's ordered arithmetic by O(1e-16) of full
scale, so the float32 result is the oracle's rounded value except at rounding boundaries:
    |gpu - (float)oracle| <= 1 ulp_f32( max(|oracle|, 2^-24 * max|oracle of the Line|) )
and almost every sample is identical.  PIPE_HIP_PARAM_EXACT keeps the staged bit-exact chain."""
import numpy as np
import pytest

from oracle import oracle as O
from pipe_amd import synth

pytestmark = pytest.mark.gpu

P = None
torch = None


def setup_module(module):
    global P, torch
    import torch as _t
    from pipe_amd import processors as _p
    assert _t.cuda.is_available()
    P, torch = _p, _t


def ulps(got, want64):
    floor = 2.0 ** -24 * np.abs(want64).max()
    want32 = want64.astype(np.float32)
    mag = np.maximum(np.abs(want64), floor).astype(np.float32)
    return np.abs(got.astype(np.float64) - want32.astype(np.float64)) / np.spacing(mag).astype(np.float64)


DC_BLOCK = np.array([[1.0, -1.0, 0.0, -0.9995, 0.0]])             # pole at 0.9995: forgets over ~10^4 frames
LOWPASS = synth.biquad_rbj_lowpass()                              # forgets within one 769-frame tile
TWO_SECTIONS = np.vstack([synth.biquad_rbj_lowpass(3000.0), synth.biquad_rbj_lowpass(700.0, q=2.0)])
THREE_SECTIONS = np.vstack([TWO_SECTIONS, synth.biquad_rbj_lowpass(1500.0, q=1.1)])
FOUR_SECTIONS = np.vstack([THREE_SECTIONS, synth.biquad_rbj_lowpass(5000.0, q=0.6)])


def run_chain(taps, q, g, x, calls, exact=False):
    """x: (lines, frames, C) float32; calls: frame counts of consecutive process_batch calls."""
    lines, frames, C = x.shape
    kw = dict(dtype=np.float32, lines=lines, max_batch=1)
    fir = P.Fir(taps, frames, C, **kw)
    stages = [fir, P.Biquad(q, frames, C, **kw)]
    if g is not None:
        stages.append(P.Gain(g, frames, C, **kw))
    with P.Chain(stages) as p:
        if exact:
            p._set_param(3, [1.0])
        p.start()
        d_in = torch.from_numpy(x).cuda()
        outs, names, pos = [], [], 0
        for n in calls:
            xin = d_in[:, pos:pos + n, :].contiguous()
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            names.append(p.kernel_name())
            outs.append(y)
            pos += n
        p.flush()   # reports a look-back that gave up
        return torch.cat(outs, dim=1).cpu().numpy(), names


def oracle_chain(taps, q, g, x):
    C = x.shape[-1]
    y = O.Biquad(q, C).process(O.Fir(taps, C).process(x.astype(np.float64)))
    return (O.gain(y, g) if g is not None else y).reshape(-1, C)


@pytest.mark.parametrize("lines,C,frames,ntaps,q,g,calls", [
    (24, 8, 4096, 256, LOWPASS, 0.7071067811865476, [4096]),           # BASELINE configs[3] shape, fewer Lines
    (1, 2, 769 * 300 + 17, 256, LOWPASS, 0.5, [769 * 100, 769 * 200 + 17]),   # one long series, two calls
    (1, 2, 769 * 90, 256, DC_BLOCK, None, [769 * 90]),                 # slow filter: look-back goes to a P record
    (3, 4, 50_000, 100, DC_BLOCK, 1.25, [20_000, 30_000]),             # several windows of 32 predecessors, ragged
    (5, 6, 30_000, 64, TWO_SECTIONS, 0.9, [9_999, 20_001]),            # 2 sections, 3 pairs (units straddle tiles)
    (2, 2, 40_000, 511, TWO_SECTIONS, None, [40_000]),                 # longest filter: tiles of 514 frames; two sections fused
    (9, 4, 3 * 4096, 256, TWO_SECTIONS, 0.7, [4096, 4096, 4096]),      # two sections, three launches: both sections' slots carry
    (3, 2, 20_032, 100, TWO_SECTIONS, 1.5, [10_016, 4000, 6016]),      # ... a ragged end in the middle (tail kernel, two sections)
    (7, 16, 5_000, 16, LOWPASS, 2.0, [5_000]),                         # shortest filter, first output in lane 0
    (9, 4, 3 * 4096, 256, THREE_SECTIONS, 0.7, [4096, 4096, 4096]),    # three sections (round 6: the global look-back), three launches
    (3, 2, 20_032, 100, FOUR_SECTIONS, 1.5, [10_016, 4000, 6016]),     # four sections, a ragged end in the middle (tail kernel, four sections)
    (40, 8, 4096, 256, FOUR_SECTIONS, None, [4096]),                   # four sections at the configs[3] shape in small
    # odd channel counts (round 6): the last channel alone in its pair, its other half a channel that does not exist
    (5, 3, 20_000, 100, LOWPASS, 0.9, [9_999, 10_001]),                # three channels, ragged calls (tail kernel: three series a Line)
    (8, 1, 30_000, 256, LOWPASS, None, [12_000, 18_000]),              # mono Lines: every frame its own "pair"
    (3, 5, 3 * 4096, 64, TWO_SECTIONS, 1.1, [4096, 4096, 4096]),       # five channels, two sections, three launches
    (256, 3, 4096, 256, LOWPASS, 0.5, [4096]),                         # three channels, whole Lines per workgroup (block-local)
    (2, 7, 9_000, 200, DC_BLOCK, 1.0, [4_000, 5_000]),                 # seven channels, the general look-back
])
def test_fused_chain_within_one_ulp_of_oracle(lines, C, frames, ntaps, q, g, calls, monkeypatch):
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    rng = np.random.default_rng(7 + lines * 31 + C)
    x = rng.uniform(-1, 1, size=(lines, frames, C)).astype(np.float32)
    got, names = run_chain(taps, q, g, x, calls)
    # one section and two forgetful sections (the sections one after the other over the tile,
    # ols32_kernel.hpp fused_epilogue_sections): fused, whether or not the Lines end on a segment boundary
    assert all("chain_fused_kernel" in n for n in names), names
    assert not np.isnan(got).any()
    worst, differ = 0.0, 0
    for l in sorted({0, lines // 2, lines - 1}):
        want = oracle_chain(taps, q, g, x[l])
        d = ulps(got[l], want)
        worst = max(worst, float(d.max()))
        differ += int((got[l] != want.astype(np.float32)).sum())
        assert d.max() <= 1.0, f"line {l}: {d.max()} ulp at frame {np.argmax(d) // C}"
    assert differ <= max(4, frames * C * 3 // 100_000), differ


def test_exact_mode_keeps_the_staged_bit_exact_chain(monkeypatch):
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    lines, C, frames, ntaps = 4, 4, 8192, 128
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    x = np.random.default_rng(3).uniform(-1, 1, size=(lines, frames, C)).astype(np.float32)
    got, names = run_chain(taps, LOWPASS, 0.5, x, [frames], exact=True)
    assert all("chain_fused" not in n for n in names), names
    for l in range(lines):
        assert np.array_equal(got[l], oracle_chain(taps, LOWPASS, 0.5, x[l]).astype(np.float32))


def test_full_config3_shape_every_line(monkeypatch):
    """BASELINE configs[3] at full size: 512 Lines x 8 ch x 4096 frames, FIR-256 -> biquad -> gain.
    EVERY Line is checked: the block-local look-back deals Lines to workgroups (b, b + 256), a mapping a
    spot check could miss.  All 512 against the staged bit-exact chain on the same device (which is the
    oracle's chain bit for bit: test_exact_mode_keeps_the_staged_bit_exact_chain), and 72 of them --
    both Lines of 36 workgroups -- against the oracle itself."""
    lines, C, frames, ntaps = 512, 8, 4096, 256
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    g = 0.7071067811865476
    x = np.stack([synth.samples(synth.line_seed(900 + l), 0, frames * C, np.float32).reshape(frames, C)
                  for l in range(lines)])
    got, names = run_chain(taps, LOWPASS, g, x, [frames])
    assert all("chain_fused_kernel" in n and "local" in n for n in names), names
    assert not np.isnan(got).any()
    exact, enames = run_chain(taps, LOWPASS, g, x, [frames], exact=True)
    assert all("chain_fused" not in n for n in enames), enames
    # the exact chain's float64 value is not on the host, so the floor is taken from its float32 result
    floor = 2.0 ** -24 * np.abs(exact).max(axis=(1, 2), keepdims=True)
    mag = np.maximum(np.abs(exact), floor).astype(np.float32)
    d = np.abs(got.astype(np.float64) - exact.astype(np.float64)) / np.spacing(mag).astype(np.float64)
    worst = d.reshape(lines, -1).max(axis=1)
    assert worst.max() <= 1.0, f"line {int(worst.argmax())}: {worst.max()} ulp"
    differ = (got != exact).reshape(lines, -1).sum(axis=1)
    assert differ.max() <= 4 and differ.sum() <= lines * frames * C * 3 // 100_000, (int(differ.max()), int(differ.sum()))
    for b in list(range(0, 256, 8)) + [1, 85, 170, 255]:
        for l in (b, b + 256):
            assert np.array_equal(exact[l], oracle_chain(taps, LOWPASS, g, x[l]).astype(np.float32)), f"exact chain, line {l}"
            dd = ulps(got[l], oracle_chain(taps, LOWPASS, g, x[l]))
            assert dd.max() <= 1.0, f"line {l}: {dd.max()} ulp"


@pytest.mark.parametrize("q", [LOWPASS, DC_BLOCK, TWO_SECTIONS], ids=["forgetful", "general", "two_sections"])
def test_cascade_state_survives_every_change_of_form(q, monkeypatch):
    """Between two fused launches the cascade's state lives in the plan's tagged slots (written by the
    launch's last tiles when the buffer ends on a segment boundary, frames % 32 == 0, by the tail
    kernel otherwise); the staged form and a partial restart use the biquad stage's own array.  A
    stream that keeps changing between all of them must still be the oracle's stream."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    lines, C, ntaps, g = 6, 4, 256, 0.5
    # (frames, exact): exact = PIPE_HIP_PARAM_EXACT on for this call -> the staged bit-exact chain
    plan = [(4096, False), (4096, False), (1000, False), (4096, True), (4096, False), (37, True), (2048, False),
            (4096 + 5, False), (4096, False)]
    total = sum(n for n, _ in plan)
    restart_before, restarted = 5, (1, 2)   # Lines 1..2 start again from silence before call 5
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    x = np.random.default_rng(11).uniform(-1, 1, size=(lines, total, C)).astype(np.float32)
    kw = dict(dtype=np.float32, lines=lines, max_batch=1)
    with P.Chain([P.Fir(taps, 8192, C, **kw), P.Biquad(q, 8192, C, **kw), P.Gain(g, 8192, C, **kw)]) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        outs, names, pos = [], [], 0
        for k, (n, exact) in enumerate(plan):
            if k == restart_before:
                p.start_lines(restarted[0], restarted[1] - restarted[0] + 1)
            p._set_param(3, [1.0 if exact else 0.0])
            xin = d_in[:, pos:pos + n, :].contiguous()
            y = torch.full_like(xin, float("nan"))
            p.process_batch(xin, y, n)
            torch.cuda.synchronize()
            names.append(p.kernel_name())
            outs.append(y)
            pos += n
        p.flush()
    got = torch.cat(outs, dim=1).cpu().numpy()
    assert ["chain_fused" in nm for nm in names] == [not e for _, e in plan], names
    cut = sum(n for n, _ in plan[:restart_before])
    for l in range(lines):
        if restarted[0] <= l <= restarted[1]:
            want = np.concatenate([oracle_chain(taps, q, g, x[l, :cut]), oracle_chain(taps, q, g, x[l, cut:])])
        else:
            want = oracle_chain(taps, q, g, x[l])
        d = ulps(got[l], want)
        assert d.max() <= 1.0, f"line {l}: {d.max()} ulp at frame {np.argmax(d) // C}"


@pytest.mark.parametrize("lines,C,calls", [
    (256, 2, [4096, 2500, 4096]),    # one pair: a tile's predecessor is the other half of its own wave
    (512, 6, [3000]),                # three pairs, four tiles: 12 items per Line
    (256, 6, [2000, 2000]),          # three pairs, three tiles: 9 items, the Line is padded to whole units
    (2100, 2, [1600]),               # Lines not a multiple of the workgroups: 9 Lines on some, 8 on others
    (768, 4, [15000]),               # three Lines of 40 items per workgroup: 8 rounds, the round counters wrap
    (256, 16, [1700]),               # eight pairs: a predecessor is 8 items (half a round) back
])
def test_block_local_look_back_equals_the_global_one(lines, C, calls, monkeypatch, ab_switch):
    """With at least as many Lines as CUs a workgroup runs whole Lines and the tile aggregates pass
    through a record ring in LDS instead of global memory (ols32_kernel.hpp, kLocalRing).  Same sums in
    the same order: bit for bit the result of the global look-back, and the oracle's within 1 ulp."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    frames = sum(calls)
    taps = synth.fir_lowpass_taps(256, f32_rounded=True)
    x = np.random.default_rng(lines + C).uniform(-1, 1, size=(lines, frames, C)).astype(np.float32)
    got, names = run_chain(taps, LOWPASS, 0.5, x, calls)
    assert all(n.endswith(",local>") for n in names), names
    for l in (0, 1, lines // 2, lines - 1):
        d = ulps(got[l], oracle_chain(taps, LOWPASS, 0.5, x[l]))
        assert d.max() <= 1.0, f"line {l}: {d.max()} ulp"
    if not ab_switch("PIPE_HIP_CHAIN_LOCAL", "0"):
        return  # (A/B leg: the global look-back forced on a shape the block-local one takes)
    ref, names = run_chain(taps, LOWPASS, 0.5, x, calls)
    assert all("chain_fused" in n and "local" not in n for n in names), names
    assert np.array_equal(got, ref)


def test_two_global_look_back_chains_on_two_streams(monkeypatch):
    """Tiles of the global look-back wait for tiles of other workgroups of their launch: two such
    launches sharing the device's CUs could starve each other, so chain_fused.hip runs launches of that
    form on one device one after the other.  Two chains driven from two threads, each on its own
    stream, 40 calls each: both streams stay the oracle's and nothing gives up (flush reports it)."""
    import threading
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    lines, C, F, calls = 200, 8, 4096, 25         # fewer Lines than CUs: the global form, 256 workgroups a launch
    taps = synth.fir_lowpass_taps(256, f32_rounded=True)
    xs = [np.random.default_rng(70 + k).uniform(-1, 1, size=(lines, F * calls, C)).astype(np.float32) for k in range(2)]
    outs, names, errs = [None, None], [None, None], []

    def drive(k):
        try:
            kw = dict(dtype=np.float32, lines=lines, max_batch=1)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(LOWPASS, F, C, **kw), P.Gain(0.5, F, C, **kw)]) as p:
                    p.start()
                    d_in = torch.from_numpy(xs[k]).cuda()
                    ys = []
                    for i in range(calls):
                        xin = d_in[:, i * F:(i + 1) * F, :].contiguous()
                        y = torch.empty_like(xin)
                        p.process_batch(xin, y, F)
                        ys.append(y)
                    st.synchronize()
                    p.flush()
                    names[k] = p.kernel_name()
                    outs[k] = torch.cat(ys, dim=1).cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=drive, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for k in range(2):
        assert "chain_fused" in names[k] and "local" not in names[k], names[k]
        for l in (0, lines - 1):
            d = ulps(outs[k][l], oracle_chain(taps, LOWPASS, 0.5, xs[k][l]))
            assert d.max() <= 1.0, f"chain {k} line {l}: {d.max()} ulp"


@pytest.mark.parametrize("lines", [24, 256], ids=["global", "local"])
def test_mutations_between_fused_launches(lines, monkeypatch):
    """mutable.Mutation on a fused chain (pipe.go:433: applied before the ProcessFunc of the buffer it
    travels with): new taps (the tap spectrum is re-uploaded on the launch stream), new biquad
    coefficients (the look-back matrices are rebuilt, the cascade's state carries over), a new gain --
    each takes effect at the next launch and at no other."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    from pipe_amd import _lib as L_
    C, F = 4, 4096
    h1 = synth.fir_lowpass_taps(256, f32_rounded=True)
    h2 = synth.fir_lowpass_taps(256, fc=0.1, f32_rounded=True)
    q1, q2 = synth.biquad_rbj_lowpass(), synth.biquad_rbj_lowpass(3000.0, q=1.5)
    g1, g2 = 0.5, 2.0
    x = np.random.default_rng(lines).uniform(-1, 1, size=(lines, 6 * F, C)).astype(np.float32)
    kw = dict(dtype=np.float32, lines=lines, max_batch=1)
    with P.Chain([P.Fir(h1, F, C, **kw), P.Biquad(q1, F, C, **kw), P.Gain(g1, F, C, **kw)]) as p:
        p.start()
        d_in = torch.from_numpy(x).cuda()
        ys, names = [], []
        for k in range(6):
            if k == 1:
                p.set_stage_param(0, L_.PARAM_TAPS, h2)
            if k == 2:
                p.set_stage_param(1, L_.PARAM_COEFFS, q2)
            if k == 3:
                p.set_stage_param(2, L_.PARAM_GAIN, [g2])
            if k == 5:
                p.set_stage_param(0, L_.PARAM_TAPS, h1)
                p.set_stage_param(1, L_.PARAM_COEFFS, q1)
            xin = d_in[:, k * F:(k + 1) * F, :].contiguous()
            y = torch.empty_like(xin)
            p.process_batch(xin, y, F)
            names.append(p.kernel_name())
            ys.append(y)
        p.flush()
        torch.cuda.synchronize()
    got = torch.cat(ys, dim=1).cpu().numpy()
    assert all("chain_fused" in n for n in names) and (lines < 256 or all("local" in n for n in names)), names
    for l in (0, lines - 1):
        fir, bq = O.Fir(h1, C), O.Biquad(q1, C)
        want = []
        for k in range(6):
            if k == 1:
                fir.set_taps(h2)
            if k == 2:
                bq.set_coeffs(q2)
            if k == 5:
                fir.set_taps(h1)
                bq.set_coeffs(q1)
            yk = bq.process(fir.process(x[l, k * F:(k + 1) * F].astype(np.float64)))
            want.append(O.gain(yk, g2 if k >= 3 else g1).reshape(F, C))
        d = ulps(got[l], np.concatenate(want))
        assert d.max() <= 1.0, f"line {l}: {d.max()} ulp at frame {np.argmax(d) // C}"


SLOW_PLUS_FAST = np.vstack([DC_BLOCK, synth.biquad_rbj_lowpass(2000.0)])


def test_two_section_cascade_at_the_config3_shape(monkeypatch):
    """512 Lines x 8 ch x 4096 frames, FIR-256 -> two-section biquad -> gain: the block-local form with
    one record ring per section, two launches (both sections' state slots carry).  Every Line against the
    staged bit-exact chain on the device, 24 Lines against the oracle; the global form on fewer Lines
    gives the same bits as the local one on the Lines they share."""
    lines, C, frames, ntaps, g = 512, 8, 4096, 256, 0.7071067811865476
    taps = synth.fir_lowpass_taps(ntaps, f32_rounded=True)
    x = np.stack([synth.samples(synth.line_seed(1200 + l), 0, 2 * frames * C, np.float32).reshape(2 * frames, C)
                  for l in range(lines)])
    got, names = run_chain(taps, TWO_SECTIONS, g, x, [frames, frames])
    assert all(n == "chain_fused_kernel<f32,f32,fir+biquad2+gain,local>" for n in names), names
    assert not np.isnan(got).any()
    exact, enames = run_chain(taps, TWO_SECTIONS, g, x, [frames, frames], exact=True)
    assert all("chain_fused" not in n for n in enames), enames
    floor = 2.0 ** -24 * np.abs(exact).max(axis=(1, 2), keepdims=True)
    mag = np.maximum(np.abs(exact), floor).astype(np.float32)
    d = np.abs(got.astype(np.float64) - exact.astype(np.float64)) / np.spacing(mag).astype(np.float64)
    worst = d.reshape(lines, -1).max(axis=1)
    assert worst.max() <= 1.0, f"line {int(worst.argmax())}: {worst.max()} ulp"
    differ = (got != exact).reshape(lines, -1).sum(axis=1)
    assert differ.sum() <= lines * 2 * frames * C * 3 // 100_000, (int(differ.max()), int(differ.sum()))
    for l in list(range(0, 256, 32)) + list(range(256, 512, 32)) + [1, 255, 257, 511, 100, 356, 200, 456]:
        dd = ulps(got[l], oracle_chain(taps, TWO_SECTIONS, g, x[l]))
        assert dd.max() <= 1.0, f"line {l}: {dd.max()} ulp"
    # the global look-back (fewer Lines than CUs) on the first 40 Lines: the same sums in the same order
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    got40, names40 = run_chain(taps, TWO_SECTIONS, g, x[:40], [frames, frames])
    assert all(n == "chain_fused_kernel<f32,f32,fir+biquad2+gain>" for n in names40), names40
    assert np.array_equal(got40, got[:40])


def test_two_sections_with_a_slow_one_take_the_staged_chain(monkeypatch):
    """A DC blocker next to a low-pass: the blocker does not forget within a look-back window, and the
    two-section form has no general look-back -- the staged chain runs, within the same tolerance."""
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    lines, C, frames = 3, 4, 20_480
    taps = synth.fir_lowpass_taps(128, f32_rounded=True)
    x = np.random.default_rng(5).uniform(-1, 1, size=(lines, frames, C)).astype(np.float32)
    got, names = run_chain(taps, SLOW_PLUS_FAST, None, x, [frames])
    assert all("chain_fused" not in n for n in names), names
    for l in range(lines):
        d = ulps(got[l], oracle_chain(taps, SLOW_PLUS_FAST, None, x[l]))
        assert d.max() <= 1.0, f"line {l}: {d.max()} ulp"


def test_a_fused_launch_that_gives_up_is_run_again_staged(monkeypatch):
    """PIPE_HIP_PARAM_DEBUG: tile 2 of every series publishes nothing and waits give up after 2 ms -- what a preempted
    predecessor tile does to a launch.  The reference aborts the whole run on a ProcessFunc error (pipe.go:438-440);
    here a synchronous entry puts the state back (the cascade's from the slot the launch did not write, the FIR's
    history from the half it did not write), runs the call again on the staged chain and answers OK with the
    stream's own result; the calls after it are the fused kernel's again and still the stream's."""
    from pipe_amd import _lib as L
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "64")  # (96 Lines x 6 tiles: fused also on this small call)
    lines, C, F, N = 96, 2, 4096, 256
    taps = synth.fir_lowpass_taps(N, f32_rounded=True)
    q = synth.biquad_rbj_lowpass()
    x = np.stack([synth.samples(synth.line_seed(l), 0, 4 * F * C, np.float32).reshape(4 * F, C) for l in range(lines)])
    want = np.stack([(O.Biquad(q, C).process(O.Fir(taps, C).process(x[l].astype(np.float64))).reshape(4 * F, C) * 0.5)
                     for l in range(lines)])
    kw = dict(dtype=np.float32, lines=lines, max_batch=1)
    with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.5, F, C, **kw)]) as ch:
        ch.start()
        for k in range(4):
            if k == 1:
                ch._set_param(5, [2.0, 2000.0])  # PIPE_HIP_PARAM_DEBUG {tile, limit_us}
            got = ch.process(np.ascontiguousarray(x[:, k * F:(k + 1) * F]))
            w = want[:, k * F:(k + 1) * F]
            floor = 2.0 ** -24 * np.abs(want).max(axis=(1, 2), keepdims=True)
            ulp = np.spacing(np.maximum(np.abs(w), floor).astype(np.float32)).astype(np.float64)
            err = np.abs(got.reshape(lines, F, C).astype(np.float64) - w.astype(np.float32).astype(np.float64)) / ulp
            assert err.max() <= 1.0, (k, err.max())
            if k != 1:
                assert ch.kernel_name().startswith("chain_fused_kernel"), (k, ch.kernel_name())
            else:
                assert not ch.kernel_name().startswith("chain_fused_kernel"), ch.kernel_name()
        ch.flush()
    # asynchronous entry: the failure is reported at the next synchronous one, the state is the one before the batch
    import torch
    with P.Chain([P.Fir(taps, F, C, **kw), P.Biquad(q, F, C, **kw), P.Gain(0.5, F, C, **kw)]) as ch:
        ch.start()
        d_in = [torch.from_numpy(np.ascontiguousarray(x[:, k * F:(k + 1) * F])).cuda() for k in range(3)]
        d_out = torch.empty_like(d_in[0])
        ch.process_batch(d_in[0], d_out, F)
        ch._set_param(5, [2.0, 2000.0])
        ch.process_batch(d_in[1], d_out, F)
        torch.cuda.synchronize()
        with pytest.raises(L.PipeHipError):
            ch.flush()
        for k in (1, 2):  # the failed batch again, then the next one
            ch.process_batch(d_in[k], d_out, F)
            torch.cuda.synchronize()
            w = want[:, k * F:(k + 1) * F]
            floor = 2.0 ** -24 * np.abs(want).max(axis=(1, 2), keepdims=True)
            ulp = np.spacing(np.maximum(np.abs(w), floor).astype(np.float32)).astype(np.float64)
            err = np.abs(d_out.cpu().numpy().reshape(lines, F, C).astype(np.float64) - w.astype(np.float32).astype(np.float64)) / ulp
            assert err.max() <= 1.0, (k, err.max())
        ch.flush()


# ---- float64 buffers: what a Go pipe carries (pipe.go:394,437) -------------------------------------------------------
@pytest.mark.parametrize("lines,C,ntaps,q,g,calls", [
    (24, 8, 256, LOWPASS, 0.7071067811865476, [4096, 4096]),      # configs[3] in small, two launches: history and slots carry in float64
    (256, 2, 256, LOWPASS, 0.5, [4096]),                          # whole Lines per workgroup: the block-local look-back
    (3, 4, 100, DC_BLOCK, 1.25, [20_000, 30_000]),                # the general look-back
    (5, 6, 64, TWO_SECTIONS, 0.9, [9_999, 20_001]),               # two sections, ragged ends: the tail kernel on float64 input
    (256, 2, 256, TWO_SECTIONS, None, [4096]),                    # two sections, block-local
    (6, 3, 128, LOWPASS, 0.8, [10_000, 6_011]),                   # three channels, ragged
    (9, 4, 256, FOUR_SECTIONS, 1.0, [4096, 4096]),                # four sections
])
def test_float64_buffers_take_the_fused_kernel_only_when_asked(lines, C, ntaps, q, g, calls, monkeypatch):
    """Without PIPE_HIP_PARAM_RELAXED_F64 a float64 chain is the staged chain of ordered forms, bit for bit the oracle's;
    with it (set on the chain: every stage gets it) the same calls take the fused kernel with float64 loads and stores,
    within the sum of the two stages' float64 bounds (include/pipe_hip.h); PIPE_HIP_PARAM_EXACT wins over the opt-in."""
    from tests import _tol
    monkeypatch.setenv("PIPE_HIP_FIR_OLS_MIN_ITEMS", "1")
    frames = sum(calls)
    taps = synth.fir_lowpass_taps(ntaps)
    x = np.stack([synth.samples(synth.line_seed(800 + l), 0, frames * C, np.float64).reshape(frames, C) for l in range(lines)])

    def run(relaxed, exact):
        kw = dict(dtype=np.float64, lines=lines, max_batch=1)
        stages = [P.Fir(taps, max(calls), C, **kw), P.Biquad(q, max(calls), C, **kw)]
        if g is not None:
            stages.append(P.Gain(g, max(calls), C, **kw))
        with P.Chain(stages) as p:
            p.start()
            if relaxed:
                p.set_relaxed_f64(True)
            if exact:
                p.set_exact(True)
            d_in = torch.from_numpy(x).cuda()
            outs, names, pos = [], [], 0
            for n in calls:
                xin = d_in[:, pos:pos + n, :].contiguous()
                y = torch.full_like(xin, float("nan"))
                p.process_batch(xin, y, n)
                torch.cuda.synchronize()
                names.append(p.kernel_name())
                outs.append(y)
                pos += n
            p.flush()
            return torch.cat(outs, dim=1).cpu().numpy(), names

    check = sorted({0, lines // 2, lines - 1})
    want = {l: oracle_chain(taps, q, g, x[l]) for l in check}
    plain, names = run(False, False)
    assert all("chain_fused" not in n and "fir_ols" not in n for n in names), names
    for l in check:
        assert np.array_equal(plain[l], want[l]), l
    got, names = run(True, False)
    assert all(n.startswith("chain_fused_kernel<f64,f64,") for n in names), names
    kap = _tol.kappa(q)
    gg = 1.0 if g is None else abs(g)
    for l in check:
        bound = 256.0 * kap * 2.0 ** -53 * np.abs(want[l]).max() + gg * kap * _tol.fir_f64_bound(taps)
        err = np.abs(got[l] - want[l]).max()
        assert err <= bound, (l, err, bound)
    pinned, names = run(True, True)
    assert all("chain_fused" not in n for n in names), names
    for l in check:
        assert np.array_equal(pinned[l], want[l]), l
