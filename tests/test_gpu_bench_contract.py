"""bench.py's one JSON line: the fields the driver and the judge read (a short run, bounded CPU legs)."""
import json
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    # a session of its own, ended as a group whatever happens: a bench run cut short by the per-test limit must not
    # leave its counter passes (rocprofv3 grandchildren) on the GPU under the tests that follow
    pr = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, env=e, start_new_session=True)
    try:
        stdout, stderr = pr.communicate(timeout=400)
    finally:
        try:
            os.killpg(pr.pid, signal.SIGKILL)
        except OSError:
            pass
    assert pr.returncode == 0, stderr[-2000:]
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]          # exactly ONE JSON line
    return json.loads(lines[0])


@pytest.mark.limit(420)
def test_default_line_has_every_contract_field():
    d = run_bench("--steps", "3", "--warmup", "1", "--buffers", "4096", "--cpu-buffers", "64")
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == baseline["metric"] and d["unit"] == "Msamples/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "configs[1]" in d["config"]["workload"]
    assert "model" not in d["config"]
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] * 1e3 / (4096 * 4096 * 2) - 1.0) < 0.02
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) < 1.0
    assert r["algorithmic_bytes_per_launch"] == 8 * 4096 * 4096 * 2     # 8 B per scalar sample
    # HBM traffic from the PMC counters, measured by two more passes of the command itself
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.6
    if r["traffic"] is not None:
        assert r["traffic_source"].startswith(("live", "profiles/"))
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "Msamples/s" and c["sample"]
    # (the ordered-fma form: on the float64 matrix pipe for a call of this size, fir_mfma.hip)
    assert d["bit_exact_form"]["kernel"].startswith(("fir_mfma_kernel", "fir_direct_kernel")) and d["bit_exact_form"]["f64_frac"] > 0
    # BASELINE configs[3] and configs[4] ride in the same line (what the driver records)
    c4 = d["c4_chain"]
    assert c4["kernel"].startswith("chain_fused_kernel") and c4["algorithmic_bytes_per_launch"] == 8 * 512 * 4096 * 8
    assert 0 < c4["avg_kernel_ms"] <= c4["ms_per_step"] * 1.05
    assert abs(c4["roofline_frac"] - c4["algorithmic_bytes_per_launch"] / (c4["avg_kernel_ms"] * 1e-3) / 8e12) < 1e-3
    c5 = d["c5_resampler_mix"]
    r5 = c5["resampler"]
    assert r5["kernel"].startswith(("resample_wave_kernel", "resample_pair_kernel", "resample_tiled_kernel")) and r5["in_frames"] == 1024 * 4096
    assert r5["out_frames"] in (-(-1024 * 4096 * 160 // 147), 1024 * 4096 * 160 // 147)
    assert r5["algorithmic_bytes_per_launch"] == (r5["in_frames"] + r5["out_frames"]) * 2 * 4
    assert c5["mix"]["kernel"] == "mix_kernel<f32>" and 0 < c5["mix"]["roofline_frac"] < 1.0
    # the biquad stage alone: the one-pass LDS-tile form, on the configs[3] shape and on one long stereo Line
    for tag in ("lines_512x8", "one_stereo_line"):
        b = d["biquad_alone"][tag]
        assert b["kernel"].startswith("biquad_tile_kernel") and b["algorithmic_bytes_per_launch"] == 8 * 512 * 4096 * 8
        assert 0 < b["roofline_frac"] < 1.0
    bs = d["biquad_alone"]["steady_state_512x8x16_buffers"]   # the same kernel once a launch forgets its edges (1 GiB + 1 GiB)
    assert bs["kernel"].startswith("biquad_tile_kernel") and bs["algorithmic_bytes_per_launch"] == 16 * 8 * 512 * 4096 * 8
    assert 0.3 < bs["roofline_frac"] < 0.85
    # SURVEY 8(d)'s "C2" resident shape to the letter: 1 Line x 256 buffers, one launch (1364 transforms: overlap-save
    # since round 6's dispatch rule; a launch of this size is as long as a lone wave's unit)
    c2 = d["c2_k256"]
    assert c2["kernel"].startswith("fir_ols_kernel") and c2["algorithmic_bytes_per_launch"] == 8 * 256 * 4096 * 2
    assert c2["sets"] >= 2 and c2["sets"] * c2["set_bytes"] >= 512 << 20 and 0 < c2["roofline_frac"] < 0.44
    tb = d["biquad_alone"].get("traffic")   # (live PMC passes; absent without rocprofv3)
    assert tb is None or 1.0 <= tb / (8 * 512 * 4096 * 8) < 1.2
    # every "of HBM" figure is a streaming one: the timed launches rotate through sets that exceed twice the
    # 256 MiB Infinity Cache; the one-set (cache-resident) figure rides beside it
    for o in (c4, r5, c5["resampler_64_lines"], c5["mix"], d["biquad_alone"]["lines_512x8"], d["biquad_alone"]["one_stereo_line"]):
        assert o["sets"] >= 2 and o["sets"] * o["set_bytes"] >= 512 << 20
        assert 0 < o["roofline_frac"] < 0.85 and 0 < o["roofline_frac_l3_resident"] < 1.0
    ss = c4["steady_state_16_buffers_per_line"]   # the chain once a launch's edges are amortised (2 GiB per launch)
    assert ss["kernel"].startswith("chain_fused_kernel") and ss["algorithmic_bytes_per_launch"] == 16 * c4["algorithmic_bytes_per_launch"]
    assert 0.15 < ss["roofline_frac"] < 0.5
    g = d["gain_reference"]
    assert g["kernel"].startswith("gain_kernel") and 0.5 < g["roofline_frac"] < 0.85   # (the guide's achievable HBM rate: ~0.79)
    # socket power and shader clock over a loaded window of the headline launch (hwmon; None where the box has no such node)
    pw = r.get("power")
    assert pw is None or (pw["window_s"] >= 2.0 and 200 < pw["power_w"] < 1600 and (pw["sclk_mhz"] is None or 500 < pw["sclk_mhz"] < 2600))
    # a rank's share on one GPU: t(L) of configs[3] / configs[2] and the projected efficiency at 2 / 4 / 8 GPUs
    sp = d["scale_projection"]
    assert [(x["lines"], x["buffers_per_line"]) for x in sp["config3"]["rows"]] == [(512, 1), (256, 1), (128, 1), (64, 1), (256, 2), (128, 4), (64, 8)]
    assert set(sp["config3"]["strong"]) == set(sp["config3"]["k_plan"]) == set(sp["config2"]["strong"]) == {"2", "4", "8"}
    t3 = {(x["lines"], x["buffers_per_line"]): x["ms"] for x in sp["config3"]["rows"]}
    assert abs(sp["config3"]["k_plan"]["8"]["efficiency"] - t3[(512, 1)] / t3[(64, 8)]) < 0.02
    assert sp["config3"]["k_plan"]["8"]["efficiency"] > 0.7   # the plan bench.py --config 3 --gpus 8 runs fills every rank


def test_config3_line_names_the_fused_chain():
    d = run_bench("--config", "3", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-live-pmc")
    assert "configs[3]" in d["config"]["workload"] and d["scaling"] == "strong"
    assert d["roofline"]["kernel"].startswith("chain_fused_kernel") and d["config"]["lines_total"] == 512
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 8 * 512 * 4096 * 8


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment: two processes, Line i on
    rank i mod 2, ONE line with n_gpus 2 and both ranks' samples in `value` (the 1-GPU box shares the
    device between the ranks over gloo: a rehearsal of the launch / barrier / reduce logic, no scaling
    figure -- run.go:112-132 Lines share nothing, merger.go:25-30 one executor per goroutine)."""
    d = run_bench("--gpus", "2", "--config", "3", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                  "--no-live-pmc", env={"PIPE_BENCH_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["lines_total"] == 512 and d["config"]["lines_this_gpu"] == 256
    # the K-per-rank plan (shard.plan_buffers): with 2 ranks every Line advances by 2 buffers per step, a rank's
    # launch holds 512 Line-buffers like the N = 1 launch -- per-rank work constant
    assert "self-spawned" in d["config"]["ranks"] and d["config"]["buffers_per_step"] == 2
    # ... which is weak scaling on the TIME axis, and the line says what it costs: one more buffer of latency per step
    assert d["scaling"] == "weak (K = 2 buffers per Line per step)"
    assert d["config"]["added_latency_buffers"] == 1 and abs(d["config"]["added_latency_ms_of_signal"] - 85.33) < 0.01
    # value counts BOTH ranks' samples: 512 Lines x 2 x 4096 x 8 per step over the slowest rank's time
    assert abs(d["value"] * d["ms_per_step"] * 1e3 / (512 * 2 * 4096 * 8) - 1.0) < 0.02
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 8 * 256 * 2 * 4096 * 8   # one rank's launch
    # --buffers 1: the strong-scaling launch (256 Lines x one buffer per rank)
    d = run_bench("--gpus", "2", "--config", "3", "--buffers", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                  "--no-live-pmc", env={"PIPE_BENCH_DIST_BACKEND": "gloo"})
    assert d["scaling"] == "strong" and d["config"]["buffers_per_step"] == 1 and d["config"]["added_latency_buffers"] == 0
    assert abs(d["value"] * d["ms_per_step"] * 1e3 / (512 * 4096 * 8) - 1.0) < 0.02
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 8 * 256 * 4096 * 8


def test_threads_mode_is_one_process_with_a_thread_per_rank():
    """--threads: the ranks are threads of ONE process (the shape of a Go host), no process group."""
    d = run_bench("--gpus", "2", "--threads", "--config", "3", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                  env={"PIPE_BENCH_SHARE_DEVICES": "1"})
    assert d["n_gpus"] == 2 and d["config"]["lines_total"] == 512 and d["config"]["lines_this_gpu"] == 256
    assert "threads" in d["config"]["ranks"] and d["scaling"] == "weak (K = 2 buffers per Line per step)"
    assert abs(d["value"] * d["ms_per_step"] * 1e3 / (512 * 2 * 4096 * 8) - 1.0) < 0.02
    w = run_bench("--gpus", "2", "--threads", "--steps", "3", "--warmup", "1", "--buffers", "2048", "--no-cpu-baseline",
                  env={"PIPE_BENCH_SHARE_DEVICES": "1"})
    assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["config"]["lines_total"] == 2
    assert abs(w["value"] * w["ms_per_step"] * 1e3 / (2 * 2048 * 4096 * 2) - 1.0) < 0.02
