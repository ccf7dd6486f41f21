#!/usr/bin/env python3
"""bench.py -- Msamples/s through the Processor stage on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config {1,2,3}] [--threads]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 runs one rank per GPU, three ways to start them (the Lines each rank gets are the same):
  * under a launcher (RANK / WORLD_SIZE / MASTER_* in the environment): this process is one rank;
  * `python bench.py --gpus N` with no launcher: bench.py starts its own N ranks (it re-executes
    itself under torch.distributed.run on 127.0.0.1 with a free port) -- one process per GPU;
  * `--threads`: ONE process, one host thread + one HIP stream per GPU, no process group at all --
    the shape of the reference's host (one Go process, a goroutine per executor: run.go:171-196,
    merger.go:25-30; every C-ABI entry selects its handle's device itself).

--config names a BASELINE.json config:
  1 (default, the config the metric is quoted on): per rank ONE Line, 2 channels, float32,
    `--buffers` consecutive 4096-frame pipe buffers resident in HBM; one step = one pass of the
    256-tap FIR Processor over that batch (pipe_hip_process_batch), history carried from step to
    step exactly as if the buffers had gone through ProcessFunc one by one.  Weak scaling: rank r
    owns Line r.
  2: configs[2], 64 Lines x 256 buffers of 4096 x 2, the same FIR; the 64 Lines are dealt to the
    ranks (Line i on rank i mod N) -- strong scaling.
  3: configs[3], 512 Lines x 8 channels x 4096-frame buffers, FIR-256 -> biquad -> gain as one
    fused kernel; Line i on rank i mod N, and every Line advances by N buffers per step (one at N = 1):
    a rank's launch then always holds 512 Line-buffers and fills its GPU -- per-rank work constant,
    "weak (K = G buffers per Line per step)" (SURVEY.md 8d "C4"; `--buffers 1` gives the strong-scaling launch, `scale_projection` in the
    N = 1 line says what each costs).
Lines share no state (run.go:112-132): no data-path collective in any of them; RCCL carries only
the barrier and the max-over-ranks of the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
"roofline" (algorithmic bytes / kernel time from hipEvents attached to the kernel's dispatch),
"cpu_baseline" (the oracle's restatement of the reference loop on one host core; kind "port")
and, separately labelled, "cpu_baseline_all_cores" and "cpu_optimized".
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 f64 FMA lanes/clk x 2 flop x 2.4 GHz
BYTES_PER_SAMPLE = {"f32": 8, "f64": 16}  # in + out once, taps/history amortised


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=[1, 2, 3], default=1)
    ap.add_argument("--buffers", type=int, default=None,
                    help="consecutive 4096-frame buffers per Line resident in HBM per step "
                         "(config 1: 131072 = 4.3 GB in, 4.3 GB out; config 2: 256; config 3: 1)")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--channels", type=int, default=None)
    ap.add_argument("--taps", type=int, default=256)
    ap.add_argument("--lines", type=int, default=None, help="config 1: Lines per rank; configs 2/3: Lines in total")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--threads", action="store_true",
                    help="N > 1 as threads of ONE process (one per GPU, no process group) instead of N processes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the loaded window that samples socket power / shader clock")
    ap.add_argument("--power-window", type=float, default=2.5, help="seconds of the headline launch under the power sampler")
    ap.add_argument("--no-scale-projection", action="store_true", help="skip the one-GPU measurement of a rank's share (N = 1)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config[2]-shape line at N = 1")
    ap.add_argument("--no-per-call", action="store_true",
                    help="skip the per_call row (the counter passes do: a profiler that serialises dispatches turns every "
                         "parked doorbell wait into a watchdog timeout)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc passes of this command (N = 1)")
    ap.add_argument("--cpu-buffers", type=int, default=4096,
                    help="buffers of the same workload timed on the CPU (bounded sample)")
    return ap.parse_args()


def usable_cores() -> int:
    """Host cores this process may really use: the affinity mask, cut by a cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


KERNEL_SOURCES = ("ols_math.hpp", "ols32_core.hpp", "ols32_kernel.hpp", "fir_hist.hpp", "fir_ols32.hip",
                  "fir_ols_impl.hpp", "chain_fused.hip", "fir_ols.hip", "resampler.hip", "resampler_rows.hip", "Makefile")


def csrc_sha16() -> str:
    """Identity of the sources (and build flags) of the kernels this file times: a PMC figure is only
    quoted for the build it was taken on."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        p = os.path.join(ROOT, "pipe_amd", "csrc", name)
        h.update(name.encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(kernel: str, algorithmic_bytes: int):
    """HBM bytes per launch from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE), which cannot run
    inside this process: the committed figure, and only if it was taken on THIS kernel of THIS
    build at THIS size -- otherwise null."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        pmc = json.load(open(path))
    except (OSError, ValueError):
        return None
    for entry in pmc.get("kernels", [pmc]):
        if (entry.get("bench_kernel") == kernel and entry.get("csrc_sha16") == csrc_sha16()
                and entry.get("algorithmic_bytes_per_launch") == algorithmic_bytes):
            return entry.get("hbm_bytes_per_launch")
    return None


# device kernel names (rocprofv3) of the kernels this file reports traffic for
PMC_KERNELS = {
    "main_ols": r"fir_ols32_kernel<float, float, 0",
    "main_chain": r"fir_ols32_kernel<float, float, [12]",
    "c4_chain": r"fir_ols32_kernel<float, float, [12]",
    "c5_resampler": r"resample_(wave|pair|tiled)_kernel<",
    "biquad_alone": r"biquad_tile_kernel<float, float, 1, false, 3",
    "headline_f64": r"fir_ols32_kernel<double, double, 0",
}


def live_pmc(args, want):
    """HBM bytes per launch from the PMC counters, measured NOW: two more passes of this very command
    (short, without the CPU legs) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE`
    (separate passes, kernel trace only: MI355X_MICROARCH.md), FETCH_SIZE x 2 (gfx950 counts 64 B per
    128-B request; calibrated for this library's access widths in profiles/r03_fetch_calibration.txt)
    + WRITE_SIZE.  `want`: {label: (device kernel regex, algorithmic bytes per launch)}: the same device
    kernel may run several of this file's workloads (the headline and c3_shape), so only the dispatches
    whose counter is within 2.5x of what the algorithmic bytes predict are averaged.  Returns {label: bytes or None};
    any failure (no rocprofv3, a timeout) gives None and the committed figure is used instead."""
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {}
    out = tempfile.mkdtemp(prefix="pipe_bench_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-live-pmc",
             "--no-power", "--no-scale-projection", "--no-per-call",
             "--config", str(args.config), "--frames", str(args.frames), "--taps", str(args.taps), "--dtype", args.dtype]
    for name in ("buffers", "channels", "lines"):
        if getattr(args, name) is not None:
            child += [f"--{name}", str(getattr(args, name))]
    if args.no_secondary:
        child.append("--no-secondary")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = ["timeout", "-k", "5", "90", "rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--"] + child
            # (a session of its own: a pass that overruns is ended WITH its grandchildren -- a profiled child left behind
            # would share the GPU with everything timed after it)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=100)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, 9)
                except OSError:
                    pass
                pr.wait()
                raise
            per = {}
            for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
                db = sqlite3.connect(p)
                for name, cname, val, disp in db.execute("select name, counter_name, counter_value, dispatch_id from pmc_events"):
                    if cname == counter:
                        key = (p, disp)
                        per[key] = (name, per.get(key, (name, 0.0))[1] + val)  # one row per XCC: summed
            for label, (pat, alg) in want.items():
                # bytes in ~ bytes out ~ alg / 2; FETCH_SIZE (KiB) counts half of the bytes read
                expect_kib = alg / (4.0 if counter == "FETCH_SIZE" else 2.0) / 1024.0
                vals = [v for (n, v) in per.values() if re.search(pat, n) and 0.4 < v / expect_kib < 2.5]
                sums.setdefault(label, {})[counter] = (sum(vals) / len(vals)) if vals else None
    except (OSError, subprocess.SubprocessError, sqlite3.Error):
        sums = {}
    finally:
        shutil.rmtree(out, ignore_errors=True)
    res = {}
    for label, c in sums.items():
        f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        res[label] = int((2 * f + w) * 1024) if f and w else None
    return res


class PowerLog:
    """Socket power and shader clock from the amdgpu hwmon nodes (what scripts/clock_log.py reads), sampled
    by a thread while a loaded window runs: the card at this process's PCI address (sysfs_card_of); where that cannot
    be told every card in sysfs is sampled and the one whose median power is highest under load is taken."""

    def __init__(self, period=0.01, card=None):
        import threading
        self.period, self.stop, self.nodes, self.samples = period, threading.Event(), [], []
        for card in ([card] if card else sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))):
            n = {}
            for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
                for name in ("freq1_input", "power1_average", "power1_input"):
                    q = os.path.join(hw, name)
                    if os.path.exists(q):
                        n.setdefault("power" if name.startswith("power") else "sclk", q)
            if "power" in n:
                self.nodes.append(n)
                self.samples.append([])
        self.thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self.stop.is_set():
            for n, out in zip(self.nodes, self.samples):
                pw = self._read(n["power"])
                ck = self._read(n["sclk"]) if "sclk" in n else None
                if pw is not None:
                    out.append((pw / 1e6, ck / 1e6 if ck else None))
            time.sleep(self.period)

    def __enter__(self):
        if self.nodes:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.nodes:
            self.thread.join()
        return False

    def summary(self):
        """Medians of the card with the highest median power; None without hwmon nodes."""
        best = None
        for out in self.samples:
            # (the first fifth of the window is the ramp)
            body = out[len(out) // 5:]
            if len(body) < 5:
                continue
            pw = sorted(x[0] for x in body)
            ck = sorted(x[1] for x in body if x[1])
            row = {"power_w": round(pw[len(pw) // 2], 1), "power_w_p90": round(pw[int(len(pw) * 0.9)], 1),
                   "sclk_mhz": round(ck[len(ck) // 2], 0) if ck else None, "samples": len(body)}
            if best is None or row["power_w"] > best["power_w"]:
                best = row
        return best


def sysfs_card_of(torch, local):
    """/sys/class/drm/cardN/device of HIP device `local`, by PCI address (a box may list more cards than HIP sees --
    other tenants' GPUs -- and they may be busy); None when it cannot be told."""
    try:
        pr = torch.cuda.get_device_properties(local)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
    except Exception:  # noqa: BLE001
        return None
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if os.path.basename(os.path.realpath(card)).startswith(want):
            return card
    return None


def device_foreign_load(card):
    """What else is on THIS process's GPU before it launches anything (its context exists, no kernel has run): the
    card's gpu_busy_percent (0 - 100: a foreign process computing on the device) and the number of OTHER processes with
    a compute queue on the same GPU (KFD's per-process queue list; None where it cannot be read).  Scalars for the bench
    line: a slow driver run on a contended box must be tellable from a regression (VERDICT r5 "weak" 4)."""
    busy = None
    if card:
        try:
            busy = float(open(os.path.join(card, "gpu_busy_percent")).read())
        except (OSError, ValueError):
            pass
    others = None
    try:
        root = "/sys/class/kfd/kfd/proc"

        def gpuids(pid):
            out = set()
            for q in glob.glob(os.path.join(root, pid, "queues", "*", "gpuid")):
                try:
                    out.add(open(q).read().strip())
                except OSError:
                    pass
            return out
        mine = gpuids(str(os.getpid()))
        if mine:
            others = len([d for d in os.listdir(root) if d.isdigit() and int(d) != os.getpid() and gpuids(d) & mine])
    except OSError:
        pass
    return busy, others


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the
    driver's launcher line would (one process per GPU, rendezvous on 127.0.0.1)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, PIPE_BENCH_LAUNCH="self-spawned ranks (bench.py started torch.distributed.run)")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def run_threads(args) -> int:
    """--threads: the N ranks are threads of this process, rank r on GPU r."""
    import threading
    import torch
    from pipe_amd import shard
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not os.environ.get("PIPE_BENCH_SHARE_DEVICES"):
        sys.exit(f"--threads --gpus {args.gpus}: only {ndev} device(s) visible "
                 "(PIPE_BENCH_SHARE_DEVICES=1 lets ranks share devices: a rehearsal, its numbers mean nothing)")
    sync0 = shard.ThreadSync(args.gpus)
    results, errors = [None] * args.gpus, []

    def work(r):
        try:
            results[r] = run_rank(args, r, args.gpus, r % ndev, sync0.for_rank(r), "threads of one process")
        except BaseException as e:  # noqa: BLE001
            errors.append((r, e))
            sync0.abort()

    ts = [threading.Thread(target=work, args=(r,), name=f"rank{r}") for r in range(args.gpus)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)] or errors
        raise real[0][1]
    print(json.dumps(results[0]))
    return 0


def main():
    args = parse()
    if args.threads and args.gpus > 1:
        return run_threads(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    import torch
    from pipe_amd import shard
    rank, world, local = shard.rank_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # RCCL only for the barrier and the max-over-ranks: Lines do not communicate.
    # PIPE_BENCH_DIST_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks (ranks
    # then share devices, so its numbers mean nothing): it exercises the rank / barrier / reduce
    # logic of this file end to end.
    backend = os.environ.get("PIPE_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl" or os.environ.get("PIPE_BENCH_SHARE_DEVICES"):
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = shard.init(backend, rank, world, device=torch.device("cuda", local) if backend == "nccl" else None)
    sync = shard.ProcessSync(dist, "cuda" if backend == "nccl" else "cpu")
    sync.settle()
    if sync.fallback and rank == 0:
        print(f"warning: RCCL collective failed ({sync.fallback}); barriers and reductions over gloo", file=sys.stderr)
    launch = os.environ.get("PIPE_BENCH_LAUNCH", "launcher ranks (RANK/WORLD_SIZE from the environment)"
                            if world > 1 else "single process")
    if sync.fallback:
        launch += " -- RCCL's first all-reduce failed, barriers and reductions over gloo"
    result = run_rank(args, rank, world, local, sync, launch)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()
    return 0


def parity_stats(torch, got, want, floor, what):
    """How far a relaxed form's float32 results are from the bit-exact form's ((float)oracle bit for bit: tests/),
    in TRUE float32 ulps of the exact value -- no floor in the ulp -- and how much of the window lies below the
    contract's floor (include/pipe_hip.h, PIPE_HIP_PARAM_EXACT).  `floor`: a scalar or a tensor broadcastable to
    `want`.  Computed on the device after the timed region; the oracle is not involved."""
    a, b = got.double(), want.double()
    _, e = torch.frexp(want.abs())                      # |want| = m * 2^e, m in [0.5, 1)
    ulp = torch.ldexp(torch.ones_like(b), (e - 24).clamp(min=-149))   # spacing of float32 at |want| (denormals: 2^-149)
    err = (a - b).abs() / ulp
    fl = floor if torch.is_tensor(floor) else torch.tensor(float(floor), dtype=torch.float64, device=b.device)
    fl = fl.to(torch.float64)
    _, ef = torch.frexp(torch.maximum(b.abs(), fl.expand_as(b)).float())
    floored = (a - b).abs() / torch.ldexp(torch.ones_like(b), (ef - 24).clamp(min=-149))
    n = b.numel()
    differ = int((got != want).sum().item())
    return {
        "against": what, "samples": n,
        "frac_differing": round(differ / n, 9),
        "frac_off_by_more_than_1_true_ulp": round(float((err > 1.0).double().mean().item()), 9),
        "max_err_true_ulp": round(float(err.max().item()), 3),
        "p999999_err_true_ulp": round(float(torch.quantile(err.flatten()[: 1 << 24].float(), 0.999999).item()), 3),
        "frac_below_floor": round(float((b.abs() < fl).double().mean().item()), 6),
        "max_err_floored_ulp": round(float(floored.max().item()), 3),   # the contract as tested: <= 1
        "floor": "2^-24 * ||h||_1 * max|x| (FIR) / 2^-24 * max|y of the Line| (chain)",
    }


def run_rank(args, rank, world, local, sync, launch):
    """One rank's whole run; returns the bench line (meaningful on rank 0)."""
    import numpy as np
    import torch

    from pipe_amd import processors as P
    from pipe_amd import shard, synth

    torch.cuda.set_device(local)  # (per thread: the --threads ranks each select their own)
    dev = torch.device("cuda", local)
    torch.zeros(1, device=dev)    # (the context and its queues exist; nothing of the workload has run)
    torch.cuda.synchronize(dev)
    my_card = sysfs_card_of(torch, local)
    time.sleep(0.05)
    foreign_busy, foreign_procs = device_foreign_load(my_card) if rank == 0 else (None, None)

    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    t_dtype = torch.float32 if args.dtype == "f32" else torch.float64
    F, N = args.frames, args.taps
    cfg = args.config
    C = args.channels or (8 if cfg == 3 else 2)
    my_lines, total_lines, scaling = shard.plan_lines(cfg, rank, world, args.lines)
    K, scaling = shard.plan_buffers(cfg, world, args.buffers, scaling)
    L = len(my_lines)
    assert L >= 1, "more ranks than Lines"
    frames_per_line = F * K
    n_elems = L * frames_per_line * C

    taps = synth.fir_lowpass_taps(N, f32_rounded=(args.dtype == "f32"))
    kw = dict(dtype=np_dtype, device=local, lines=L, max_batch=K)
    fir = P.Fir(taps, F, C, **kw)
    if cfg == 3:
        proc = P.Chain([fir, P.Biquad(synth.biquad_rbj_lowpass(), F, C, **kw),
                        P.Gain(0.7071067811865476, F, C, **kw)])
    else:
        proc = fir
    proc.start()

    # synthetic input, generated on the device.  Configs 2 / 3 step over a working set that would fit the
    # 256 MiB Infinity Cache (config 3: 64 MiB in, 64 MiB out): they rotate through `nsets` distinct
    # input / output sets, more than 640 MiB together, so that every launch streams from and to HBM.
    itemsize = 4 if args.dtype == "f32" else 8
    nsets = 1 if cfg == 1 else max(1, min(32, -(-(640 << 20) // (2 * n_elems * itemsize))))
    d_ins, d_outs = [], []
    for j in range(nsets):
        t = torch.empty(n_elems, dtype=t_dtype, device=dev)
        for l, gl in enumerate(my_lines):
            P.synth_fill(t[l * frames_per_line * C:(l + 1) * frames_per_line * C], synth.line_seed(gl + 4096 * j))
        d_ins.append(t)
        d_outs.append(torch.empty_like(t))
    d_in, d_out = d_ins[0], d_outs[0]
    torch.cuda.synchronize(dev)
    # The timed launches go to ONE stream named explicitly (torch's default stream has the NULL handle,
    # which the C ABI reads as "the handle's own stream" and the Python harness then brackets with
    # cross-stream waits: ~30 us of dependency latency per step that is the harness's, not the path's).
    bench_stream = torch.cuda.Stream(device=dev)
    stream = bench_stream.cuda_stream

    def timed(p, steps, warmup, d_i, d_o, fpl, barrier=False, call=None):
        """(elapsed s, kernel ms total, launches, kernel name) of `steps` passes after `warmup`.
        `call` may be a LIST of calls: they are taken in turn (launch i runs calls[i % len]) -- distinct
        input / output sets rotated through the loop, so that no launch finds its buffers in the
        256 MiB Infinity Cache from the launch before."""
        calls = call if isinstance(call, (list, tuple)) else [call or (lambda: p.process_batch(d_i, d_o, fpl, stream=stream))]
        nc = len(calls)
        for i in range(warmup):
            calls[i % nc]()
        torch.cuda.synchronize(dev)
        p.set_profiling(True)  # hipEvents attached to the dominant kernel's own dispatch
        p.kernel_time(reset=True)
        if barrier:
            sync.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            calls[(warmup + i) % nc]()
        torch.cuda.synchronize(dev)
        if barrier:
            sync.barrier()
        el = time.perf_counter() - t0
        kms, n = p.kernel_time(reset=True)
        p.set_profiling(False)
        return el, kms, n, p.kernel_name()

    # The same workload pinned to the bit-exact direct form (PIPE_HIP_PARAM_EXACT): reported next
    # to the headline, never as `value`.  It runs BEFORE the headline's warmup: its full load also
    # brings clocks and TLBs to steady state, whatever --warmup the caller chose.
    exact_ms = None
    if args.dtype == "f32" and cfg == 1:
        fir.set_exact(True)
        _, ems, en, exact_kernel = timed(fir, 5, 2, d_in, d_out, frames_per_line)
        fir.set_exact(False)
        exact_ms = ems / max(en, 1)

    # Short launches (configs 2 / 3: 0.07 - 0.35 ms a step) read the clock ramp unless the device is busy
    # for a while first (the first launches after an idle period run 10 - 25 % slow, docs/NOTEBOOK.md
    # section 4); config 1's bit-exact leg above is that load already.  Untimed, before the W warm-up steps.
    if cfg != 1:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.25:
            for _ in range(50):
                proc.process_batch(d_in, d_out, frames_per_line, stream=stream)
            torch.cuda.synchronize(dev)
    main_calls = [(lambda a=a_, b=b_: proc.process_batch(a, b, frames_per_line, stream=stream)) for a_, b_ in zip(d_ins, d_outs)]
    elapsed, kernel_ms, launches, kname = timed(proc, args.steps, args.warmup, d_in, d_out, frames_per_line,
                                                barrier=True, call=main_calls)
    power = None
    if rank == 0 and world == 1 and cfg == 1 and not args.no_power:  # (N = 1 only: no rank may trail the others into the group's teardown)
        # socket power and shader clock over a loaded window of the SAME launch (>= 2.5 s; hwmon, 10 ms
        # samples, the first fifth dropped): the headline kernel runs at the package power cap, and the
        # clock it is held at belongs next to the roofline fraction
        proc.set_profiling(True)
        proc.kernel_time(reset=True)
        with PowerLog(card=my_card) as plog:
            t_pw, n_pw = time.perf_counter(), 0
            while time.perf_counter() - t_pw < args.power_window:
                for _ in range(20):
                    proc.process_batch(d_in, d_out, frames_per_line, stream=stream)
                torch.cuda.synchronize(dev)
                n_pw += 20
        pw_ms, pw_n = proc.kernel_time(reset=True)
        proc.set_profiling(False)
        power = plog.summary()
        if power:
            power["window_s"] = round(time.perf_counter() - t_pw, 2)
            power["launches"] = n_pw
            power["avg_kernel_ms"] = round(pw_ms / max(pw_n, 1), 5)  # the same launch over the whole loaded window
    elapsed = sync.max(elapsed)
    total_samples_per_step = int(sync.sum(n_elems)) if world > 1 else n_elems

    # a cheap self-check that work really happened: DC gain of the filter is 1, so the output mean
    # tracks the input mean (no oracle here: that is tests/ + smoke())
    chk_in = float(d_in[: 1 << 20].double().mean().item())
    chk_out = float(d_out[N * C: (1 << 20)].double().mean().item())

    value = total_samples_per_step * args.steps / elapsed / 1e6     # scalar samples = frames x channels
    ms_per_step = elapsed / args.steps * 1e3
    bps = BYTES_PER_SAMPLE[args.dtype]
    avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    achieved_gbs = n_elems * bps / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    # float64 operations the launched kernel form executes per scalar sample:
    #   direct form      : 2 * taps (ordered fma chain, bit-exact)
    #   overlap-save FFT : per lane and unit (two 1024-point items, 64 lanes) 760 add + 162 mul + 948 fma
    #                      = 2818 flops as issued (scripts/count_dp.sh; an fma counted as two);
    #   fused chain      : + two biquad passes over the item (9 and 10 flops per sample) and the scan
    is_fused = "chain_fused" in kname
    is_ols = "ols" in kname or is_fused
    if is_fused:
        tile = 1024 - (N - 1 + 31) // 32 * 32
        flop_per_sample = 2818 * 64 / (2 * tile * 2.0) + 19.0 * 1024 / tile + 1.0
    elif is_ols:
        flop_per_sample = 2818 * 64 / (2 * (1025 - N) * 2.0)
    else:
        flop_per_sample = 2.0 * N
    flops = flop_per_sample * n_elems
    alg_bytes = n_elems * bps

    workload = {
        1: f"configs[1]: 1 Line/GPU x {C} ch x {F}-frame buffers x {N}-tap FIR, "
           f"{K} consecutive buffers resident in HBM per step",
        2: f"configs[2]: {total_lines} Lines x {C} ch x {K} buffers of {F} frames x {N}-tap FIR, "
           f"Line i on GPU i mod {world}, one launch per step",
        3: f"configs[3]: {total_lines} Lines x {C} ch x {F}-frame buffers, {N}-tap FIR -> biquad -> gain "
           f"(one fused kernel), Line i on GPU i mod {world}, every Line advances by {K} buffer(s) per step",
    }[cfg]
    result = {
        "metric": "Msamples/sec through 256-tap FIR Processor, 48 kHz 2 ch, 1/2/4/8 GPU + CPU ref",  # BASELINE.json's metric, verbatim
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64",  # arithmetic type (all forms compute in float64); buffers are config.io_dtype
        "data": "synthetic",
        "config": {
            "workload": workload, "baseline_config": cfg,
            "lines_total": total_lines, "lines_this_gpu": L, "channels": C, "buffer_frames": F,
            "buffers_per_step": K, "taps": N, "io_dtype": args.dtype,
            # (the K plan of configs[3] on G ranks trades latency for scaling: a step hands back K buffers of every
            # Line at once -- K - 1 buffers later than a one-buffer step would have delivered the first of them)
            "added_latency_buffers": K - 1 if cfg == 3 else 0,
            "added_latency_ms_of_signal": round((K - 1) * F / 48.0, 2) if cfg == 3 else 0.0,
            "parallelism": f"line-shard x{world}", "ranks": launch, "samples": "scalar (frames x channels)",
        },
        "mframes_per_s": round(value / C, 3),
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved_gbs, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
            "traffic": committed_traffic(kname, alg_bytes),
            "kernel": kname,
            "avg_kernel_ms": round(avg_kernel_s * 1e3, 5),
            "launches": launches,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithm": ("FIR -> biquad -> gain fused: overlap-save FFT + segment scan + tile look-back "
                          "(<= 1 ulp f32 of the oracle chain)") if is_fused else
                         ("overlap-save, 1024-point float64 FFT per half-wave (<= 1 ulp f32 of the oracle)"
                          if is_ols else "direct form, ordered float64 fma chain (bit-exact)"),
            "flop_per_sample": round(flop_per_sample, 1),
            # float64 VALU rate of the launched form, so that `frac` (HBM) is not misread: the
            # direct form is VALU-bound long before HBM, and the overlap-save form
            # runs at the chip's power cap (profiles/r02_fir_ols32_phase_profile.txt)
            "valu_f64": {"achieved_tflops": round(flops / avg_kernel_s / 1e12, 3) if avg_kernel_s else 0.0,
                         "peak_tflops": F64_VALU_PEAK_TFLOPS,
                         "frac": round(flops / avg_kernel_s / 1e12 / F64_VALU_PEAK_TFLOPS, 4) if avg_kernel_s else 0.0},
        },
        "selfcheck": {"in_mean": chk_in, "out_mean": chk_out},
    }
    if exact_ms:
        result["bit_exact_form"] = {
            # (fir_mfma_kernel: the ordered fma chain on the float64 matrix pipe, whose dense peak on this part
            # equals the vector peak; fir_direct_kernel: the same chain on the VALU -- PIPE_HIP_FIR_NO_MFMA=1)
            "kernel": exact_kernel, "avg_kernel_ms": round(exact_ms, 5),
            "msamples_per_s": round(n_elems / (exact_ms * 1e-3) / 1e6, 1),
            "f64_frac": round(2.0 * N * n_elems / (exact_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS, 4),
            "f64_peak_tflops": F64_VALU_PEAK_TFLOPS,
        }

    if power:
        result["roofline"]["power"] = power  # medians over the loaded window: the clock the fraction was reached at
        # ... and as SCALARS (a parser that keeps scalars only must still see them): the state of the box next to the
        # fraction -- a throttled or contended box and a regression of the kernel look the same in `frac` alone
        result["roofline"]["sclk_mhz"] = power.get("sclk_mhz")
        result["roofline"]["power_w"] = power.get("power_w")
        result["roofline"]["steady_kernel_ms"] = power.get("avg_kernel_ms")
    result["roofline"]["host_gap_ms_per_step"] = round(ms_per_step - avg_kernel_s * 1e3, 4)
    result["roofline"]["gpu_busy_other"] = foreign_busy            # gpu_busy_percent before this process launched anything
    result["roofline"]["other_gpu_processes"] = foreign_procs      # other pids with a KFD context at that moment

    # What the relaxed contract means in numbers: the headline's form against the bit-exact form
    # on the first 512 buffers (4 M samples, both from silence) -- after the timed region, nothing of it is timed.
    if rank == 0 and world == 1 and cfg == 1 and args.dtype == "f32" and is_ols and not args.no_secondary:
        n_w = min(n_elems, 512 * F * C)
        w_rel = torch.empty(n_w, dtype=t_dtype, device=dev)
        w_ex = torch.empty(n_w, dtype=t_dtype, device=dev)
        fir.start()
        fir.process_batch(d_in[:n_w], w_rel, n_w // C, stream=stream)
        rel_kernel = fir.kernel_name()
        fir.set_exact(True)
        fir.start()
        fir.process_batch(d_in[:n_w], w_ex, n_w // C, stream=stream)
        ex_kernel = fir.kernel_name()
        fir.set_exact(False)
        fir.start()
        torch.cuda.synchronize(dev)
        floor = 2.0 ** -24 * float(np.abs(taps).sum()) * float(d_in[:n_w].abs().max().item())
        ps = parity_stats(torch, w_rel, w_ex, floor, f"{ex_kernel} (PIPE_HIP_PARAM_EXACT) on the same {n_w} samples")
        ps["kernel"] = rel_kernel
        result["roofline"]["parity_stats"] = ps
        del w_rel, w_ex

    # ---- the headline's workload on FLOAT64 buffers: what a Go pipe carries (pipe.go:394,437) ---------------------
    # Every stage output of the reference is allocator.Float64().  Without an opt-in such a batch takes the ordered sum
    # (bit for bit the oracle's: `bit_exact_form_f64` below, 2 N flops a sample on the float64 matrix pipe); with
    # PIPE_HIP_PARAM_RELAXED_F64 (hip.Options{RelaxedFloat64}) it takes the same overlap-save kernel as the headline
    # with 8-byte loads and stores: 16 algorithmic bytes a sample.  Reported beside the headline, never as `value`.
    if rank == 0 and world == 1 and cfg == 1 and args.dtype == "f32" and not args.no_secondary:
        Kd = K
        nd = frames_per_line * C
        free_b, _ = torch.cuda.mem_get_info(dev)
        while Kd > 512 and 2 * Kd * F * C * 8 > 0.6 * free_b:
            Kd //= 2
        nd = Kd * F * C
        x64 = torch.empty(nd, dtype=torch.float64, device=dev)
        y64 = torch.empty(nd, dtype=torch.float64, device=dev)
        P.synth_fill(x64, synth.line_seed(my_lines[0]))
        taps64 = synth.fir_lowpass_taps(N)
        with P.Fir(taps64, F, C, dtype=np.float64, device=local, lines=1, max_batch=Kd) as f64p:
            f64p.start()
            _, ekms, en64, ekn = timed(f64p, 3, 1, x64, y64, Kd * F)      # default: the ordered form
            ex_ms64 = ekms / max(en64, 1)
            f64p.set_relaxed_f64(True)
            f64p.start()
            el64, k64, n64, kn64 = timed(f64p, max(5, args.steps // 4), 3, x64, y64, Kd * F)
            ms64 = k64 / max(n64, 1)
            # distance from the ordered form on the first 512 buffers, in units of 2^-53 ||h||_1 max|x| (the header's
            # bound is 64 of them)
            n_w = min(nd, 512 * F * C)
            w_rel = torch.empty(n_w, dtype=torch.float64, device=dev)
            w_ex = torch.empty(n_w, dtype=torch.float64, device=dev)
            f64p.start()
            f64p.process_batch(x64[:n_w], w_rel, n_w // C, stream=stream)
            f64p.set_exact(True)
            f64p.start()
            f64p.process_batch(x64[:n_w], w_ex, n_w // C, stream=stream)
            exk = f64p.kernel_name()
            torch.cuda.synchronize(dev)
            unit = 2.0 ** -53 * float(np.abs(taps64).sum()) * float(x64[:n_w].abs().max().item())
            err64 = float((w_rel - w_ex).abs().max().item()) / unit
            del w_rel, w_ex
        result["headline_f64_buffers"] = {
            "workload": f"the headline's workload on float64 buffers: 1 Line x {C} ch x {Kd} buffers of {F} frames, {N}-tap FIR, "
                        "PIPE_HIP_PARAM_RELAXED_F64 set (hip.Options{RelaxedFloat64: true})",
            "kernel": kn64, "avg_kernel_ms": round(ms64, 5), "launches": n64,
            "msamples_per_s": round(nd / (ms64 * 1e-3) / 1e6, 1),
            "algorithmic_bytes_per_launch": nd * 16,
            "roofline_frac": round(nd * 16 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "traffic": None,
            "max_err_vs_ordered_form_in_2^-53_h1_xmax": round(err64, 3), "bound_in_the_same_unit": 64,
            "against": f"{exk} (PIPE_HIP_PARAM_EXACT) on the first {n_w} samples",
            "bit_exact_form_f64": {"kernel": ekn, "avg_kernel_ms": round(ex_ms64, 5),
                                   "msamples_per_s": round(nd / (ex_ms64 * 1e-3) / 1e6, 1),
                                   "note": "the default for float64 buffers: no opt-in, bit for bit the oracle's"},
            "speedup_over_the_ordered_form": round(ex_ms64 / ms64, 2),
        }
        del x64, y64

    # ---- the other BASELINE configs next to the headline (N = 1, config 1 only) ----------------------------
    # Every "fraction of HBM" below is a STREAMING figure: the timed launches rotate through distinct
    # input / output sets that together exceed twice the 256 MiB Infinity Cache (inputs: slices of the
    # headline's 4 GiB input at different offsets; outputs: slices of one 1.25 GiB arena), so no launch
    # finds its buffers in the cache from the launch before.  The same launch over ONE set (what earlier
    # rounds reported) rides beside it as `roofline_frac_l3_resident`.
    if cfg == 1 and world == 1 and not args.no_secondary and args.dtype == "f32":
        del d_out, d_outs, main_calls
        ARENA = 320 << 20  # elements: 1.25 GiB of float32
        arena = torch.empty(ARENA, dtype=t_dtype, device=dev)
        if d_in.numel() < ARENA:  # (a --buffers smaller than the default: its own synthetic input)
            d_src = torch.empty(ARENA, dtype=t_dtype, device=dev)
            P.synth_fill(d_src, synth.line_seed(7))
            torch.cuda.synchronize(dev)
        else:
            d_src = d_in

        def sets_of(n_in, n_out, min_bytes=640 << 20, max_sets=48):
            """[(input slice, output slice)]: distinct sets, >= min_bytes together (256-byte aligned starts)."""
            ai, ao = -(-n_in // 64) * 64, -(-n_out // 64) * 64
            k = max(2, min(max_sets, -(-min_bytes // ((n_in + n_out) * itemsize))))
            k = min(k, d_src.numel() // ai, ARENA // ao)
            return [(d_src[q * ai:q * ai + n_in], arena[q * ao:q * ao + n_out]) for q in range(k)]

        def both(p, n_in, n_out, fpl, steps, warmup, alg_bytes, mk_call=None):
            """The launch timed over ONE set (cache-resident where it fits) and rotating through the sets."""
            ss = sets_of(n_in, n_out)
            mk = mk_call or (lambda a, b: (lambda: p.process_batch(a, b, fpl, stream=stream)))
            calls = [mk(a, b) for a, b in ss]
            el1, k1, n1, _ = timed(p, steps, warmup, None, None, 0, call=calls[:1])
            el, k, n, name = timed(p, steps, warmup, None, None, 0, call=calls)
            ms1, ms = k1 / max(n1, 1), k / max(n, 1)
            return {"kernel": name, "avg_kernel_ms": round(ms, 5), "ms_per_step": round(el / steps * 1e3, 5),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "roofline_frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "roofline_frac_l3_resident": round(alg_bytes / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "avg_kernel_ms_l3_resident": round(ms1, 5),
                    "sets": len(ss), "set_bytes": (n_in + n_out) * itemsize,
                    "warmup_launches": warmup, "timed_launches": steps}, ms

        # SURVEY.md 8(d) "C3" shape: 64 Lines x 256 buffers (1 GiB per set: streaming by its size)
        L2, K2 = 64, 256
        n2 = L2 * F * K2 * C
        with P.Fir(taps, F, C, dtype=np_dtype, device=local, lines=L2, max_batch=K2) as f2:
            f2.start()
            di = d_src[:n2]
            do = arena[:n2]
            _, kms2, nl2, kn2 = timed(f2, 100, 300, di, do, F * K2)
            ms2 = kms2 / max(nl2, 1)
            result["c3_shape"] = {
                "workload": f"SURVEY 8d C3: {L2} Lines x {K2} buffers x {F} x {C} f32 resident, one launch per step",
                "kernel": kn2, "avg_kernel_ms": round(ms2, 5),
                "msamples_per_s": round(n2 / (ms2 * 1e-3) / 1e6, 1),
                "roofline_frac": round(n2 * bps / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "warmup_launches": 300, "timed_launches": 100,
            }

        # SURVEY.md 8(d) "C2" resident shape, to the letter: ONE Line x 256 consecutive buffers (8 MiB in + 8 MiB out).
        # 1364 transforms are not one unit for every wave of the chip: the launch is as long as a lone wave's unit
        # (15 - 16 us, profiles/r06_fir_small_calls.txt), not as the kernel's rate -- reported so that the point
        # exists, never as `value`.
        K1 = 256
        n1 = F * K1 * C
        with P.Fir(taps, F, C, dtype=np_dtype, device=local, lines=1, max_batch=K1) as f1:
            f1.start()
            r1, _ = both(f1, n1, n1, F * K1, 200, 300, n1 * bps)
            r1["workload"] = (f"SURVEY 8d C2: 1 Line x {K1} consecutive buffers x {F} x {C} f32 resident in HBM, one launch per step "
                              "(launch-bound at this size; the headline is the same Line at 131072 buffers)")
            r1["msamples_per_s"] = round(n1 / (r1["avg_kernel_ms"] * 1e-3) / 1e6, 1)
            result["c2_k256"] = r1

        # BASELINE configs[3] (SURVEY 8d "C4") at N = 1: 512 Lines x 8 ch x one 4096-frame buffer through
        # FIR-256 -> biquad -> gain as ONE fused kernel; the same launch `--config 3` times
        L4, C4 = 512, 8
        n4 = L4 * F * C4
        kw4 = dict(dtype=np_dtype, device=local, lines=L4, max_batch=1)
        with P.Chain([P.Fir(taps, F, C4, **kw4), P.Biquad(synth.biquad_rbj_lowpass(), F, C4, **kw4),
                      P.Gain(0.7071067811865476, F, C4, **kw4)]) as ch4:
            ch4.start()
            r4, ms4 = both(ch4, n4, n4, F, 400, 600, n4 * bps)
            r4["workload"] = f"configs[3]: {L4} Lines x {C4} ch x {F}-frame buffer, {N}-tap FIR -> biquad -> gain, one launch per step"
            r4["msamples_per_s"] = round(n4 / (r4["ms_per_step"] * 1e-3) / 1e6, 1)
            r4["traffic"] = committed_traffic(r4["kernel"], n4 * bps)
            # the fused kernel against the bit-exact staged chain on one full launch (every Line, both from silence)
            c_rel = torch.empty(n4, dtype=t_dtype, device=dev)
            c_ex = torch.empty(n4, dtype=t_dtype, device=dev)
            ch4.start()
            ch4.process_batch(d_src[:n4], c_rel, F, stream=stream)
            rel4 = ch4.kernel_name()
            ch4.set_exact(True)
            ch4.start()
            ch4.process_batch(d_src[:n4], c_ex, F, stream=stream)
            ex4 = ch4.kernel_name()
            ch4.set_exact(False)
            ch4.start()
            torch.cuda.synchronize(dev)
            line_max = c_ex.view(L4, F * C4).abs().amax(dim=1, keepdim=True).double() * 2.0 ** -24
            ps4 = parity_stats(torch, c_rel.view(L4, F * C4), c_ex.view(L4, F * C4), line_max,
                               f"the staged chain pinned exact ({ex4} first) on the same {n4} samples")
            ps4["kernel"] = rel4
            r4["parity_stats"] = ps4
            del c_rel, c_ex
            result["c4_chain"] = r4
        # the same chain on FLOAT64 buffers (what a Go pipe carries): with PIPE_HIP_PARAM_RELAXED_F64 on the chain it takes
        # the fused kernel too, 16 algorithmic bytes a sample; three rotating sets of 128 + 128 MiB (streaming)
        try:
            sets64 = [(torch.empty(n4, dtype=torch.float64, device=dev), torch.empty(n4, dtype=torch.float64, device=dev)) for _ in range(3)]
        except RuntimeError:
            sets64 = []
        if sets64:
            for q64, (xi, _) in enumerate(sets64):
                P.synth_fill(xi, synth.line_seed(900 + q64))
            kw64 = dict(dtype=np.float64, device=local, lines=L4, max_batch=1)
            with P.Chain([P.Fir(synth.fir_lowpass_taps(N), F, C4, **kw64), P.Biquad(synth.biquad_rbj_lowpass(), F, C4, **kw64),
                          P.Gain(0.7071067811865476, F, C4, **kw64)]) as ch64:
                ch64.start()
                _, ks, nls, kns = timed(ch64, 20, 10, None, None, 0,
                                        call=[(lambda a=a_, b=b_: ch64.process_batch(a, b, F, stream=stream)) for a_, b_ in sets64])
                ch64.set_relaxed_f64(True)
                ch64.start()
                _, kf, nlf, knf = timed(ch64, 400, 300, None, None, 0,
                                        call=[(lambda a=a_, b=b_: ch64.process_batch(a, b, F, stream=stream)) for a_, b_ in sets64])
                msf, mss_ = kf / max(nlf, 1), ks / max(nls, 1)
                result["c4_chain"]["float64_buffers"] = {
                    "workload": "the same launch on float64 buffers, PIPE_HIP_PARAM_RELAXED_F64 set (hip.Options{RelaxedFloat64: true})",
                    "kernel": knf, "avg_kernel_ms": round(msf, 5), "algorithmic_bytes_per_launch": n4 * 16,
                    "roofline_frac": round(n4 * 16 / (msf * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "msamples_per_s": round(n4 / (msf * 1e-3) / 1e6, 1),
                    "default_without_the_opt_in": {"kernel": kns + " (first stage of the staged chain, bit for bit the oracle's)",
                                                   "avg_chain_ms": round(mss_, 5),
                                                   "msamples_per_s": round(n4 / (mss_ * 1e-3) / 1e6, 1)}}
            del sets64
        # the same chain with 16 buffers per Line and launch (2 GiB per launch: streaming by its size): the kernel once a
        # launch's edges -- everybody's first window at once, the last epilogue alone -- are amortised; it then sits at
        # the socket's power cap (profiles/r04_chain_k_power.jsonl)
        K16 = 16
        n16 = L4 * K16 * F * C4
        kw16 = dict(dtype=np_dtype, device=local, lines=L4, max_batch=K16)
        if n16 <= min(ARENA, d_src.numel()):
            with P.Chain([P.Fir(taps, F, C4, **kw16), P.Biquad(synth.biquad_rbj_lowpass(), F, C4, **kw16),
                          P.Gain(0.7071067811865476, F, C4, **kw16)]) as ch16:
                ch16.start()
                _, k16, nl16, kn16 = timed(ch16, 40, 20, d_src[:n16], arena[:n16], K16 * F)
                ms16 = k16 / max(nl16, 1)
                result["c4_chain"]["steady_state_16_buffers_per_line"] = {
                    "kernel": kn16, "avg_kernel_ms": round(ms16, 5), "algorithmic_bytes_per_launch": n16 * bps,
                    "roofline_frac": round(n16 * bps / (ms16 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

        # the biquad stage alone (no BASELINE config of its own; configs[3] runs it fused): the time-segmented
        # form through LDS tiles, one pass, on the configs[3] shape and on ONE stereo Line of the same sample count
        bq = {}
        for tag, (Lb, Cb, Kb) in (("lines_512x8", (512, 8, 1)), ("one_stereo_line", (1, 2, 2048))):
            nb = Lb * Kb * F * Cb
            with P.Biquad(synth.biquad_rbj_lowpass(), F, Cb, dtype=np_dtype, device=local, lines=Lb, max_batch=Kb) as bqp:
                bqp.start()
                rb, _ = both(bqp, nb, nb, Kb * F, 200, 300, nb * bps)
                rb["workload"] = f"{Lb} Lines x {Cb} ch x {Kb} buffers of {F} frames, 1 section"
                rb["avg_ms"] = rb["avg_kernel_ms"]
                rb["msamples_per_s"] = round(nb / (rb["avg_kernel_ms"] * 1e-3) / 1e6, 1)
                bq[tag] = rb
        # the same kernel once a launch is long enough to forget its edges (16 buffers per Line: 1 GiB in + 1 GiB out,
        # streaming by its size; the 16.7 M-sample launches above are two rounds of workgroups that load, compute and
        # store in step)
        Kbs = 16
        nbs = 512 * Kbs * F * 8
        if nbs <= min(ARENA, d_src.numel()):
            with P.Biquad(synth.biquad_rbj_lowpass(), F, 8, dtype=np_dtype, device=local, lines=512, max_batch=Kbs) as bqp:
                bqp.start()
                _, kbs, nlbs, knbs = timed(bqp, 40, 20, d_src[:nbs], arena[:nbs], Kbs * F)
                msbs = kbs / max(nlbs, 1)
                bq["steady_state_512x8x16_buffers"] = {
                    "kernel": knbs, "avg_kernel_ms": round(msbs, 5), "algorithmic_bytes_per_launch": nbs * bps,
                    "roofline_frac": round(nbs * bps / (msbs * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "msamples_per_s": round(nbs / (msbs * 1e-3) / 1e6, 1)}
        result["biquad_alone"] = bq

        # BASELINE configs[4]: the 44.1 -> 48 kHz polyphase resampler (160/147, 24 taps per phase) over
        # 1024 consecutive 4096 x 2 buffers per launch, and the 2-input mix ("merger fan-in": the
        # build-defined sum) over a stream of the resampler's output size
        T5, up5, down5, K5 = 24, 160, 147, 1024
        n_in5 = K5 * F
        cap5 = -(-n_in5 * up5 // down5) + 1
        got = [0]
        with P.Resampler(synth.resampler_proto(up5, down5, T5), T5, up5, down5, F, C, dtype=np_dtype, device=local,
                         max_batch=K5) as rs:
            rs.start()

            def rs_mk(a, b):
                def call():
                    got[0] = rs.resample_batch(a, n_in5, b, cap5, stream=stream)
                return call
            rs_mk(d_src[:n_in5 * C], arena[:cap5 * C])()  # (the output frame count of such a call)
            torch.cuda.synchronize(dev)
            by5 = (n_in5 + got[0]) * C * 4
            r5, ms5 = both(rs, n_in5 * C, cap5 * C, 0, 200, 300, by5, mk_call=rs_mk)
            r5.update({"in_frames": n_in5, "out_frames": got[0],
                       "msamples_out_per_s": round(got[0] * C / (ms5 * 1e-3) / 1e6, 1),
                       "traffic": committed_traffic(r5["kernel"], by5)})
            c5 = {
                "workload": f"configs[4]: 1 Line x {C} ch, {up5}/{down5} polyphase resampler ({T5} taps per phase), "
                            f"{K5} buffers of {F} frames per launch; 2-input mix of the same stream",
                "resampler": r5,
            }
        # the same Line at the HEADLINE's stream length (the 1024-buffer launch above is 28 us, a third of it edges: the
        # first blocks' loads and the last blocks' stores; the headline takes 131 072 buffers a launch): steady state
        Ks = min(K, d_src.numel() // (F * C))
        n_ins = Ks * F
        caps = -(-n_ins * up5 // down5) + 1
        try:
            out_s = torch.empty(caps * C, dtype=t_dtype, device=dev)
        except RuntimeError:
            out_s = None
        if out_s is not None and Ks >= 8 * K5:
            with P.Resampler(synth.resampler_proto(up5, down5, T5), T5, up5, down5, F, C, dtype=np_dtype, device=local,
                             max_batch=Ks) as rss:
                rss.start()
                gots = [0]

                def rss_call():
                    gots[0] = rss.resample_batch(d_src[:n_ins * C], n_ins, out_s, caps, stream=stream)
                _, kss, nss, knss = timed(rss, 10, 3, None, None, 0, call=[rss_call])
                mss = kss / max(nss, 1)
                bys = (n_ins + gots[0]) * C * 4
                c5["resampler_steady_state"] = {
                    "workload": f"the same Line, {Ks} buffers of {F} frames per launch (the headline's stream length; streaming by its size)",
                    "kernel": knss, "avg_kernel_ms": round(mss, 5), "algorithmic_bytes_per_launch": bys,
                    "in_frames": n_ins, "out_frames": gots[0],
                    "roofline_frac": round(bys / (mss * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "msamples_out_per_s": round(gots[0] * C / (mss * 1e-3) / 1e6, 1)}
            del out_s
        # 64 such Lines in one launch (one Line x 32 us does not fill the chip)
        L5 = 64
        with P.Resampler(synth.resampler_proto(up5, down5, T5), T5, up5, down5, F, C, dtype=np_dtype, device=local,
                         lines=L5, max_batch=K5 // 16) as rsl:
            rsl.start()
            n_in5l = (K5 // 16) * F
            cap5l = -(-n_in5l * up5 // down5) + 1
            gotl = [0]

            def rsl_mk(a, b):
                def call():
                    gotl[0] = rsl.resample_batch(a, n_in5l, b, cap5l, stream=stream)
                return call
            rsl_mk(d_src[:L5 * n_in5l * C], arena[:L5 * cap5l * C])()
            torch.cuda.synchronize(dev)
            by5l = L5 * (n_in5l + gotl[0]) * C * 4
            r5l, ms5l = both(rsl, L5 * n_in5l * C, L5 * cap5l * C, 0, 200, 300, by5l, mk_call=rsl_mk)
            r5l.update({"lines": L5, "in_frames_per_line": n_in5l, "out_frames_per_line": gotl[0],
                        "msamples_out_per_s": round(L5 * gotl[0] * C / (ms5l * 1e-3) / 1e6, 1)})
            c5["resampler_64_lines"] = r5l
        # the same stream with 8 channels (configs[3]'s Lines): 256 buffers of 4096 x 8 -- the same bytes per launch -- take
        # the row form (resampler_rows.hip: lanes = periods of the phase pattern, taps in scalar registers)
        C8, K8 = 8, K5 * C // 8
        with P.Resampler(synth.resampler_proto(up5, down5, T5), T5, up5, down5, F, C8, dtype=np_dtype, device=local,
                         max_batch=K8) as rs8:
            rs8.start()
            n_in8 = K8 * F
            cap8 = -(-n_in8 * up5 // down5) + 1
            got8 = [0]

            def rs8_mk(a, b):
                def call():
                    got8[0] = rs8.resample_batch(a, n_in8, b, cap8, stream=stream)
                return call
            rs8_mk(d_src[:n_in8 * C8], arena[:cap8 * C8])()
            torch.cuda.synchronize(dev)
            by8 = (n_in8 + got8[0]) * C8 * 4
            r8, ms8 = both(rs8, n_in8 * C8, cap8 * C8, 0, 200, 300, by8, mk_call=rs8_mk)
            r8.update({"channels": C8, "in_frames": n_in8, "out_frames": got8[0],
                       "msamples_out_per_s": round(got8[0] * C8 / (ms8 * 1e-3) / 1e6, 1)})
            c5["resampler_8_channels"] = r8
        nm = got[0] * C
        with P.Mix(2, F, C, dtype=np_dtype, device=local, max_batch=cap5 // F + 1) as mx:
            mx.start()
            am = -(-nm // 64) * 64  # (256-byte aligned inputs: an odd element offset would put the mix on its 4-byte path)
            km = max(2, min(16, -(-(640 << 20) // (3 * nm * 4))))
            mcalls = []
            for q in range(km):
                a5, b5, mo = d_src[(2 * q) * am:(2 * q) * am + nm], d_src[(2 * q + 1) * am:(2 * q + 1) * am + nm], arena[q * am:q * am + nm]
                mcalls.append(lambda a5=a5, b5=b5, mo=mo: mx.mix_batch([a5, b5], mo, got[0], stream=stream))
            _, k1, n1, _ = timed(mx, 200, 300, None, None, 0, call=mcalls[:1])
            _, kmsm, nlm, knm = timed(mx, 200, 300, None, None, 0, call=mcalls)
            msm, msm1 = kmsm / max(nlm, 1), k1 / max(n1, 1)
            c5["mix"] = {"kernel": knm, "avg_kernel_ms": round(msm, 5), "algorithmic_bytes_per_launch": 3 * nm * 4,
                         "roofline_frac": round(3 * nm * 4 / (msm * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "roofline_frac_l3_resident": round(3 * nm * 4 / (msm1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "sets": km, "set_bytes": 3 * nm * 4}
        result["c5_resampler_mix"] = c5

        # the HBM reference point: the gain Processor (1 flop per sample) streaming 1 GiB in, 1 GiB out
        ng = 256 << 20
        with P.Gain(0.5, F, C, dtype=np_dtype, device=local, max_batch=ng // (F * C)) as gp:
            gp.start()
            gcalls = [lambda: gp.process_batch(d_src[:ng], arena[:ng], ng // C, stream=stream)]
            _, kg, ngl, kng = timed(gp, 50, 20, None, None, 0, call=gcalls)
            msg = kg / max(ngl, 1)
            result["gain_reference"] = {
                "workload": "gain Processor, 1 GiB in + 1 GiB out per launch (streaming by its size)",
                "kernel": kng, "avg_kernel_ms": round(msg, 5), "algorithmic_bytes_per_launch": 2 * ng * 4,
                "roofline_frac": round(2 * ng * 4 / (msg * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

        # ---- configs[1] as the reference runs it: ONE 4096 x 2 buffer per ProcessFunc call (latency-bound, never `value`) ----
        # 8 FIR handles (8 Lines' stages) called round-robin -- the synchronous host loop's order, run.go:112-132 -- on the
        # plain path, with the device's doorbell held by one of them, and all of them in the SHARED doorbell queue
        # (PIPE_HIP_PARAM_RESIDENT_SHARED); through ctypes (a C caller: ~4 us less a call, examples/percall_latency.c)
        pc = {}
        xb = np.ascontiguousarray(synth.samples(synth.line_seed(5), 0, F * C).reshape(F, C).astype(np_dtype))
        for mode_pc in (() if args.no_per_call else ("plain", "exclusive_doorbell", "shared_doorbell_queue")):
            hs = [P.Fir(taps, F, C, dtype=np_dtype, device=local) for _ in range(8)]
            for h in hs:
                h.start()
                if mode_pc == "exclusive_doorbell":
                    h.set_resident(True)
                elif mode_pc == "shared_doorbell_queue":
                    h.set_resident_shared(True)
            for _ in range(10):
                for h in hs:
                    h.process(xb)
            tt = []
            for _ in range(150):
                for h in hs:
                    t0 = time.perf_counter()
                    h.process(xb)
                    tt.append(time.perf_counter() - t0)
            tt.sort()
            pc[mode_pc] = {"median_us": round(tt[len(tt) // 2] * 1e6, 2), "p90_us": round(tt[len(tt) * 9 // 10] * 1e6, 2)}
            for h in hs:
                h.close()
        pc["workload"] = f"8 handles x one {F} x {C} {args.dtype} buffer per pipe_hip_process call, {N}-tap FIR, round-robin, 1200 calls each mode"
        if not args.no_per_call:
            result["per_call"] = pc

        # ---- what a rank's share costs (one GPU; no multi-GPU hardware is needed for this) ---------------------
        # configs[3] / configs[2] deal their Lines to G ranks: rank r runs total / G Lines.  t(L) below is one
        # launch over L Lines on THIS GPU (streaming sets); the projected speed-up of G GPUs is
        # t(total) / t(total / G) -- Lines share nothing, the only cross-rank cost is the barrier.  With few Lines
        # a launch no longer fills the chip (configs[3] at G = 8: 768 units for 2048 waves), so `--config 3 --gpus G`
        # advances every Line by K = G buffers per launch (per-rank work constant, shard.plan_buffers): the
        # `k_plan` rows are that launch -- t(total / G Lines x G buffers), efficiency against t(total Lines x 1).
        if not args.no_scale_projection:
            sp = {"method": "one GPU; t(L) = avg kernel ms of one launch over L Lines (streaming sets), "
                            "speedup(G) = t(total) / t(total / G), efficiency = speedup / G"}
            rows3, t3 = [], {}
            for Lr, Kr in ((512, 1), (256, 1), (128, 1), (64, 1), (256, 2), (128, 4), (64, 8)):
                kwr = dict(dtype=np_dtype, device=local, lines=Lr, max_batch=Kr)
                nr = Lr * Kr * F * C4
                with P.Chain([P.Fir(taps, F, C4, **kwr), P.Biquad(synth.biquad_rbj_lowpass(), F, C4, **kwr),
                              P.Gain(0.7071067811865476, F, C4, **kwr)]) as chr_:
                    chr_.start()
                    rr, msr = both(chr_, nr, nr, Kr * F, 200, 300, nr * bps)
                    t3[(Lr, Kr)] = msr
                    rows3.append({"lines": Lr, "buffers_per_line": Kr, "kernel": rr["kernel"], "ms": round(msr, 5),
                                  "roofline_frac": rr["roofline_frac"]})
            sp["config3"] = {
                "rows": rows3,
                "strong": {str(G): {"speedup": round(t3[(512, 1)] / t3[(512 // G, 1)], 3),
                                    "efficiency": round(t3[(512, 1)] / t3[(512 // G, 1)] / G, 3)} for G in (2, 4, 8)},
                # K = G buffers per Line per launch: a rank's launch does the work of the N = 1 launch
                "k_plan": {str(G): {"efficiency": round(t3[(512, 1)] / t3[(512 // G, G)], 3)} for G in (2, 4, 8)},
            }
            rows2, t2 = [], {}
            for Lr in (64, 32, 16, 8):
                nr = Lr * K2 * F * C
                with P.Fir(taps, F, C, dtype=np_dtype, device=local, lines=Lr, max_batch=K2) as fr:
                    fr.start()
                    _, kr, nlr, knr = timed(fr, 60, 100, d_src[:nr], arena[:nr], F * K2)
                    t2[Lr] = kr / max(nlr, 1)
                    rows2.append({"lines": Lr, "buffers_per_line": K2, "kernel": knr, "ms": round(t2[Lr], 5),
                                  "roofline_frac": round(nr * bps / (t2[Lr] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
            sp["config2"] = {
                "rows": rows2,
                "strong": {str(G): {"speedup": round(t2[64] / t2[64 // G], 3),
                                    "efficiency": round(t2[64] / t2[64 // G] / G, 3)} for G in (2, 4, 8)},
            }
            result["scale_projection"] = sp
        del arena

    # roofline.traffic measured in THIS run (N = 1): the committed figure above stays only if the passes fail
    if rank == 0 and world == 1 and not args.no_live_pmc and os.environ.get("PIPE_BENCH_LIVE_PMC", "1") != "0":
        want = {"main": (PMC_KERNELS["main_chain" if is_fused else "main_ols"], alg_bytes)} if is_ols else {}
        if "c4_chain" in result and cfg == 1:
            want["c4_chain"] = (PMC_KERNELS["c4_chain"], result["c4_chain"]["algorithmic_bytes_per_launch"])
        if "c5_resampler_mix" in result:
            want["c5_resampler"] = (PMC_KERNELS["c5_resampler"], result["c5_resampler_mix"]["resampler"]["algorithmic_bytes_per_launch"])
        if "headline_f64_buffers" in result:
            want["headline_f64"] = (PMC_KERNELS["headline_f64"], result["headline_f64_buffers"]["algorithmic_bytes_per_launch"])
        if "biquad_alone" in result:  # (both shapes run this kernel over the same byte count)
            want["biquad_alone"] = (PMC_KERNELS["biquad_alone"], result["biquad_alone"]["lines_512x8"]["algorithmic_bytes_per_launch"])
        t_pmc = time.perf_counter()
        live = live_pmc(args, want) if want else {}
        src_live = "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH_SIZE x 2 + WRITE_SIZE)"
        src_file = "profiles/pmc_latest.json (committed; kernel, bytes and source hash match this build)"
        result["roofline"]["traffic_source"] = src_file if result["roofline"]["traffic"] else None
        if live.get("main"):
            result["roofline"]["traffic"], result["roofline"]["traffic_source"] = live["main"], src_live
        if live.get("c4_chain"):
            result["c4_chain"]["traffic"], result["c4_chain"]["traffic_source"] = live["c4_chain"], src_live
        if live.get("c5_resampler"):
            result["c5_resampler_mix"]["resampler"]["traffic"] = live["c5_resampler"]
            result["c5_resampler_mix"]["resampler"]["traffic_source"] = src_live
        if live.get("headline_f64"):
            result["headline_f64_buffers"]["traffic"], result["headline_f64_buffers"]["traffic_source"] = live["headline_f64"], src_live
        if live.get("biquad_alone"):
            result["biquad_alone"]["traffic"] = live["biquad_alone"]  # mean over the two shapes' launches
            result["biquad_alone"]["traffic_source"] = src_live
        result["roofline"]["pmc_passes_s"] = round(time.perf_counter() - t_pmc, 1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O  # cpu_baseline leg: the oracle is the thing timed
        cb = O.cpu_baseline(lines=1, channels=2, frames=F, buffers=args.cpu_buffers, ntaps=N, threads=1)
        result["cpu_baseline"] = {
            "value": round(cb["msamples_per_s"], 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"1 Line x 2 ch x {F}-frame buffers x {args.cpu_buffers} buffers, {N}-tap FIR, "
                      f"oracle restatement of pipe.Run (sync, 1 thread), {cb['seconds']:.1f} s",
        }
        ncpu = usable_cores()
        per = max(8, 2 * args.cpu_buffers // max(ncpu, 1))
        cbm = O.cpu_baseline(lines=ncpu, channels=2, frames=F, buffers=per, ntaps=N, threads=ncpu)
        result["cpu_baseline_all_cores"] = {
            "value": round(cbm["msamples_per_s"], 4), "unit": "Msamples/s", "cores": ncpu, "kind": "port",
            "sample": f"{ncpu} Lines (one thread each; usable cores = affinity mask cut by the cgroup quota) "
                      f"x {per} buffers each, {cbm['seconds']:.1f} s",
        }
        try:
            # ~10^10 scalar samples: a few seconds on a current server CPU
            cbo = O.cpu_optimized(lines=4 * ncpu, channels=2, frames=F,
                                  buffers=max(64, int(1.2e10 / (4 * ncpu * F * 2))), ntaps=N, threads=ncpu)
            result["cpu_optimized"] = {
                "value": round(cbo["msamples_per_s"], 3), "unit": "Msamples/s", "cores": ncpu,
                "kind": "optimised float32 CPU FIR (AVX via -O3 -march=native, float32 accumulation): "
                        "NOT the reference's algorithm, NOT bit-compatible with the oracle; reported "
                        "separately from cpu_baseline (SURVEY.md 8d)",
                "sample": f"{4 * ncpu} Lines x {cbo['buffers_per_line']} buffers, {cbo['seconds']:.1f} s",
            }
            result["cpu_baseline_all_cores"]["speedup_over_1_core"] = round(
                cbm["msamples_per_s"] / max(cb["msamples_per_s"], 1e-9), 2)
        except Exception as e:  # noqa: BLE001 -- an optional leg must not cost the bench line
            result["cpu_optimized"] = {"error": str(e)[:200]}

    proc.close()
    return result


if __name__ == "__main__":
    sys.exit(main())
