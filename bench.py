#!/usr/bin/env python3
"""bench.py -- Msamples/s through the 256-tap FIR Processor on MI355X.

Workload (BASELINE.json configs[1], batched on the time axis so that it can reach
a roofline at all -- SURVEY.md F8): per rank ONE Line, 2 channels, float32,
`--buffers` consecutive 4096-frame pipe buffers resident in HBM; one *step* = one
pass of the FIR Processor over that batch (pipe_hip_process_batch), with filter
history carried from step to step exactly as if the buffers had been pushed
through ProcessFunc one by one (tests/test_gpu_parity.py proves that equality).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: Lines are independent (run.go:112-132), so rank r simply owns Line r --
weak scaling, no data-path collective; RCCL is used only for the barrier and the
max-over-ranks of the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: "roofline" (algorithmic bytes / kernel time from hipEvents on the launch
stream) and "cpu_baseline" (the oracle's restatement of the reference loop timed
on the host cores; kind "port").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 f64 FMA lanes/clk x 2 flop x 2.4 GHz
BYTES_PER_SAMPLE = {"f32": 8, "f64": 16}  # in + out, taps/history amortised


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--buffers", type=int, default=131072,
                    help="consecutive 4096-frame buffers of the Line resident in HBM per step (default: 4.3 GB in, 4.3 GB out)")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--taps", type=int, default=256)
    ap.add_argument("--lines", type=int, default=1, help="Lines per rank")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-buffers", type=int, default=4096,
                    help="buffers of the same workload timed on the CPU (bounded sample)")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch

    from pipe_amd import processors as P
    from pipe_amd import shard, synth

    rank, world, local = shard.rank_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # RCCL only for the barrier and the max-over-ranks: Lines do not communicate.
    # PIPE_BENCH_DIST_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks (ranks
    # then share devices, so its numbers mean nothing): it exercises the rank / barrier / reduce
    # logic of this file end to end.
    backend = os.environ.get("PIPE_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = shard.init(backend, rank, world, device=torch.device("cuda", local) if backend == "nccl" else None)
    reduce_device = "cuda" if backend == "nccl" else "cpu"

    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    t_dtype = torch.float32 if args.dtype == "f32" else torch.float64
    F, C, N, K, L = args.frames, args.channels, args.taps, args.buffers, args.lines
    frames_per_line = F * K
    n_elems = L * frames_per_line * C

    taps = synth.fir_lowpass_taps(N, f32_rounded=(args.dtype == "f32"))
    fir = P.Fir(taps, F, C, dtype=np_dtype, device=local, lines=L, max_batch=K)
    fir.start()

    # synthetic input, generated on the device; global Line i lives on rank i mod world
    my_lines = shard.line_indices(rank, world, L * world)
    d_in = torch.empty(n_elems, dtype=t_dtype, device="cuda")
    d_out = torch.empty_like(d_in)
    for l, gl in enumerate(my_lines):
        P.synth_fill(d_in[l * frames_per_line * C:(l + 1) * frames_per_line * C], synth.line_seed(gl))
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        shard.barrier(dist)

    # The same workload pinned to the bit-exact direct form (PIPE_HIP_PARAM_EXACT): reported next
    # to the headline, never as `value`.  It runs BEFORE the headline's warmup: its ~80 ms of full
    # load also bring clocks and TLBs to steady state, whatever --warmup the caller chose.
    exact_ms = None
    if args.dtype == "f32":
        fir.set_exact(True)
        for _ in range(2):
            fir.process_batch(d_in, d_out, frames_per_line, stream=stream)
        torch.cuda.synchronize()
        fir.set_profiling(True)
        fir.kernel_time(reset=True)
        for _ in range(5):
            fir.process_batch(d_in, d_out, frames_per_line, stream=stream)
        torch.cuda.synchronize()
        ems, en = fir.kernel_time(reset=True)
        fir.set_profiling(False)
        fir.set_exact(False)
        exact_ms = ems / max(en, 1)

    for _ in range(args.warmup):
        fir.process_batch(d_in, d_out, frames_per_line, stream=stream)
    torch.cuda.synchronize()

    fir.set_profiling(True)  # hipEvents around the FIR kernel, on the launch stream
    fir.kernel_time(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fir.process_batch(d_in, d_out, frames_per_line, stream=stream)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms, launches = fir.kernel_time(reset=True)
    fir.set_profiling(False)
    kname = fir.kernel_name()

    elapsed = shard.max_over_ranks(elapsed, dist, device=reduce_device)

    # a cheap self-check that work really happened: DC gain of the filter is 1, so
    # the output mean tracks the input mean (no oracle here: that is tests/ + smoke())
    chk_in = float(d_in[: 1 << 20].double().mean().item())
    chk_out = float(d_out[N * C: (1 << 20)].double().mean().item())

    samples_per_step_rank = n_elems                 # scalar samples = frames x channels
    value = shard.aggregate_throughput(samples_per_step_rank, args.steps, world, elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    bps = BYTES_PER_SAMPLE[args.dtype]
    avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    achieved_gbs = samples_per_step_rank * bps / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    # real float64 flops the launched kernel form executes per scalar sample:
    #   direct form      : 2 * taps (ordered fma chain, bit-exact)
    #   overlap-save FFT : 1056 DP instructions (208 fma) per lane per 1024-point item of
    #                      (1024 - taps + 1) frames x 2 channels -> ~53 flop/sample at 256 taps
    is_ols = "ols" in kname
    if is_ols:
        flop_per_sample = (1056 + 208) * 64 / ((1024 - (N - 1)) * 2.0)
    else:
        flop_per_sample = 2.0 * N
    flops = flop_per_sample * samples_per_step_rank
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:  # PMC passes cannot run inside this process: the committed figure, if it is of this workload
            pmc = json.load(open(pmc_path))
            if pmc.get("algorithmic_bytes_per_launch") == samples_per_step_rank * bps and is_ols:
                traffic = pmc.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "Msamples/sec through 256-tap FIR Processor, 48 kHz 2 ch",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",  # arithmetic type (all forms compute in float64); buffers are config.io_dtype
        "data": "synthetic",
        "config": {
            "workload": f"configs[1]: 1 Line/GPU x {C} ch x {F}-frame buffers x {N}-tap FIR, "
                        f"{K} consecutive buffers resident in HBM per step",
            "lines_per_gpu": L, "channels": C, "buffer_frames": F, "buffers_per_step": K,
            "taps": N, "io_dtype": args.dtype, "parallelism": f"line-shard x{world}",
            "samples": "scalar (frames x channels)",
        },
        "mframes_per_s": round(value / C, 3),
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved_gbs, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "kernel": kname,
            "avg_kernel_ms": round(avg_kernel_s * 1e3, 5),
            "launches": launches,
            "algorithmic_bytes_per_launch": samples_per_step_rank * bps,
            "algorithm": "overlap-save, 1024-point float64 FFT per wave (<= 1 ulp f32 of the oracle)" if is_ols
                         else "direct form, ordered float64 fma chain (bit-exact)",
            "flop_per_sample": round(flop_per_sample, 1),
            # float64 VALU rate of the launched form, so that `frac` (HBM) is not misread: the
            # direct form is VALU-bound long before HBM
            "valu_f64": {"achieved_tflops": round(flops / avg_kernel_s / 1e12, 3) if avg_kernel_s else 0.0,
                         "peak_tflops": F64_VALU_PEAK_TFLOPS,
                         "frac": round(flops / avg_kernel_s / 1e12 / F64_VALU_PEAK_TFLOPS, 4) if avg_kernel_s else 0.0},
        },
        "selfcheck": {"in_mean": chk_in, "out_mean": chk_out},
    }
    if exact_ms:
        result["bit_exact_form"] = {
            "kernel": "fir_direct_kernel", "avg_kernel_ms": round(exact_ms, 5),
            "msamples_per_s": round(samples_per_step_rank / (exact_ms * 1e-3) / 1e6, 1),
            "valu_f64_frac": round(2.0 * N * samples_per_step_rank / (exact_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TFLOPS, 4),
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O  # cpu_baseline leg: the oracle is the thing timed
        cb = O.cpu_baseline(lines=1, channels=C, frames=F, buffers=args.cpu_buffers, ntaps=N, threads=1)
        result["cpu_baseline"] = {
            "value": round(cb["msamples_per_s"], 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"1 Line x {C} ch x {F}-frame buffers x {args.cpu_buffers} buffers, {N}-tap FIR, "
                      f"oracle restatement of pipe.Run (sync, 1 thread), {cb['seconds']:.1f} s",
        }
        ncpu = os.cpu_count() or 1
        per = max(8, args.cpu_buffers // 16)
        cbm = O.cpu_baseline(lines=ncpu, channels=C, frames=F, buffers=per, ntaps=N, threads=ncpu)
        result["cpu_baseline_all_cores"] = {
            "value": round(cbm["msamples_per_s"], 4), "unit": "Msamples/s", "cores": ncpu, "kind": "port",
            "sample": f"{ncpu} Lines (one per thread) x {per} buffers each, {cbm['seconds']:.1f} s",
        }

    if rank == 0:
        print(json.dumps(result))
    fir.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
