"""Line -> rank sharding and cross-rank timing for multi-GPU runs.

Lines are independent units of work (run.go:112-132 iterates them round-robin, no
state is shared between them), so N GPUs = N processes, each owning a static
slice of the Lines; there is no data-path collective.  The only communication is
the barrier that brackets the timed region and a MAX all-reduce of the elapsed
time, which works identically over RCCL ("nccl") on GPUs and gloo on CPU.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple


def line_indices(rank: int, world: int, total_lines: int) -> List[int]:
    """Static round-robin: Line i runs on rank i mod world."""
    return list(range(rank, total_lines, world))


def plan_lines(config: int, rank: int, world: int, lines: Optional[int] = None):
    """Which Lines this rank runs for bench.py --config N (BASELINE.json configs[N]).
    Returns (global Line indices of this rank, Lines in total, "weak" | "strong").
      config 1: `lines` (default 1) Lines PER rank -- the job grows with the ranks: weak scaling;
      config 2 / 3: 64 / 512 Lines IN TOTAL (or `lines`), Line i on rank i mod world: strong scaling."""
    if config == 1:
        per = lines or 1
        total = per * world
        return line_indices(rank, world, total), total, "weak"
    total = lines or (64 if config == 2 else 512)
    return line_indices(rank, world, total), total, "strong"


def plan_buffers(config: int, world: int, buffers: Optional[int] = None, scaling: str = "strong"):
    """Consecutive pipe buffers every Line advances by per step, and the scaling label that goes with it.
      config 1: 131072 (the whole stream of the step resident); config 2: 256;
      config 3: `world` -- with 512 / world Lines a one-buffer launch no longer fills a GPU (at 8 ranks:
      768 units for 2048 waves), so every rank's launch takes `world` buffers of each of its Lines: the
      launch holds 512 Line-buffers whatever the rank count, per-rank work is constant: weak scaling ON THE TIME
      AXIS -- a step now delivers K buffers of every Line at once, i.e. K - 1 buffers (K - 1 times 85 ms of
      signal at 48 kHz) of added latency per step; the label says so wherever the number is quoted.
    An explicit `buffers` keeps the label plan_lines gave (config 3 --buffers 1: strong scaling)."""
    if buffers:
        return int(buffers), scaling
    if config == 1:
        return 131072, scaling
    if config == 2:
        return 256, scaling
    k = max(1, world)
    return k, (k_plan_label(k) if world > 1 else scaling)


def k_plan_label(k: int) -> str:
    """The `scaling` value of a config-3 run on the K plan (the driver's own default run is config 1: plain "weak")."""
    return f"weak (K = {k} buffers per Line per step)"


def step_units(config: int, rank: int, world: int, step: int, lines: Optional[int] = None, buffers: Optional[int] = None):
    """The (Line, buffer index) pairs rank `rank` processes in step `step` (0-based) of bench.py --config N
    --gpus world: its Lines (plan_lines) times the K consecutive buffers of the step (plan_buffers)."""
    mine, _, scaling = plan_lines(config, rank, world, lines)
    k, _ = plan_buffers(config, world, buffers, scaling)
    return [(l, b) for l in mine for b in range(step * k, (step + 1) * k)]


def rank_from_env() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, rank: int, world: int, device=None):
    """Returns torch.distributed (initialised) or None for a single process."""
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    import datetime
    # a rank whose first collective never completes (a peer died in RCCL's setup) must not hang the others
    # for ever: ProcessSync.settle() then falls through to the gloo agreement
    kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("PIPE_BENCH_DIST_TIMEOUT_S", "180")))}
    del device  # (RCCL's communicator is made by the first device collective, inside settle()'s try: no device_id)
    if backend == "nccl":
        # RCCL for device tensors, gloo for host tensors: ProcessSync.settle() falls back to the
        # latter if the first RCCL collective fails (the data path has no collective to lose)
        # (no device_id: RCCL's communicator is then made by the first device collective, inside settle()'s try)
        backend = "cpu:gloo,cuda:nccl"
    try:
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    except Exception:  # noqa: BLE001 -- e.g. gloo cannot find an interface: RCCL alone then (no fallback to offer)
        if backend != "cpu:gloo,cuda:nccl":
            raise
        if dist.is_initialized():
            dist.destroy_process_group()
        dist.init_process_group("nccl", rank=rank, world_size=world, **kw)
    return dist


def barrier(dist) -> None:
    if dist is not None:
        dist.barrier()


def max_over_ranks(value: float, dist, device: Optional[str] = None) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist, device: Optional[str] = None) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class ProcessSync:
    """The ranks are PROCESSES (one per GPU): barrier and reductions through torch.distributed
    (RCCL on GPUs, gloo on CPU); with dist None a single process."""

    def __init__(self, dist, device: Optional[str] = None):
        self.dist, self.device = dist, device
        self.fallback = None  # why the host-tensor path was taken, if it was

    def settle(self) -> None:
        """One RCCL all-reduce on a device tensor; if ANY rank fails it, every rank moves its barriers and
        reductions to host tensors (gloo) -- agreed through a gloo reduction, so nobody is left waiting."""
        if self.dist is None or self.device in (None, "cpu"):
            return
        import torch
        ok, why = 1.0, ""
        try:
            t = torch.ones(1, dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t)
            torch.cuda.synchronize()
            ok = 1.0 if float(t.item()) == float(self.dist.get_world_size()) else 0.0
        except Exception as e:  # noqa: BLE001 -- whatever RCCL raised, the bench must still report
            ok, why = 0.0, f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.float64)
        try:
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        except Exception:  # noqa: BLE001 -- no host-tensor backend in the group (RCCL-only init): nothing to agree through
            if ok < 1.0:
                raise RuntimeError(f"RCCL collective failed and no gloo backend to fall back to: {why}")
            return
        if float(flag.item()) < 1.0:
            self.device = "cpu"
            self.fallback = why or "another rank's RCCL all-reduce failed"

    def barrier(self) -> None:
        if self.dist is not None and self.device == "cpu":
            import torch
            self.dist.all_reduce(torch.zeros(1))  # (a barrier on the host-tensor backend)
            return
        barrier(self.dist)

    def max(self, value: float) -> float:
        return max_over_ranks(value, self.dist, self.device)

    def sum(self, value: float) -> float:
        return sum_over_ranks(value, self.dist, self.device)


class ThreadSync:
    """The ranks are THREADS of one process, one per GPU -- what a Go host is: one process, a
    goroutine per Line executor (run.go:171-196, merger.go:25-30), every C-ABI entry selecting its
    handle's device itself.  No process group: a threading.Barrier and a shared slot per rank."""

    def __init__(self, world: int, shared=None):
        import threading
        if shared is None:
            shared = {"barrier": threading.Barrier(world), "slots": [0.0] * world}
        self.world, self.shared, self.rank = world, shared, 0

    def for_rank(self, rank: int) -> "ThreadSync":
        t = ThreadSync(self.world, self.shared)
        t.rank = rank
        return t

    def barrier(self) -> None:
        self.shared["barrier"].wait()

    def abort(self) -> None:  # a rank failed: the others must not wait for it for ever
        self.shared["barrier"].abort()

    def _gather(self, value: float):
        self.shared["slots"][self.rank] = float(value)
        self.barrier()
        vals = list(self.shared["slots"])
        self.barrier()  # nobody overwrites a slot before everybody has read it
        return vals

    def max(self, value: float) -> float:
        return max(self._gather(value))

    def sum(self, value: float) -> float:
        return float(sum(self._gather(value)))


def aggregate_throughput(samples_per_rank_step: int, steps: int, world: int, max_elapsed_s: float) -> float:
    """Whole-job Msamples/s: all ranks' samples over the slowest rank's time."""
    return samples_per_rank_step * world * steps / max_elapsed_s / 1e6
