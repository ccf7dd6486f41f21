"""world_size-2 CPU test (gloo) of the N>1 path of bench.py: Line sharding, the
barrier-bracketed timed region and the max-over-ranks reduction (pipe_amd/shard.py).
The per-rank work is a stand-in: the oracle FIR over the rank's own Lines, which
also proves that sharded Lines reproduce the single-process result exactly."""
import os
import socket
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_lines, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from oracle import oracle as O
    from pipe_amd import shard, synth
    dist = shard.init("gloo", rank, world)
    mine = shard.line_indices(rank, world, total_lines)
    taps = synth.fir_lowpass_taps(32)
    shard.barrier(dist)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))  # rank 1 is the slow one
    sums = {}
    for l in mine:
        x = synth.samples(synth.line_seed(l), 0, 2 * 512).reshape(512, 2)
        sums[l] = float(O.Fir(taps, 2).process(x).sum())
    shard.barrier(dist)
    elapsed = time.perf_counter() - t0
    worst = shard.max_over_ranks(elapsed, dist)
    n_lines = shard.sum_over_ranks(float(len(mine)), dist)
    # bench.py --config 2 / 3: a fixed set of Lines dealt to the ranks (strong scaling); every rank
    # sums the ranks' sample counts exactly as bench.py does for `value`
    for cfg, total in ((2, 64), (3, 512)):
        got, tot, kind = shard.plan_lines(cfg, rank, world)
        assert tot == total and kind == "strong" and got == list(range(rank, total, world))
        assert shard.sum_over_ranks(float(len(got)), dist) == total
    got, tot, kind = shard.plan_lines(1, rank, world, 3)
    assert kind == "weak" and tot == 3 * world and len(got) == 3
    # the K-per-rank plan of config 3: every Line advances by `world` buffers per step, so a rank's launch
    # always holds 512 Line-buffers (per-rank work constant: "weak"); all ranks together advance the 512
    # Lines by the same K, and an explicit --buffers keeps the strong-scaling label
    got, tot, kind = shard.plan_lines(3, rank, world)
    K, kind = shard.plan_buffers(3, world, None, kind)
    assert K == world and kind == shard.k_plan_label(world) and kind.startswith("weak") and len(got) * K == 512
    assert shard.sum_over_ranks(float(len(got) * K), dist) == 512 * world
    assert shard.plan_buffers(3, world, 1, "strong") == (1, "strong")
    assert shard.plan_buffers(2, world, None, "strong") == (256, "strong")
    assert shard.plan_buffers(1, world, None, "weak") == (131072, "weak")
    np.save(os.path.join(out_dir, f"r{rank}.npy"),
            np.array([elapsed, worst, n_lines] + [v for _, v in sorted(sums.items())] ))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing(tmp_path):
    world, total_lines = 2, 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total_lines, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npy")
    r1 = np.load(tmp_path / "r1.npy")
    # both ranks agree on the max, and it is at least the slow rank's sleep
    assert r0[1] == r1[1] and r0[1] >= max(r0[0], r1[0]) - 1e-9 and r0[1] >= 0.1
    assert r0[2] == total_lines and r1[2] == total_lines
    # every Line processed exactly once, with the single-process result
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from pipe_amd import shard, synth
    assert sorted(shard.line_indices(0, 2, 5) + shard.line_indices(1, 2, 5)) == list(range(5))
    taps = synth.fir_lowpass_taps(32)
    want = {l: float(O.Fir(taps, 2).process(synth.samples(synth.line_seed(l), 0, 1024).reshape(512, 2)).sum())
            for l in range(total_lines)}
    got = dict(zip(shard.line_indices(0, 2, 5), r0[3:]))
    got.update(zip(shard.line_indices(1, 2, 5), r1[3:]))
    assert got == want
    assert shard.aggregate_throughput(1000, 10, 2, 0.5) == 1000 * 2 * 10 / 0.5 / 1e6


def test_k_per_rank_plan_fills_every_rank():
    """config 3 dealt to G ranks: Lines x buffers per rank is the N = 1 launch's 512 for every G that divides it."""
    sys.path.insert(0, ROOT)
    from pipe_amd import shard
    for world in (1, 2, 4, 8):
        per_rank = []
        for r in range(world):
            mine, total, kind = shard.plan_lines(3, r, world)
            K, kind = shard.plan_buffers(3, world, None, kind)
            per_rank.append(len(mine) * K)
            assert kind == (f"weak (K = {world} buffers per Line per step)" if world > 1 else "strong")
        assert per_rank == [512] * world


def test_every_line_buffer_pair_is_processed_exactly_once_for_1_2_4_8_ranks():
    """The K plan changes WHEN a (Line, buffer) is processed, never whether: over S steps on G ranks the union of the
    ranks' units is every Line x every buffer of the first S * K, each exactly once -- for configs[3] (K = G), its
    strong-scaling form (--buffers 1), configs[2] and the weak configs[1]."""
    sys.path.insert(0, ROOT)
    from collections import Counter
    from pipe_amd import shard
    for world in (1, 2, 4, 8):
        for config, lines, buffers, steps in ((3, None, None, 3), (3, None, 1, 3), (2, None, 2, 2), (1, 2, 3, 2), (3, 37, None, 2)):
            seen = Counter()
            total = None
            for r in range(world):
                _, total, scaling = shard.plan_lines(config, r, world, lines)
                k, _ = shard.plan_buffers(config, world, buffers, scaling)
                for t in range(steps):
                    seen.update(shard.step_units(config, r, world, t, lines, buffers))
            want = {(l, b) for l in range(total) for b in range(steps * k)}
            assert set(seen) == want and set(seen.values()) == {1}, (world, config, lines, buffers)


def test_thread_ranks_barrier_and_reductions():
    """bench.py --threads: the ranks are threads of one process (the reference's host shape:
    run.go:171-196, one goroutine per executor); same partition, barrier and reductions."""
    import threading
    sys.path.insert(0, ROOT)
    from pipe_amd import shard
    world = 3
    root = shard.ThreadSync(world)
    out = [None] * world

    def work(r):
        s = root.for_rank(r)
        mine, total, kind = shard.plan_lines(3, r, world)
        s.barrier()
        time.sleep(0.02 * (r + 1))
        s.barrier()
        out[r] = (s.max(0.1 * (r + 1)), s.sum(len(mine)), s.max(float(r)), total, kind)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert all(o is not None for o in out)
    for o in out:
        assert abs(o[0] - 0.3) < 1e-12 and o[1] == 512 and o[2] == 2.0 and o[3] == 512 and o[4] == "strong"


def test_thread_ranks_abort_releases_the_others():
    import threading
    sys.path.insert(0, ROOT)
    from pipe_amd import shard
    root = shard.ThreadSync(2)
    seen = []

    def waiter():
        try:
            root.for_rank(0).barrier()
        except threading.BrokenBarrierError:
            seen.append("released")

    t = threading.Thread(target=waiter)
    t.start()
    time.sleep(0.05)
    root.abort()
    t.join(timeout=10)
    assert seen == ["released"]


def test_process_sync_falls_back_to_host_tensors_when_the_device_collective_fails():
    """bench.py's ranks try ONE RCCL all-reduce; if it fails anywhere, every rank agrees (through gloo) to run
    its barriers and reductions on host tensors.  Here: no GPU, so the device tensor cannot even be made."""
    import torch
    from pipe_amd import shard

    class FakeDist:
        class ReduceOp:
            MIN, MAX, SUM = "min", "max", "sum"
        calls = []

        def all_reduce(self, t, op=None):
            assert t.device.type == "cpu"
            self.calls.append(op)

        def get_world_size(self):
            return 2

        def barrier(self):
            raise AssertionError("the device barrier must not be used after the fallback")

    if torch.cuda.is_available():
        return
    d = FakeDist()
    s = shard.ProcessSync(d, "cuda")
    s.settle()
    assert s.device == "cpu" and s.fallback
    s.barrier()
    assert s.max(3.0) == 3.0
    assert d.calls[0] == "min" and len(d.calls) == 3
