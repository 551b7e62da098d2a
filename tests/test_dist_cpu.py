"""CPU suite, part 3: the N > 1 path (ecsimd_amd/shard.py) over gloo, world_size 2.

Each rank regenerates ITS slice of the synthetic streams from (seed, global index), computes the
scalar multiplications of that slice (with the CPU oracle standing in for the kernel -- this is a
test), and the double-buffered runner gathers every step's shard to rank 0; rank 0 must end up with
exactly what a single process computes for the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ecsimd_amd.shard import ShardedRunner, shard_range, plan  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 1 << 22, (1 << 24) + 5):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_strong_and_weak_plans_cover_the_global_index_space():
    """bench.py --scaling strong / weak: every global index is owned by exactly one rank, BASELINE configs[3]
    (2^24 over 8) gives 2^21 per GPU, and all ranks allocate the same number of result rows (a gather moves equal pieces)."""
    assert [plan("strong", 1 << 24, r, 8)[:2] for r in (0, 3, 7)] == [(0, 1 << 21), (3 << 21, 1 << 21), (7 << 21, 1 << 21)]
    assert plan("strong", 1 << 24, 0, 1) == (0, 1 << 24, 1 << 24, 1 << 24)
    for scaling, units in (("strong", 1 << 24), ("strong", 1000003), ("strong", 7), ("weak", 1 << 22), ("weak", 5)):
        for world in (1, 2, 3, 4, 6, 8):
            plans = [plan(scaling, units, r, world) for r in range(world)]
            total = plans[0][2]
            assert total == (units if scaling == "strong" else units * world) and all(p[2] == total for p in plans)
            assert len({p[3] for p in plans}) == 1 and all(p[1] <= p[3] for p in plans)
            owned = sorted((p[0], p[0] + p[1]) for p in plans)
            assert owned[0][0] == 0 and owned[-1][1] == total and all(a[1] == b[0] for a, b in zip(owned, owned[1:]))
    with pytest.raises(ValueError):
        plan("medium", 8, 0, 1)


def _worker_uneven(rank, world, port, total, q):
    """Strong scaling with a total the ranks do not divide: shorter shards leave their last buffer row unused."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count, tot, rows = plan("strong", total, rank, world)
        runner = ShardedRunner((rows, 4), torch.int64, "cpu", world, rank)
        for step in range(3):
            def compute(out):
                out.zero_()
                out[:count] = (torch.arange(first, first + count, dtype=torch.int64)[:, None] * 4 + torch.arange(4)) + 1000 * step
            runner.step(compute)
        runner.gather = False                        # compute-only steps must not disturb what rank 0 holds
        runner.step(lambda out: out.fill_(-1))
        runner.fence()
        if rank == 0:
            assert runner.received.shape == (world, rows, 4) and runner.gathered[1].data_ptr() == runner.received[1].data_ptr()
            q.put(torch.cat([runner.gathered[r][:plan("strong", total, r, world)[1]] for r in range(world)]).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_strong_scaling_gather_with_uneven_shards():
    world, total = 2, 101
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = (np.arange(total, dtype=np.int64)[:, None] * 4 + np.arange(4)) + 2000
    assert np.array_equal(got, exp)


def _worker(rank, world, port, n_per_rank, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import fill_random_np, SEED, P256, CURVE_PARAMS, ints_to_arr
        from oracle.loader import Oracle
        ora = Oracle()
        first, count = shard_range(n_per_rank * world, rank, world)
        assert (first, count) == (rank * n_per_rank, n_per_rank)
        c = CURVE_PARAMS[P256]
        gx, gy = ints_to_arr([c["gx"]] * count), ints_to_arr([c["gy"]] * count)
        runner = ShardedRunner((3, count, 4), torch.int64, "cpu", world, rank, via_host=True)   # host staging is a GPU-rehearsal detail: ignored for CPU tensors
        for s in range(steps):
            k = fill_random_np(count, SEED, 1 + s, first_index=first)          # a different stream per step

            def compute(out):
                J = ora.scalar_mult(P256, k, gx, gy)
                out.copy_(torch.from_numpy(np.stack(J).view(np.int64)))
            runner.step(compute)
        runner.fence()
        if rank == 0:
            got = np.concatenate([g.numpy().view(np.uint64) for g in runner.gathered], axis=1)   # (3, world*count, 4)
            q.put(got)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_single_process():
    from helpers import fill_random_np, SEED, P256, CURVE_PARAMS, ints_to_arr
    from oracle.loader import Oracle
    world, n_per_rank, steps = 2, 48, 3
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_per_rank, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole batch, last step's stream
    n = world * n_per_rank; c = CURVE_PARAMS[P256]
    k = fill_random_np(n, SEED, steps, first_index=0)
    exp = np.stack(Oracle().scalar_mult(P256, k, ints_to_arr([c["gx"]] * n), ints_to_arr([c["gy"]] * n), threads=4))
    assert got.shape == exp.shape and np.array_equal(got, exp)
