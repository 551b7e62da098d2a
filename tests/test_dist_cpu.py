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

from ecsimd_amd.shard import ShardedRunner, shard_range  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 1 << 22, (1 << 24) + 5):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


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
        runner = ShardedRunner((3, count, 4), torch.int64, "cpu", world, rank)
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
