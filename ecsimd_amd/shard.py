"""Multi-GPU sharding of a batch: one process per GPU, contiguous index ranges, no collective on
the data path, ONE gather of each result shard to rank 0 (RCCL over xGMI when the tensors are on
GPUs; the same code runs over gloo on CPU tensors, which is how tests/test_dist_cpu.py covers it).

The reference has no multi-anything; this follows SURVEY.md 8(e): scalar
multiplications are independent units, rank r owns a contiguous range of the global indices of the
synthetic input streams, and the gather of step i runs on a side stream while step i+1 computes.

Two partitions of the global index space (bench.py `--scaling`):
  weak    every rank owns n units: rank r = [r*n, (r+1)*n)                      (per-GPU work fixed)
  strong  N units in total, rank r owns shard_range(N, r, world)                 (BASELINE configs[3]:
          2^24 over 8 GPUs = 2^21 each; the first N % world ranks take one unit more)
"""
from __future__ import annotations

import collections

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous partition of [0, n_total): (first_index, count) of `rank`; the first
    n_total % world ranks take one extra unit."""
    base, extra = divmod(n_total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def plan(scaling: str, units: int, rank: int, world: int):
    """(first global index, units of this rank, units of the whole job, rows of every rank's result buffer).
    `units` is per rank for "weak" and the global total for "strong".  All ranks allocate the same number of
    rows (the largest shard) because a gather moves equal-sized pieces; a shorter shard leaves its last row unused."""
    if scaling == "weak":
        return rank * units, units, units * world, units
    if scaling != "strong":
        raise ValueError(f"unknown scaling mode {scaling!r}")
    first, count = shard_range(units, rank, world)
    return first, count, units, shard_range(units, 0, world)[1]


class ShardedRunner:
    """Runs `compute(out)` once per step into one of two result buffers and gathers each finished
    buffer to rank 0, overlapped with the next step's compute.

    compute(out) must ENQUEUE its work on the current stream (GPU) or run synchronously (CPU).
    All ranks must use the same `shape`.  Rank 0 owns ONE pre-sized receive buffer `received`
    of shape (world, *shape) -- `gathered[r]` is its r-th slice, rank r's last result -- so a step's
    gather allocates nothing.  `gather=False` skips the collective (compute-only timing).
    """

    def __init__(self, shape, dtype, device, world: int, rank: int, group=None, always_gather: bool = False, time_gathers: bool = True,
                 keep_timings: int = 1024, via_host: bool = False):
        self.world, self.rank, self.group = world, rank, group
        self.dist = world > 1 or always_gather          # always_gather: rehearse the collective path with one rank
        self.cuda = torch.device(device).type == "cuda"
        # via_host: REHEARSAL ONLY (bench.py, ECSIMD_BENCH_REHEARSE_ONE_GPU=1).  RCCL refuses two ranks on one device, so N ranks
        # sharing ONE GPU gather over gloo instead: the finished buffer goes to pinned host memory on the side stream, the host waits
        # for that copy (no overlap with the next step) and gloo gathers host tensors.  Results still come from the HIP kernels.
        self.via_host = bool(via_host and self.cuda and self.dist)
        self.outs = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        self.host_outs = [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(2)] if self.via_host else None
        self.received = torch.empty((world,) + tuple(shape), dtype=dtype, device=("cpu" if self.via_host else device)) if (self.dist and rank == 0) else None
        self.gathered = [self.received[r] for r in range(world)] if self.received is not None else None
        self.comm = torch.cuda.Stream(device=device) if (self.cuda and self.dist) else None
        self._gather_done = [None, None]      # per buffer: event after its last gather (GPU only)
        self._gather_events = collections.deque(maxlen=keep_timings)   # (start, end) of the last gathers, on the side stream: bounded,
        self.time_gathers = time_gathers                               # a long-running runner must not pile up HIP events
        self.gather = True
        self.steps = 0

    def step(self, compute, before=None, after=None):
        i = self.steps & 1
        out = self.outs[i]
        if self.cuda and self._gather_done[i] is not None:
            torch.cuda.current_stream().wait_event(self._gather_done[i])     # buffer still being sent?
        if before is not None:
            before()
        compute(out)
        if after is not None:
            after()
        if self.dist and self.gather:
            if self.via_host:
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(ready)
                    self.host_outs[i].copy_(out, non_blocking=True)
                    fin = torch.cuda.Event(); fin.record()
                fin.synchronize()
                dist.gather(self.host_outs[i], self.gathered, dst=0, group=self.group)
            elif self.cuda:
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(ready)
                    if self.time_gathers:
                        t0 = torch.cuda.Event(enable_timing=True); t0.record()
                    dist.gather(out, self.gathered, dst=0, group=self.group)
                    fin = torch.cuda.Event(enable_timing=self.time_gathers); fin.record()
                self._gather_done[i] = fin
                if self.time_gathers:
                    self._gather_events.append((t0, fin))
            else:
                dist.gather(out, self.gathered, dst=0, group=self.group)
        self.steps += 1
        return out

    def gather_ms(self, last: int):
        """Side-stream durations (ms) of the last `last` gathers (GPU only; call after fence())."""
        return [a.elapsed_time(b) for a, b in list(self._gather_events)[-last:]] if (self.cuda and last and not self.via_host) else []

    def fence(self):
        """Everything enqueued so far has finished on every rank."""
        if self.cuda:
            torch.cuda.synchronize()
        if self.dist:
            dist.barrier(group=self.group)
            if self.cuda:
                torch.cuda.synchronize()

    def last_result(self):
        return self.outs[(self.steps - 1) & 1]
