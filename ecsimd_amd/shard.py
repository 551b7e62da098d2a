"""Multi-GPU sharding of a batch: one process per GPU, contiguous index ranges, no collective on
the data path, ONE gather of each result shard to rank 0 (RCCL over xGMI when the tensors are on
GPUs; the same code runs over gloo on CPU tensors, which is how tests/test_dist_cpu.py covers it).

The reference has no multi-anything; this follows SURVEY.md 8(e): scalar
multiplications are independent units, rank r owns global indices [r*n, (r+1)*n) of the synthetic
input streams, and the gather of step i runs on a side stream while step i+1 computes.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous partition of [0, n_total): (first_index, count) of `rank`; the first
    n_total % world ranks take one extra unit."""
    base, extra = divmod(n_total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


class ShardedRunner:
    """Runs `compute(out)` once per step into one of two result buffers and gathers each finished
    buffer to rank 0, overlapped with the next step's compute.

    compute(out) must ENQUEUE its work on the current stream (GPU) or run synchronously (CPU).
    All ranks must use the same `shape`; rank 0 ends up with `gathered[r]` = rank r's last result.
    """

    def __init__(self, shape, dtype, device, world: int, rank: int, group=None, always_gather: bool = False):
        self.world, self.rank, self.group = world, rank, group
        self.dist = world > 1 or always_gather          # always_gather: rehearse the collective path with one rank
        self.cuda = torch.device(device).type == "cuda"
        self.outs = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        self.gathered = ([torch.empty(shape, dtype=dtype, device=device) for _ in range(world)]
                         if (self.dist and rank == 0) else None)
        self.comm = torch.cuda.Stream(device=device) if (self.cuda and self.dist) else None
        self._gather_done = [None, None]      # per buffer: event after its last gather (GPU only)
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
        if self.dist:
            if self.cuda:
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(self.comm):
                    self.comm.wait_event(ready)
                    dist.gather(out, self.gathered, dst=0, group=self.group)
                    fin = torch.cuda.Event()
                    fin.record()
                self._gather_done[i] = fin
            else:
                dist.gather(out, self.gathered, dst=0, group=self.group)
        self.steps += 1
        return out

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
