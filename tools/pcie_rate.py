#!/usr/bin/env python3
"""PCIe-inclusive rate of the ladder: scalars/points start in (pinned) host memory, results end there.
H2D, compute and D2H of consecutive batches overlap on three streams.  Never bench.py's `value`
(that one has inputs resident in HBM); recorded in DESIGN.md section 4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from ecsimd_amd import Engine, P256
e = Engine(0); n = 1 << 22; SEED = 0x5EEDEC51D0000001; batches = 6
k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
bx, by = e.scalar_mult_base(P256, s, flags=2)
hk, hx, hy = (t.cpu().pin_memory() for t in (k, bx, by))
hout = [torch.empty((3, n, 4), dtype=torch.int64).pin_memory() for _ in range(2)]
dk = [torch.empty_like(k) for _ in range(2)]; dx = [torch.empty_like(k) for _ in range(2)]; dy = [torch.empty_like(k) for _ in range(2)]
dout = [torch.empty((3, n, 4), dtype=torch.int64, device=e.tdev) for _ in range(2)]
up, comp, down = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def run(nb):
    ev_up = [None, None]; ev_comp = [None, None]; ev_down = [None, None]
    for b in range(nb):
        i = b & 1
        with torch.cuda.stream(up):
            if ev_comp[i] is not None: up.wait_event(ev_comp[i])          # inputs of batch b-2 consumed
            dk[i].copy_(hk, non_blocking=True); dx[i].copy_(hx, non_blocking=True); dy[i].copy_(hy, non_blocking=True)
            ev_up[i] = torch.cuda.Event(); ev_up[i].record()
        with torch.cuda.stream(comp):
            comp.wait_event(ev_up[i])
            if ev_down[i] is not None: comp.wait_event(ev_down[i])        # output buffer drained
            e.scalar_mult(P256, dk[i], dx[i], dy[i], flags=0, out=[dout[i][0], dout[i][1], dout[i][2]])   # classical in -> from_affine inside
            ev_comp[i] = torch.cuda.Event(); ev_comp[i].record()
        with torch.cuda.stream(down):
            down.wait_event(ev_comp[i])
            hout[i].copy_(dout[i], non_blocking=True)
            ev_down[i] = torch.cuda.Event(); ev_down[i].record()
    torch.cuda.synchronize()
run(2)
t = time.perf_counter(); run(batches); dt = time.perf_counter() - t
h2d = 3 * n * 32; d2h = 3 * n * 32
print(f"PCIe-inclusive: {batches * n / dt / 1e6:.2f} M scalar mults/s ({dt / batches * 1e3:.1f} ms per 2^22 batch; {h2d/1e6:.0f} MB up + {d2h/1e6:.0f} MB down per batch)")

# the C ABI's own host-array entry point (ecsimd_hip_scalar_mult_host, r5): PAGEABLE numpy arrays in and out, chunks of 2^19 on two contexts
import numpy as np
kn, xn, yn = (np.ascontiguousarray(e.to_numpy(t)) for t in (k, bx, by))
out = e.scalar_mult_host(P256, kn, xn, yn)                     # warm-up: the helper context, the staging, and the output arrays' pages (a caller reuses its buffers)
reps = 4
t = time.perf_counter()
for _ in range(reps):
    e.scalar_mult_host(P256, kn, xn, yn, out=out)
dt = (time.perf_counter() - t) / reps
print(f"ecsimd_hip_scalar_mult_host, pageable arrays: {n / dt / 1e6:.2f} M scalar mults/s ({dt * 1e3:.1f} ms per 2^22 batch, classical base in, Jacobian out)")
