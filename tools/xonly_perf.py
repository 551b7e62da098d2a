import sys, torch
sys.path.insert(0, '.')
from ecsimd_amd import Engine, P256, OUT_AFFINE
e = Engine(0); n = 1 << 22
k = e.fill_random(n, 1, 1); s = e.fill_random(n, 1, 2)
bx, by = e.scalar_mult_base(P256, s, flags=OUT_AFFINE)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
full = t(lambda: e.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE))
xo = t(lambda: e.scalar_mult(P256, k, bx, by, flags=OUT_AFFINE, x_only=True))
print(f"ladder affine (x, y): {full:.2f} ms = {n/full/1e3:.2f} M/s;  x only: {xo:.2f} ms = {n/xo/1e3:.2f} M/s")
