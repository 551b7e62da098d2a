import os, sys, torch
sys.path.insert(0, os.getcwd())
from ecsimd_amd import Engine, P256, OUT_AFFINE, ALG_WINDOWED_BIG
e = Engine(0)
SEED = 0x5EEDEC51D0000001
for log2n in (20, 22, 24):
    n = 1 << log2n
    k = e.fill_random(n, SEED, 1); s = e.fill_random(n, SEED, 2)
    bx, by = e.scalar_mult_base(P256, s, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    J = e.scalar_mult(P256, k, bx, by) if log2n <= 22 else e.scalar_mult(P256, k, bx, by)
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]
    ms = t(lambda: e.to_affine(P256, J))
    out = [e.empty(n) for _ in range(3)]
    msb = t(lambda: e.scalar_mult_base(P256, k, flags=OUT_AFFINE | ALG_WINDOWED_BIG, out=out))
    print(f"cap={os.environ.get('ECS_TA_CAP','32')} shift={os.environ.get('ECS_TA_SHIFT','17')} 2^{log2n}: to_affine {ms*1e3:8.1f} us = {n/ms/1e6:7.2f} G/s   fixed-base-big {n/msb/1e3:8.1f} M/s")
