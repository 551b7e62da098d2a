import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ecsimd_amd import Engine, SECP256K1, OUT_AFFINE, ALG_WINDOWED
e = Engine(0); n = 1 << 12; cv = SECP256K1
k = e.fill_random(n, 5, 1); s = e.fill_random(n, 5, 2)
bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE)
lx, ly = e.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE)
wx, wy = e.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
dx = (wx != lx).any(dim=1); dy = (wy != ly).any(dim=1)
print("lanes", n, "x differ", int(dx.sum()), "y differ", int(dy.sum()), "first", torch.nonzero(dx | dy)[:8].flatten().tolist())
