import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ecsimd_amd import Engine, P256, SECP256K1, OUT_AFFINE
cv = P256 if (len(sys.argv) < 2 or sys.argv[1] == "p256") else SECP256K1
e = Engine(0); n = 1 << 22
s = e.fill_random(n, 11, 2)
Q = e.scalar_mult_base(cv, s, flags=OUT_AFFINE)
ee = e.fill_random(n, 12, 1); r = e.fill_random(n, 13, 1); ss = e.fill_random(n, 14, 1)
r = e.fill_random(n, 13, 1, clear_top_bits=1); ss = e.fill_random(n, 14, 1, clear_top_bits=1)
for _ in range(3):
    ok = e.ecdsa_verify(cv, ee, r, ss, Q[0], Q[1])
torch.cuda.synchronize()
print("done", int(ok.sum()) if hasattr(ok, "sum") else ok)
