#!/usr/bin/env python3
"""A/B of kernel variants on the GPU box: every variant is another build of the SAME library (make ... EXTRA=-D...) selected
with ECSIMD_HIP_LIBRARY; each runs `bench.py <args>` in a child process and this prints value / kernel time per variant.

    tools/ab_variants.py "<bench args>" name=path [name=path ...]        (path: a libecsimd_hip.so; `base` = the in-tree one)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1].split()
for spec in sys.argv[2:]:
    name, path = spec.split("=", 1)
    env = dict(os.environ)
    if path != "base":
        env["ECSIMD_HIP_LIBRARY"] = os.path.join(ROOT, path)
    vals = []
    for rep in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args, capture_output=True, text=True, env=env)
        try:
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            vals.append((d["value"] / 1e6, d["roofline"]["kernel_ms"]))
        except (IndexError, ValueError):
            vals.append((float("nan"), float("nan"))); print(r.stderr[-400:])
    print(f"{name:28s} " + "   ".join(f"{v:8.3f} M/s ({ms:8.3f} ms)" for v, ms in vals), flush=True)
