#!/usr/bin/env python3
"""Competitor timing: OpenSSL's EC_POINT_mul (variable base) on the host cores -- the restatement of the
reference's benchs/p256_ref.cpp:55-91 (bench_openssl) over a batch.  TEST INFRASTRUCTURE ONLY; bench.py's
cpu_baseline leg runs this file as a child process (so that the workers can be forked: libcrypto 3.0's
threads serialise on its library-context locks, processes do not) and reads the one JSON line it prints.

  python oracle/ossl_bench.py --curve 0 --procs 16 --seconds 3
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loader  # noqa: E402


def _worker(args):
    curve, seed, seconds = args
    lib = loader.OpenSSLCheck()
    rng = np.random.default_rng(seed)
    m = 512
    bx, by, inf = lib.scalar_mult_base(curve, rng.integers(0, 2**64, size=(m, 4), dtype=np.uint64))   # valid lane-distinct points
    assert not inf.any()
    k = rng.integers(0, 2**64, size=(m, 4), dtype=np.uint64)
    done, spent = 0, 0.0
    while spent < seconds:
        spent += lib.time_scalar_mult(curve, k, bx, by, threads=1)
        done += m
    return done, spent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", type=int, default=0)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--seconds", type=float, default=3.0)
    a = ap.parse_args()
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(a.procs) as pool:
        res = pool.map(_worker, [(a.curve, 1000 + i, a.seconds) for i in range(a.procs)])
    wall = time.perf_counter() - t0
    rate = sum(d / s for d, s in res)                     # workers run concurrently for the same nominal time
    print(json.dumps({"value": rate, "unit": "scalar_mults/s", "procs": a.procs, "per_proc": rate / a.procs,
                      "library": loader.OpenSSLCheck().version(), "wall_s": wall,
                      "sample": f"EC_POINT_mul(group, R, NULL, P, k) on lane-distinct (k, P), {a.procs} processes x {a.seconds:.0f} s"}))


if __name__ == "__main__":
    main()
