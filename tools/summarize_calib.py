#!/usr/bin/env python3
"""Condense gpurun_out/calib_<tag>/ (tools/profile_calib.sh) into profiles/<round>/hbm_counter_calibration.json: per access
pattern of tools/ubench/hbm_gather_calib.hip the bytes the kernel REQUESTED (each exactly once), its rate, what rocprofv3's
FETCH_SIZE / TCC_EA0_RDREQ* read for that dispatch, the bytes that MOVED (32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B), the
factor  moved bytes / (FETCH_SIZE x 1024)  that tools/summarize_traffic.py applies, and the over-fetch  moved / requested.

    tools/summarize_calib.py <round> <gpurun_out/calib_dir>
"""
import csv
import glob
import json
import os
import sys


def main():
    rnd, src = sys.argv[1], sys.argv[2]
    plain = [json.loads(ln) for ln in open(os.path.join(src, "plain.jsonl")) if ln.startswith("{")]
    out = {"_about": "tools/ubench/hbm_gather_calib.hip on one MI355X: every kernel requests a known number of bytes, each byte exactly once "
                     "(2 GiB buffer, 8x the Infinity Cache, caches flushed by a 2 GiB stream before each pattern); counters from separate rocprofv3 --pmc passes, "
                     "the LAST dispatch of each kernel.  moved_bytes = 32 x TCC_EA0_RDREQ_32B + 64 x _64B + 128 x _128B (the L2's memory-side read requests by size); "
                     "factor_fetch_to_moved = moved_bytes / (FETCH_SIZE x 1024): what FETCH_SIZE has to be multiplied by; overfetch = moved_bytes / requested bytes.  "
                     "Finding: on gfx950 EVERY memory-side read request of these patterns is a 128-byte one and FETCH_SIZE tallies each at 64 bytes -- so FETCH_SIZE x 2 is the "
                     "traffic for streams AND gathers, and a random 64-byte read really moves a 128-byte line (overfetch 2.0; calib_gather64 runs at the time of a 2 GiB stream).",
           "patterns": {}}
    for p in plain:
        out["patterns"][p["kernel"]] = {"requested_bytes": p["requested_bytes_per_launch"], "ms": p["ms_per_launch"], "requested_GBps": p["requested_GBps"], "counters": {}}
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        last = {}
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if name in out["patterns"]:
                last[(name, r["Counter_Name"])] = (int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"]))
        for (name, cname), (_, val, grid) in last.items():
            out["patterns"][name]["counters"][cname] = val
            out["patterns"][name]["grid_of_the_counted_dispatch"] = grid
    for name, p in out["patterns"].items():
        c = p["counters"]
        # calib_stream16 is also the cache-flush kernel: its last dispatch is a 2 GiB pass either way
        if "FETCH_SIZE" in c and c["FETCH_SIZE"] > 0:
            p["fetch_size_bytes"] = c["FETCH_SIZE"] * 1024
            p["factor_fetch"] = p["requested_bytes"] / p["fetch_size_bytes"]
        if "TCC_EA0_RDREQ_sum" in c:
            p["rdreq"] = c["TCC_EA0_RDREQ_sum"]
            p["requested_bytes_per_rdreq"] = p["requested_bytes"] / c["TCC_EA0_RDREQ_sum"] if c["TCC_EA0_RDREQ_sum"] else None
            if "TCC_EA0_RDREQ_32B_sum" in c:
                p["rdreq_32B_share"] = c["TCC_EA0_RDREQ_32B_sum"] / c["TCC_EA0_RDREQ_sum"] if c["TCC_EA0_RDREQ_sum"] else None
        if "TCC_EA0_RDREQ_128B_sum" in c:
            p["moved_bytes"] = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c["TCC_EA0_RDREQ_128B_sum"]
            p["overfetch"] = p["moved_bytes"] / p["requested_bytes"]
            p["moved_GBps"] = p["moved_bytes"] / (p["ms"] * 1e-3) / 1e9
            if p.get("fetch_size_bytes"):
                p["factor_fetch_to_moved"] = p["moved_bytes"] / p["fetch_size_bytes"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) > 0:
            p["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    dst = os.path.join("profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    json.dump(out, open(os.path.join(dst, "hbm_counter_calibration.json"), "w"), indent=1)
    print(f"{'pattern':20s} {'requested MB':>12s} {'req GB/s':>8s} {'FETCH_SIZE MB':>14s} {'moved MB':>9s} {'moved GB/s':>10s} {'moved/FETCH':>11s} {'overfetch':>9s} {'L2 hit':>7s}")
    for name, p in out["patterns"].items():
        print(f"{name:20s} {p['requested_bytes'] / 1e6:12.1f} {p['requested_GBps']:8.0f} {p.get('fetch_size_bytes', 0) / 1e6:14.1f} {p.get('moved_bytes', 0) / 1e6:9.1f} "
              f"{p.get('moved_GBps', 0):10.0f} {p.get('factor_fetch_to_moved', 0):11.3f} {p.get('overfetch', 0):9.3f} {p.get('l2_hit_rate', 0):7.3f}")


if __name__ == "__main__":
    main()
