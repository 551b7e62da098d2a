#!/usr/bin/env python3
"""Turn gpurun_out/traffic_<tag>/ (tools/profile_traffic.sh: one rocprofv3 --pmc FETCH_SIZE pass and one WRITE_SIZE pass per
bench workload and curve) into
    profiles/<round>/traffic/<workload>_<curve>.json    every kernel of ONE timed step with its counters and HBM bytes
    profiles/pmc_traffic.json                            HBM bytes per step, the figure bench.py puts into roofline.traffic

Bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 for EVERY kernel.  The x 2 is not an assumption any more: the calibration
(profiles/<round>/hbm_counter_calibration.json, tools/ubench/hbm_gather_calib.hip) shows that every memory-side read request of
this chip is a 128-byte one that FETCH_SIZE tallies at 64 bytes -- for coalesced streams and for random 64-byte gathers alike,
so a gather kernel's figure includes the half lines it fetched and did not use (that over-fetch is real traffic).  The factor is
read from the calibration file (calib_stream32 and calib_gather64 must agree, or this script stops).
A step = the dispatches after the last step marker (a one-element k_fill_random launch, ECSIMD_BENCH_STEP_MARKER=1).
WRITE_SIZE is exact for 16-byte-per-lane streaming stores (guide, HBM section): every store here is one.

    tools/summarize_traffic.py <round> <gpurun_out/traffic_dir>
"""
import csv
import glob
import json
import os
import re
import sys

KEY = {"ladder": "k_scalar_mult", "ladder-ref-compat": "k_scalar_mult_refsqr", "ladder-x": "k_scalar_mult_x", "windowed": "varwin", "windowed-ct": "varwin_ct",
       "fixed-base": "fixed_base", "fixed-base-ct": "fixed_base_ct", "fixed-base-signed": "fixed_base_signed", "fixed-base-big": "fixed_base_big"}
# kernels that read table entries (64 bytes at a random or per-lane place): their figure contains the unused half lines
GATHERS = ("k_base_windowed_g", "k_varwin_mult_odd", "k_varwin_mult_glv")


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:48]


def rows(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
    out.sort()
    return out


def last_step(rs):
    marks = [i for i, r in enumerate(rs) if r[1] == "k_fill_random" and r[2] <= 256]
    if not marks:
        raise SystemExit("no step marker in the counter listing (ECSIMD_BENCH_STEP_MARKER=1?)")
    return [r for r in rs[marks[-1] + 1:] if r[1] != "k_peak_mad32" and not r[1].startswith("k_fill")]


def main():
    rnd, src = sys.argv[1], sys.argv[2]
    calib = json.load(open(os.path.join("profiles", rnd, "hbm_counter_calibration.json")))["patterns"]
    f_stream = calib["calib_stream32"]["factor_fetch_to_moved"]
    for pat in ("calib_gather64", "calib_own512", "calib_gather128"):
        if abs(calib[pat]["factor_fetch_to_moved"] - f_stream) > 0.02:
            raise SystemExit(f"the calibration gives {pat} another FETCH_SIZE factor than streams: price the gather kernels separately")
    tpath = os.path.join("profiles", "pmc_traffic.json")
    table = json.load(open(tpath)) if os.path.exists(tpath) else {}
    dst = os.path.join("profiles", rnd, "traffic")
    os.makedirs(dst, exist_ok=True)
    for d in sorted(glob.glob(os.path.join(src, "*_*"))):
        if not os.path.isdir(d):
            continue
        base = os.path.basename(d)
        wl, curve = base.rsplit("_", 1)
        if wl not in KEY:
            continue
        f = glob.glob(os.path.join(d, "pmc_FETCH_SIZE", "*", "*_counter_collection.csv"))
        w = glob.glob(os.path.join(d, "pmc_WRITE_SIZE", "*", "*_counter_collection.csv"))
        if not f or not w:
            print("incomplete:", base); continue
        fs, ws = last_step(rows(f[0])), last_step(rows(w[0]))
        if [r[1:3] for r in fs] != [r[1:3] for r in ws]:
            raise SystemExit(f"{base}: the two passes dispatched different kernel sequences")
        extra = {}                                        # r4: further counters of the same step, kernel by kernel (TRAFFIC_EXTRA_COUNTERS)
        for xd in sorted(glob.glob(os.path.join(d, "pmc_*"))):
            cname = os.path.basename(xd)[4:]
            xf = glob.glob(os.path.join(xd, "*", "*_counter_collection.csv"))
            if cname in ("FETCH_SIZE", "WRITE_SIZE") or not os.path.isdir(xd) or not xf:
                continue
            xs = last_step(rows(xf[0]))
            if [r[1:3] for r in xs] == [r[1:3] for r in fs]:
                extra[cname] = [r[3] for r in xs]
        kernels, total = [], 0.0
        for idx, ((did, name, grid, fv), (_, _, _, wv)) in enumerate(zip(fs, ws)):
            read = fv * 1024 * f_stream
            how = f"FETCH_SIZE x 1024 x {f_stream:.3f} (128-byte requests tallied at 64)" + ("; includes the unused halves of the lines its 64-byte table reads fetch" if name in GATHERS else "")
            kernels.append({"kernel": name, "grid": grid, "FETCH_SIZE_KB": fv, "WRITE_SIZE_KB": wv, "read_bytes": read, "write_bytes": wv * 1024, "pricing": how})
            for cname, vals in extra.items():
                kernels[-1][cname] = vals[idx]
            total += read + wv * 1024
        lanes = max(k["grid"] for k in kernels)
        json.dump({"_about": f"one timed step of `bench.py --workload {wl} --curve {curve}` (2^24 units) under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)",
                   "hbm_bytes_per_step": total, "kernels": kernels}, open(os.path.join(dst, base + ".json"), "w"), indent=1)
        table[f"{KEY[wl]}_{curve}_2^24"] = total
        print(f"{base:34s} {len(kernels):3d} kernels  {total / 1e6:10.1f} MB per step  ({total / (1 << 24):7.1f} B per unit)")
    table["_source"] = (f"profiles/{rnd}/traffic/ (keys not re-measured this round keep the earlier round's figure) -- profiles/r03/traffic/<workload>_<curve>.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel of ONE timed step (every kernel of the step, "
                        "workspace round trips included), FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 with the factor 2 established by profiles/r03/hbm_counter_calibration.json "
                        "(tools/summarize_traffic.py); the 2^22 keys are round 2's")
    json.dump(table, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
