#!/usr/bin/env python3
"""Condense a rocprofv3 output directory written by tools/profile.sh or tools/profile_points.sh into
profiles/<round>/<tag>/{kernel_stats.csv, kernel_durations.json, pmc_summary.json}.

    tools/summarize_profiles.py <round> <tag> <gpurun_out/prof_dir> [--min-grid LANES] [--traffic-key KEY]

Per kernel and grid size (a kernel launched at several sizes -- e.g. the ladder at 2^24 for the bench and at 6.8 M for the
window-table build -- is reported per size, which rocprofv3's own --stats average mixes):
  * duration: calls / average / min / max from the kernel trace;
  * counters from the separate --pmc passes (the LAST dispatch of that kernel at that size);
  * HBM bytes per launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024: the gfx950 correction of
    /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 64-byte requests of 128-byte lines as 32-byte units);
  * VALU instructions per wave, VALU issue interval (SQ_BUSY_CYCLES-based), register use as rocprofv3 reports it.
rocprofv3's VGPR_Count column is the ALLOCATION of one work-item in the unified 512-entry file rounded to the 8-register
granule and counted in ... whatever unit it prints; the compiler's own figure (hipcc -Rpass-analysis=kernel-resource-usage)
is recorded next to it when the caller passes --vgprs NAME=N."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil


def newest(pattern):
    best = {}
    for f in glob.glob(pattern):
        key = os.path.dirname(f)
        if key not in best or os.path.getmtime(f) > os.path.getmtime(best[key]):
            best[key] = f
    return sorted(best.values())


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    base = (m.group(1) + (m.group(2) or "")) if m else name[:60]
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round"); ap.add_argument("tag"); ap.add_argument("src")
    ap.add_argument("--min-grid", type=int, default=1 << 17)
    ap.add_argument("--traffic-key", default=None, help="write the HBM bytes of this tag's kernels into profiles/pmc_traffic.json under this key")
    ap.add_argument("--traffic-kernels", default=None, help="comma-separated kernel-name filters for --traffic-key (default: all)")
    ap.add_argument("--vgprs", action="append", default=[], help="NAME=N: the compiler's VGPR count for kernels whose name contains NAME")
    ap.add_argument("--about", default="")
    ap.add_argument("--curves", default="", help="comma-separated labels of the same-named kernel instances in dispatch order, e.g. p256,secp256k1")
    a = ap.parse_args()
    dst = os.path.join("profiles", a.round, a.tag)
    os.makedirs(dst, exist_ok=True)
    for f in newest(os.path.join(a.src, "stats", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
    dur = collections.defaultdict(list)
    # The per-curve translation units define kernels of the SAME name (k_zdau, k_trplu, ...): they differ by kernel id.
    # Instances are numbered in order of first dispatch -- the drivers run P-256 first, secp256k1 second.
    def instance_namer():
        seen = collections.defaultdict(list)
        def name(kernel_name, kernel_id):
            k = short(kernel_name)
            if kernel_id not in seen[k]:
                seen[k].append(kernel_id)
            i = seen[k].index(kernel_id)
            return k if (i == 0 and not a.curves) else f"{k} [{(a.curves.split(',') + ['instance %d' % (i + 1)] * 8)[i] if a.curves else 'instance %d' % (i + 1)}]"
        return name
    namer = instance_namer()
    for f in newest(os.path.join(a.src, "stats", "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            g = int(r["Grid_Size_X"])
            if g >= a.min_grid and "ecsimd_hip" in r["Kernel_Name"]:
                dur[(namer(r["Kernel_Name"], r["Kernel_Id"]), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    durations = {f"{k} @ {g} lanes": {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)} for (k, g), v in sorted(dur.items())}
    json.dump(durations, open(os.path.join(dst, "kernel_durations.json"), "w"), indent=1)
    out = {"_about": a.about or f"rocprofv3 --pmc passes ({a.src}), one pass per counter group; per launch, last dispatch of each kernel at each grid size >= {a.min_grid} lanes",
           "kernels": {}}
    for f in newest(os.path.join(a.src, "pmc_*", "*", "*_counter_collection.csv")):
        namer = instance_namer()                      # kernel ids are per process: number the instances again in each pass
        for r in csv.DictReader(open(f)):
            g = int(r["Grid_Size"])
            if g < a.min_grid or "ecsimd_hip" not in r["Kernel_Name"]:
                continue
            key = f"{namer(r['Kernel_Name'], r['Kernel_Id'])} @ {g} lanes"
            kk = out["kernels"].setdefault(key, {"counters": {}})
            kk["counters"][r["Counter_Name"]] = float(r["Counter_Value"])
            kk["grid_size"] = g
            kk["rocprofv3_columns"] = {"VGPR_Count": int(r["VGPR_Count"]), "Accum_VGPR_Count": int(r["Accum_VGPR_Count"]), "SGPR_Count": int(r["SGPR_Count"]),
                                       "LDS_Block_Size": int(r["LDS_Block_Size"]), "Scratch_Size": int(r["Scratch_Size"])}
    total = 0.0
    filt = a.traffic_kernels.split(",") if a.traffic_kernels else None
    for key, kk in out["kernels"].items():
        c = kk["counters"]
        for spec in a.vgprs:
            nm, n = spec.split("=")
            if nm in key:
                kk["compiler_vgprs"] = int(n)
        if key in durations:
            kk["duration_us"] = durations[key]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            kk["hbm_bytes_per_launch"] = {"fetch_corrected_x2": c["FETCH_SIZE"] * 1024 * 2, "write": c["WRITE_SIZE"] * 1024, "total": c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024}
            if key in durations:
                kk["hbm_GBps"] = kk["hbm_bytes_per_launch"]["total"] / (durations[key]["avg_us"] * 1e-6) / 1e9
            if filt is None or any(x in key for x in filt):
                total += kk["hbm_bytes_per_launch"]["total"]
        if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
            kk["valu_wave_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "SQ_INSTS_VALU" in c and c.get("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES sums over the 8 XCDs x shader engines; per-SIMD issue interval = busy cycles per SIMD / instructions per SIMD
            kk["valu_instructions_per_simd"] = c["SQ_INSTS_VALU"] / 1024.0
        if "SQ_ACTIVE_INST_VALU" in c and c.get("SQ_BUSY_CYCLES"):
            kk["valu_active_over_busy"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
        if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"):
            kk["wait_inst_any_over_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        if "GRBM_GUI_ACTIVE" in c:
            kk["cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8
            if c.get("SQ_INSTS_VALU"):
                kk["cycles_per_valu_instruction_per_simd"] = (c["GRBM_GUI_ACTIVE"] / 8) / (c["SQ_INSTS_VALU"] / 1024.0)
    json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    if a.traffic_key:
        tpath = os.path.join("profiles", "pmc_traffic.json")
        table = json.load(open(tpath)) if os.path.exists(tpath) else {}
        table[a.traffic_key] = total
        table["_source"] = "profiles/r*/<tag>/pmc_summary.json (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 per launch; tools/summarize_profiles.py --traffic-key)"
        json.dump(table, open(tpath, "w"), indent=1)
    for key, kk in sorted(out["kernels"].items()):
        print(f"{key:58s} {kk.get('duration_us', {}).get('avg_us', 0):10.1f} us  valu/wave {kk.get('valu_wave_instructions_per_wave', 0):9.0f}  "
              f"hbm {kk.get('hbm_bytes_per_launch', {}).get('total', 0) / 1e6:9.1f} MB {kk.get('hbm_GBps', 0):7.0f} GB/s  "
              f"cycles/VALU instr/SIMD {kk.get('cycles_per_valu_instruction_per_simd', 0):.2f}")


if __name__ == "__main__":
    main()
