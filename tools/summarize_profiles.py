#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>_{ladder,fixed} (tools/profile.sh output) into profiles/<round>/."""
import collections, csv, glob, json, os, re, shutil, sys


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep the most recently written run per directory
    (the numeric prefix is a process id, not an order)."""
    best = {}
    for f in glob.glob(pattern):
        key = os.path.dirname(f)
        if key not in best or os.path.getmtime(f) > os.path.getmtime(best[key]):
            best[key] = f
    return sorted(best.values())


rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
traffic = {}
for tag, kern_filter, label in (("ladder", ["k_scalar_mult"], "k_scalar_mult_p256_2^22"), ("fixed", ["k_base_windowed", "k_to_affine_batched"], "fixed_base_p256_2^22"),
                               ("fixedbig", ["k_base_windowed_g", "k_to_affine_batched"], "fixed_base_big_p256_2^22"),
                               ("varwin", ["k_varwin_mult_odd", "k_varwin_odd_multiples", "k_varwin_to_table", "k_to_affine_batched"], "varwin_p256_2^22")):
    src = f"gpurun_out/prof_{rnd}_{tag}"
    if not os.path.isdir(src):
        continue
    os.makedirs(f"profiles/{rnd}/{tag}", exist_ok=True)
    for f in newest(f"{src}/stats/runc/*_kernel_stats.csv"):
        shutil.copy(f, f"profiles/{rnd}/{tag}/kernel_stats.csv")
    out = {"_about": f"rocprofv3 --pmc passes (tools/profile.sh {rnd}_{tag}) for bench.py --steps 2 --warmup 0 --no-cpu-baseline"
                     + {"fixed": " --workload fixed-base", "fixedbig": " --workload fixed-base-big", "varwin": " --workload windowed"}.get(tag, "") + " on MI355X; per launch, last 2^22-lane dispatch of each kernel", "kernels": {}}
    for d in newest(f"{src}/pmc_*/runc/*_counter_collection.csv"):
        for r in csv.DictReader(open(d)):
            for kf in kern_filter:
                if kf in r["Kernel_Name"] and int(r["Grid_Size"]) >= (1 << 17):
                    kk = out["kernels"].setdefault(kf, {"counters": {}})
                    kk["counters"][r["Counter_Name"]] = float(r["Counter_Value"])
                    kk["vgpr_count"] = int(r["VGPR_Count"]); kk["grid_size"] = int(r["Grid_Size"]); kk["lds_block_size"] = int(r["LDS_Block_Size"])
    tot = 0.0
    for kf, kk in out["kernels"].items():
        c = kk["counters"]
        kk["hbm_bytes_per_launch"] = {"fetch_corrected_x2": c["FETCH_SIZE"] * 1024 * 2, "write": c["WRITE_SIZE"] * 1024}
        kk["valu_wave_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "GRBM_GUI_ACTIVE" in c:
            kk["cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8
        tot += c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_step_total"] = tot
    traffic[label] = tot
    json.dump(out, open(f"profiles/{rnd}/{tag}/pmc_summary.json", "w"), indent=1)
    print(tag, {k: (round(v["valu_wave_instructions_per_wave"]), v["vgpr_count"]) for k, v in out["kernels"].items()}, "hbm bytes/step", tot)
traffic["_source"] = f"profiles/{rnd}/{{ladder,fixed,fixedbig,varwin}}/pmc_summary.json (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, per bench step)"
json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
