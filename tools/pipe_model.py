#!/usr/bin/env python3
"""tools/pipe_model.py -- what the VALU pipe does during the dominant kernel of a bench workload: the numbers the bench line's
`roofline` object carries beside the algorithmic rate, written to profiles/pmc_pipe.json.

`roofline.frac` prices a field multiplication at SURVEY.md 8(d)'s 136 mad32 -- an ALGORITHMIC rate over the multiply peak, not a utilisation: since round 4
the ladder issues 101 multiply instructions per field multiplication, and the rest of its issue slots go to shifts, masks and limb adds.  This tool says
what the pipe really did, from two committed sources:

  * the rocprofv3 --pmc passes of the same bench command (profiles/rNN/<tag>/pmc_summary.json: SQ_INSTS_VALU, SQ_WAVES, GRBM_GUI_ACTIVE per launch), and
  * the shipped ISA of the kernel (build/csrc/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s, -save-temps of the library's own build): the instruction mix of its
    main loop, priced with the per-instruction issue costs MEASURED on this chip (profiles/r02/valu_issue_rates_gfx950.txt, profiles/r04/valu_issue_rates_r4_rows.txt,
    the 4-waves-per-SIMD column);

    valu_instructions_per_unit            wave-level VALU instructions per lane and launch (counter)
    multiply_instructions_per_unit        v_mad_* / v_mul_* among them (ISA: loop mix x trip count + the straight-line rest)
    cycles_per_valu_instruction_per_simd  SIMD cycles of the launch / VALU instructions issued on that SIMD (counter): ~4.0 = the pipe never idles
    issue_bound_frac                      sum over the instruction mix of (count x measured issue cost) / SIMD cycles of the launch: how much of the
                                          elapsed time the instruction mix ALONE accounts for at the measured per-class rates (the rest: dependency stalls,
                                          s_nop, instruction fetch)

    python tools/pipe_model.py r05            # rebuild profiles/pmc_pipe.json from profiles/r05/*/pmc_summary.json + build/csrc
    python tools/pipe_model.py --check        # the committed file still describes the ISA the build ships (tests/test_bench_contract.py)
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "pmc_pipe.json")
ISA = os.path.join(ROOT, "build", "csrc", "{unit}-hip-amdgcn-amd-amdhsa-gfx950.s")
SIMDS = 256 * 4
# bench workload key (as profiles/pmc_traffic.json) -> (translation unit, kernel name substring, trip count of the main loop, profile tag, kernel label in pmc_summary)
KERNELS = {
    "k_scalar_mult_p256_2^24": ("k_ladder_p256", "13k_scalar_multILi29E", 254, "ladder", "k_scalar_mult<29>"),
    "k_scalar_mult_secp256k1_2^24": ("k_ladder_secp256k1", "13k_scalar_multILi29E", 254, "ladder_secp256k1", "k_scalar_mult<29>"),
    "k_scalar_mult_refsqr_p256_2^24": ("k_ladder_p256_refsqr", "13k_scalar_multILi32E", 254, "ladder_ref_compat", "k_scalar_mult<32>"),
    "k_scalar_mult_refsqr_secp256k1_2^24": ("k_ladder_secp256k1_refsqr", "13k_scalar_multILi32E", 254, "ladder_ref_compat_secp256k1", "k_scalar_mult<32>"),
    "k_scalar_mult_brainpoolP256r1_2^24": ("k_gladder", "16k_gc_scalar_multILi29ELb0E", 254, "ladder_brainpoolP256r1", "k_gc_scalar_mult<29, false>"),
    # the window loop of a registered curve (k_gvarwin.hip; launched in chunks of 2^22 lanes: 63 windows per launch; the table kernel is 4 % of a chunk)
    "varwin_brainpoolP256r1_2^22": ("k_gvarwin", "10k_gvw_multILb0E", 63, "windowed_brainpoolP256r1", "k_gvw_mult<false>"),
    "varwin_ct_brainpoolP256r1_2^22": ("k_gvarwin", "10k_gvw_multILb1E", 63, "windowed_ct_brainpoolP256r1", "k_gvw_mult<true>"),
}


def issue_costs():
    """mnemonic -> measured issue cycles per wave64 instruction per SIMD at 4 waves per SIMD."""
    cost = {}
    for rel in ("profiles/r02/valu_issue_rates_gfx950.txt", "profiles/r04/valu_issue_rates_r4_rows.txt"):
        for ln in open(os.path.join(ROOT, rel)):
            m = re.match(r"^(?:r4 )?(v_\w+)(?:\(\w+\)| literal| sgpr)?\s+[\d.]+ T/s\s+[\d.]+ cyc\s+[\d.]+ T/s\s+[\d.]+ cyc\s+[\d.]+ T/s\s+([\d.]+) cyc", ln)
            if m and m.group(1) not in cost:
                cost[m.group(1)] = float(m.group(2))
    return cost


FULL_RATE = ("v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32", "v_bitop3_b32")


def price(mnemonic, costs):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mnemonic)
    if base in costs:
        return costs[base]
    return 2.35 if base in FULL_RATE else 4.3                 # the two classes of profiles/r02/valu_issue_rates_gfx950.txt


def kernel_mix(unit, want):
    """(loop Counter, rest Counter) of VALU mnemonics of kernel `want`: the biggest loop and everything outside it."""
    lines = open(ISA.format(unit=unit)).read().splitlines()
    body, on = [], False
    for ln in lines:
        m = re.match(r"^([A-Za-z_][\w.$]*):", ln)
        if m and not ln.startswith(".L"):
            on = want in m.group(1)
            continue
        if on:
            if ln.startswith(".Lfunc_end"):
                break
            body.append(ln)
    labels, insts = {}, []
    for ln in body:
        t = ln.strip()
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(insts)
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append(t.split(";")[0].strip())
    best = None
    for i, t in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i and (best is None or i + 1 - labels[m.group(1)] > best[1] - best[0]):
            best = (labels[m.group(1)], i + 1)
    if best is None:
        raise SystemExit(f"{unit}: no loop in {want}")
    valu = lambda seq: collections.Counter(t.split()[0] for t in seq if t.startswith("v_"))
    return valu(insts[best[0]:best[1]]), valu(insts[:best[0]] + insts[best[1]:])


def is_multiply(mn):
    return mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo", "v_mul_hi", "v_mad_u32_u24", "v_mul_u32_u24", "v_mul_i32_i24"))


def describe(key, round_tag):
    unit, want, trips, tag, label = KERNELS[key]
    loop, rest = kernel_mix(unit, want)
    costs = issue_costs()
    cyc = lambda c: sum(n * price(mn, costs) for mn, n in c.items())
    model_cycles_per_wave = trips * cyc(loop) + cyc(rest)
    static_valu_per_wave = trips * sum(loop.values()) + sum(rest.values())
    mult_per_wave = trips * sum(n for mn, n in loop.items() if is_multiply(mn)) + sum(n for mn, n in rest.items() if is_multiply(mn))
    out = {"kernel": label, "isa_unit": unit, "loop_trip_count": trips,
           "loop_valu_instructions": sum(loop.values()), "loop_multiply_instructions": sum(n for mn, n in loop.items() if is_multiply(mn)),
           "loop_mix": dict(loop.most_common()), "loop_model_cycles": cyc(loop),
           "static_valu_instructions_per_unit": static_valu_per_wave, "multiply_instructions_per_unit": mult_per_wave}
    pm = os.path.join(ROOT, "profiles", round_tag, tag, "pmc_summary.json")
    if os.path.exists(pm):
        ks = json.load(open(pm))["kernels"]
        lanes = int(key.split("^")[1])
        k = ks[f"{label} @ {1 << lanes} lanes"]
        waves = k["counters"]["SQ_WAVES"]
        simd_cycles = k["cycles_per_xcd"]                                   # GRBM_GUI_ACTIVE / 8: the launch's cycles as every SIMD sees them
        out.update({"source": os.path.relpath(pm, ROOT), "lanes_per_launch": 1 << lanes,
                    "valu_instructions_per_unit": k["valu_wave_instructions_per_wave"],
                    "cycles_per_valu_instruction_per_simd": k["cycles_per_valu_instruction_per_simd"],
                    "kernel_ms_at_profile": k["duration_us"]["avg_us"] / 1e3,
                    "issue_bound_frac": model_cycles_per_wave * waves / (SIMDS * simd_cycles),
                    "measured_cycles_per_wave_iteration": SIMDS * simd_cycles / waves / trips})
    return out


if __name__ == "__main__":
    if "--check" in sys.argv:
        old = json.load(open(OUT))
        bad = []
        for key, rec in old["kernels"].items():
            loop, rest = kernel_mix(*KERNELS[key][:2])
            if dict(loop.most_common()) != rec["loop_mix"]:
                bad.append(key)
        print("profiles/pmc_pipe.json", "describes the shipped ISA" if not bad else f"is STALE for {bad}: re-profile (tools/gpu_step.sh profile) and run tools/pipe_model.py <round>")
        sys.exit(1 if bad else 0)
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
    data = {"_about": "tools/pipe_model.py: what the VALU pipe does during the dominant kernel (counters of the committed rocprofv3 --pmc passes + the shipped ISA's "
                      "instruction mix priced at the measured per-instruction issue costs); read by bench.py into the roofline object",
            "kernels": {key: describe(key, rnd) for key in KERNELS}}
    json.dump(data, open(OUT, "w"), indent=1)
    for key, rec in data["kernels"].items():
        print(key, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items() if k != "loop_mix"})
