#!/usr/bin/env python3
"""asm_clobber_check.py file.s ... -- a guard against one class of register-allocation defect in the inline-asm Comba columns.

field.cuh's column statement (`mac_col`) is: acc (64-bit, in/out) += a[i] * b[i] by v_mad_u64_u32, the carries counted into `ex`
by v_addc_co_u32, `ex` an EARLY-CLOBBER output.  Round 4 met a build (k_zdau_repeat<32>, P-256) in which the compiler gave `ex` the
register of a multiplicand that the SAME statement reads afterwards (and that the next column reads again) -- z.w[7] of z = z * zz --
and returned a wrong Z: an early-clobber output must never share a register with an input.  Nothing in the source allows it (the
constraint is "=&v"); it is a compiler defect (sub-register liveness of an early-clobber def that is later inserted into a 64-bit
register), and parity tests catch it only on the kernels they run.  This checker walks every inline-asm block of a `hipcc -S`
listing and refuses:
  * a v_mad_u64_u32 / v_mad_i64_i32 whose multiplicand (src0 / src1) register was written earlier IN THE SAME BLOCK by an
    instruction of that block -- multiplicands of these blocks are always inputs, never intermediate results;
  * an accumulating multiply-add (destination pair = addend pair) one of whose multiplicands lies inside that pair -- both are live
    into the instruction (the second shape the same defect took after the factors of the product were exchanged).
Exit code 1 and a listing of the offending blocks when anything is found."""
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    bad = []
    func = None
    block = None
    for no, ln in enumerate(open(path), 1):
        t = ln.strip()
        m = re.match(r"^([A-Za-z_][\w.$]*):", ln)
        if m and not ln.startswith(".L"):
            func = m.group(1)
        if t.startswith(";;#ASMSTART"):
            block = {"written": set(), "start": no}
            continue
        if t.startswith(";;#ASMEND"):
            block = None
            continue
        if block is None or not t.startswith("v_"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")] if " " in t else []
        name = t.split()[0]
        if name in ("v_mad_u64_u32", "v_mad_i64_i32") and len(ops) >= 5:
            for src in ops[2:4]:
                if regs(src) & block["written"]:
                    bad.append((path, func, no, t))
                # rule 2: an ACCUMULATING multiply-add (destination pair == addend pair) whose multiplicand lies inside that pair: the
                # multiplicand and the running sum are both live into the instruction and cannot share a register
                elif ops[0] == ops[4] and regs(src) & regs(ops[0]):
                    bad.append((path, func, no, t))
            block["written"] |= regs(ops[0])
        elif ops:
            block["written"] |= regs(ops[0])
    return bad


if __name__ == "__main__":
    found = []
    for p in sys.argv[1:]:
        found += check(p)
    for path, func, no, t in found:
        print(f"{path}:{no}: {func}: a multiplicand was overwritten inside its own asm block: {t}")
    print(f"asm_clobber_check: {len(found)} violation(s) in {len(sys.argv) - 1} file(s)")
    sys.exit(1 if found else 0)
