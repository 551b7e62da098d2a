#!/usr/bin/env python3
"""Constant-time guard on the SHIPPED ISA: reads `hipcc -S --cuda-device-only` output of a ladder translation unit and checks
the bit loop of a kernel (the 254 ZDAU iterations of curve_group.h:196-211) for anything whose timing could depend on the
scalar:

  * every branch in the loop is an s_cbranch_scc* whose SCC comes from a scalar compare of CLEAN scalars -- registers that
    inside the loop are only ever written by scalar instructions reading clean scalars or immediates (the iteration counter);
    a lane mask a VALU instruction wrote to SGPRs (carry-outs, v_cmp) is dirty, and so is VCC / EXEC: s_cbranch_vcc* /
    s_cbranch_exec* are refused outright;
  * no v_readfirstlane / v_readlane / v_permlane / ds_* / buffer_* / flat_* / stores to global memory / atomics;
  * the only global load is the scalar-word reload kwords[nb >> 5] (point.cuh ladder_core): its address registers are,
    walking back through the basic block, made of loop-invariant registers and clean scalars only;
  * scratch (spill) accesses use the constant `off` addressing form.

Usage: ct_check.py file.s kernel-substring        (prints a report; exit 1 on a violation)
"""
import re
import sys

SCC_WRITERS = re.compile(r"^s_(cmp|cmpk|bitcmp|add|sub|addc|subb|and|or|xor|andn2|orn2|nand|nor|xnor|lshl|lshr|ashr|bfe|bfm|mul_i32|min|max|abs|not|wqm|brev|"
                         r"and_saveexec|or_saveexec|xor_saveexec|andn2_saveexec|orn2_saveexec|cselect|absdiff|ff|flbit|bcnt|quadmask|sext)")
SCC_COMPARES = re.compile(r"^s_(cmp|cmpk|bitcmp)")


class Violation(AssertionError):
    pass


def regs_of(tok):
    """Scalar / vector registers named by one operand token: ('s', n) / ('v', n) tuples; 'vcc', 'exec', 'scc', 'm0' as names."""
    tok = tok.strip()
    out = []
    m = re.fullmatch(r"([sv])\[(\d+):(\d+)\]", tok)
    if m:
        return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.fullmatch(r"([sv])(\d+)", tok)
    if m:
        return [(m.group(1), int(m.group(2)))]
    for name in ("vcc", "exec", "scc", "m0"):
        if tok.startswith(name):
            out.append(name)
    return out


def parse_function(text, want):
    """[(block_label, in_loop_header_or_None, [instructions])] of the first function whose name contains `want`."""
    lines = text.splitlines()
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w.$]*):", ln)
        if m and not ln.startswith(".L") and want in m.group(1):
            start = i
            break
    if start is None:
        raise Violation(f"no function matching {want!r}")
    blocks = [["entry", None, []]]
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end"):
            break
        t = ln.strip()
        m = re.match(r"^(?:(\.LBB\w+):|; %bb\.(\d+):)\s*(;.*)?$", t)
        if m:
            comment = m.group(3) or ""
            hdr = None
            mh = re.search(r"in Loop: Header=(BB\w+)", comment)
            if mh:
                hdr = mh.group(1)
            elif "Loop Header" in comment and m.group(1):
                hdr = m.group(1)[2:]               # the header belongs to its own loop
            blocks.append([m.group(1) or f"bb.{m.group(2)}", hdr, []])
            continue
        if t.startswith(";") and blocks[-1][1] is None and not blocks[-1][2]:
            # the loop annotation of a block whose label line already carries a comment (an IR block name) sits on the NEXT line
            mh = re.search(r"in Loop: Header=(BB\w+)", t)
            if mh:
                blocks[-1][1] = mh.group(1)
            elif "Loop Header" in t and blocks[-1][0].startswith(".LBB"):
                blocks[-1][1] = blocks[-1][0][2:]
        if not t or t.startswith(";") or t.startswith("."):
            continue
        blocks[-1][2].append(t.split(";")[0].strip())
    return blocks


def split_ops(inst):
    parts = inst.split(None, 1)
    if len(parts) == 1:
        return parts[0], []
    return parts[0], [o.strip() for o in parts[1].split(",")]


def dest_count(op):
    """How many leading operands an instruction writes."""
    if op.startswith("v_cmp") and not op.startswith("v_cmpx"):
        return 1
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co", "v_div_scale")):
        return 2
    if op.startswith(("s_cmp", "s_cmpk", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_setprio", "s_sleep", "s_barrier", "s_endpgm")):
        return 0
    if op.startswith(("global_store", "scratch_store", "flat_store", "buffer_store", "ds_write")):
        return 0
    return 1


def check(text, want, allow_global_loads=1, allow_lds_reads=False):
    """allow_lds_reads (the constant-time fixed-base kernel): ds_read_* may appear in the loop when its address register is, like
    the scalar-word reload's, made of loop-invariant registers and clean scalars only -- every lane reads the SAME table entries."""
    blocks = parse_function(text, want)
    # the bit loop = the loop with the most instructions
    sizes = {}
    for _, hdr, insts in blocks:
        if hdr:
            sizes[hdr] = sizes.get(hdr, 0) + len(insts)
    if not sizes:
        raise Violation("no loop found")
    header = max(sizes, key=sizes.get)
    member = [n for n, (_, hdr, _) in enumerate(blocks) if hdr == header]
    # A ROTATED loop (round 4, k_scalar_mult<29>): the blocks that reload the scalar word sit in FRONT of the annotated header and are
    # entered by a branch from the loop's latch (`s_branch .LBB4_7` ... fall through into the header) -- the compiler's "in Loop" comments
    # leave them out, but they run once per 32 iterations.  Every block from the earliest such branch target up to the header belongs to
    # the cycle and is checked with it.
    pos = {lab: n for n, (lab, _, _) in enumerate(blocks)}
    first = member[0]
    for n in member:
        for inst in blocks[n][2]:
            op, ops = split_ops(inst)
            if (op == "s_branch" or op.startswith("s_cbranch")) and ops and ops[-1] in pos and pos[ops[-1]] < first:
                first = min(first, pos[ops[-1]])
    member = list(range(first, member[0])) + member
    loop = [(blocks[n][0], blocks[n][2]) for n in member]
    flat = [i for _, insts in loop for i in insts]
    report = {"kernel": want, "loop_header": header, "instructions": len(flat), "branches": [], "global_loads": [], "scratch": 0, "lds_reads": 0}

    # ---- clean scalars: reaching definitions over the loop's control-flow graph.  A scalar register read at some instruction is clean
    # if, walking back -- through the block, then through every predecessor block inside the loop -- every definition that can reach the
    # read is a scalar instruction (not a *_saveexec: that is a lane mask) whose own sources are clean where IT stands.  A path that leaves
    # the loop through the header's entry edge reaches a loop-invariant value: clean.  A (block, register) pair met again on its own path
    # counts as clean (the counter depends on itself).  VCC, EXEC and anything a vector instruction wrote are dirty.
    written_v = set()
    for _, insts in loop:
        for inst in insts:
            op, ops = split_ops(inst)
            for r in [r for o in ops[:dest_count(op)] for r in regs_of(o)]:
                if isinstance(r, tuple) and r[0] == "v":
                    written_v.add(r)
    labels = [lab for lab, _ in loop]
    index_of = {lab: n for n, lab in enumerate(labels)}
    block_of = {id(insts): n for n, (_, insts) in enumerate(loop)}
    succs = {n: set() for n in range(len(loop))}
    for n, (lab, insts) in enumerate(loop):
        falls = True
        for k, inst in enumerate(insts):
            op, ops = split_ops(inst)
            if op.startswith("s_cbranch") and ops and ops[-1] in index_of:
                succs[n].add(index_of[ops[-1]])
            if op == "s_branch":
                if ops and ops[0] in index_of:
                    succs[n].add(index_of[ops[0]])
                falls = False
            if op == "s_setpc_b64" and k >= 2:
                m = re.search(r"\((\.LBB\w+)-", insts[k - 2])
                if m and m.group(1) in index_of:
                    succs[n].add(index_of[m.group(1)])
                falls = False
            if op == "s_endpgm":
                falls = False
        if falls and n + 1 < len(loop):
            succs[n].add(n + 1)
    preds = {n: {m for m in succs if n in succs[m]} for n in range(len(loop))}

    # One query = one walk: any dirty definition on any path makes the whole query dirty (every level returns False at once), so a
    # (block, register) pair that the walk has already entered may be taken as clean wherever it is met again -- the greatest fixed point,
    # and each pair is expanded ONCE per query (a path-by-path walk is exponential in the diamonds of a loop with data-dependent branches).
    def def_is_clean(insts, k, seen):
        op, ops = split_ops(insts[k])
        if not op.startswith("s_") or "saveexec" in op:
            return False
        nd = dest_count(op)
        return all(clean_scalar_at(insts, k, s_, seen) for o in ops[nd:] for s_ in regs_of(o) if s_ != "scc")

    def clean_scalar_at(block_insts, idx, reg, seen=None):
        """Is scalar `reg`, read by instruction idx of this block, made of clean values on every path that reaches it?"""
        if seen is None:
            seen = set()
        if (isinstance(reg, tuple) and reg[0] == "v") or reg in ("vcc", "exec"):
            return False
        for k in range(idx - 1, -1, -1):
            op, ops = split_ops(block_insts[k])
            if reg in [r for o in ops[:dest_count(op)] for r in regs_of(o)]:
                dkey = (block_of[id(block_insts)], k, reg)
                if dkey in seen:
                    return True
                seen.add(dkey)
                return def_is_clean(block_insts, k, seen)
        n = block_of[id(block_insts)]
        key = (n, reg)
        if key in seen:
            return True
        seen.add(key)
        for m in preds[n]:                                    # (the header's entry edge from outside the loop: a loop-invariant value)
            pin = loop[m][1]
            if not clean_scalar_at(pin, len(pin), reg, seen):
                return False
        return True

    def scratch_slot(inst):
        """(first byte, last byte + 1) of a constant-addressed scratch access."""
        op, ops = split_ops(inst)
        m = re.search(r"offset:(\d+)", inst)
        first = int(m.group(1)) if m else 0
        mw = re.search(r"dwordx(\d)", op)
        return first, first + 4 * (int(mw.group(1)) if mw else 1)
    stored_in_loop = [scratch_slot(i) for i in flat if i.startswith("scratch_store")]

    def clean_vector_at(block_insts, idx, reg):
        for k in range(idx - 1, -1, -1):
            op, ops = split_ops(block_insts[k])
            nd = dest_count(op)
            if reg in [r for o in ops[:nd] for r in regs_of(o)]:
                if op.startswith("scratch_load"):
                    # a reload of a spill slot nothing in the loop stores to: a loop-invariant value (the spilled scalar pointer)
                    a, b = scratch_slot(block_insts[k])
                    return all(o.split()[0] == "off" for o in ops[1:]) and not any(a < d and c < b for c, d in stored_in_loop)
                if not op.startswith(("v_lshl_add_u64", "v_add_co_u32", "v_addc_co_u32", "v_add_u32", "v_mov_b32", "v_lshlrev_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_or_b32")):
                    return False
                for o in ops[nd:]:
                    for s in regs_of(o):
                        if isinstance(s, tuple) and s[0] == "v":
                            if not clean_vector_at(block_insts, k, s):
                                return False
                        elif s != "scc" and not clean_scalar_at(block_insts, k, s):
                            return False
                return True
        return reg not in written_v           # loop-invariant

    for lab, insts in loop:
        for idx, inst in enumerate(insts):
            op, ops = split_ops(inst)
            if allow_lds_reads and op.startswith("ds_read"):
                addr = ops[1].split()[0]
                for r in regs_of(addr):
                    if not clean_vector_at(insts, idx, r):
                        raise Violation(f"{lab}: the LDS address {addr} of `{inst}` is not made of loop-invariant registers and the window counter")
                report["lds_reads"] += 1
                continue
            if op == "s_setpc_b64" and idx >= 3 and insts[idx - 3].startswith("s_getpc_b64") and insts[idx - 2].startswith("s_add_u32") \
                    and insts[idx - 1].startswith("s_addc_u32") and ".LBB" in insts[idx - 2] and ".LBB" in insts[idx - 1]:
                report["long_jumps"] = report.get("long_jumps", 0) + 1       # branch relaxation: an unconditional jump to a label beyond s_branch's reach
                continue
            if re.match(r"^(v_readfirstlane|v_readlane|v_writelane|v_permlane|ds_|buffer_|flat_|global_store|global_atomic|s_setpc|s_swappc|s_call|s_cbranch_vcc|s_cbranch_exec|s_cbranch_cd|s_cbranch_g_fork|s_cbranch_i_fork|s_cbranch_join)", op):
                raise Violation(f"{lab}: `{inst}` is not allowed in the bit loop")
            if op.startswith("s_cbranch_scc"):
                k = idx - 1
                while k >= 0 and not SCC_WRITERS.match(split_ops(insts[k])[0]):
                    k -= 1
                if k < 0:
                    raise Violation(f"{lab}: `{inst}` has no SCC definition in its block")
                cop, cops = split_ops(insts[k])
                if not SCC_COMPARES.match(cop):
                    raise Violation(f"{lab}: `{inst}` takes SCC from `{insts[k]}`, not from a scalar compare")
                for o in cops:
                    for r in regs_of(o):
                        if not clean_scalar_at(insts, k, r):
                            raise Violation(f"{lab}: `{inst}` depends on `{insts[k]}` whose operand {o} is not a clean scalar")
                report["branches"].append(f"{insts[k]} ; {inst}")
            if op.startswith("global_load"):
                # global_load_dword vdst, v[addr], off|s[base] [offset:n]
                addr = ops[1]
                for r in regs_of(addr):
                    if not clean_vector_at(insts, idx, r):
                        raise Violation(f"{lab}: the address {addr} of `{inst}` is not made of loop-invariant registers and the iteration counter")
                base = ops[2].split()[0]
                if base != "off":
                    for r in regs_of(base):
                        if not clean_scalar_at(insts, idx, r):
                            raise Violation(f"{lab}: the scalar base of `{inst}` is not clean")
                report["global_loads"].append(inst)
            if op.startswith("scratch_"):
                addr_ops = ops[1:] if op.startswith("scratch_load") else [ops[0]] + ops[2:]
                if not all(o.split()[0] == "off" for o in addr_ops):
                    raise Violation(f"{lab}: `{inst}` does not use constant `off` addressing")
                report["scratch"] += 1
    if len(report["global_loads"]) > allow_global_loads:
        raise Violation(f"{len(report['global_loads'])} global loads in the bit loop, expected at most {allow_global_loads}: {report['global_loads']}")
    return report


def check_after_secret_load(text, want):
    """The parts of a kernel AROUND its loop (the first window, the exceptional scalar, k = 0): after the first global load that
    follows the workgroup barrier (or, in a kernel without one, the first global load) -- the scalar -- no instruction of the function may branch on a lane mask (s_cbranch_vcc* /
    s_cbranch_exec*), move a lane's value to the scalar unit (v_readfirstlane / v_readlane / v_permlane) or write LDS; every
    s_cbranch_scc* left is the loop's own (check() looks at those).  Returns the number of instructions looked at."""
    flat = [i for _, _, insts in parse_function(text, want) for i in insts]
    bar = next((k for k, i in enumerate(flat) if i.startswith("s_barrier")), 0)      # no barrier (no LDS table): the first global load is the scalar
    try:
        first = next(k for k in range(bar, len(flat)) if flat[k].startswith("global_load"))
    except StopIteration:
        raise Violation("no global load: not the kernel shape this check is for")
    for inst in flat[first:]:
        op = inst.split()[0]
        if re.match(r"^(s_cbranch_vcc|s_cbranch_exec|v_readfirstlane|v_readlane|v_writelane|v_permlane|ds_write|ds_bpermute|ds_permute|ds_swizzle|global_atomic|s_swappc|s_call)", op):
            raise Violation(f"`{inst}` after the scalar has been loaded")
    return len(flat) - first


if __name__ == "__main__":
    try:
        rep = check(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "k_scalar_mult")
    except Violation as exc:
        print("VIOLATION:", exc)
        sys.exit(1)
    for key, val in rep.items():
        print(f"{key}: {val}")
