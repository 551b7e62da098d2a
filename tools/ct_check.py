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


def parse_function(text, want, keep_implicit_defs=False):
    """[(block_label, in_loop_header_or_None, [instructions])] of the first function whose name contains `want`.  keep_implicit_defs: the compiler's
    `; implicit-def: $sgpr2_sgpr3` notes (the register's value is UNDEFINED from here on as far as the program is concerned: whatever it still
    holds cannot influence the result) are kept as pseudo-instructions `.implicit_def s2, s3`."""
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
        if keep_implicit_defs and t.startswith("; implicit-def: $"):
            regs = []
            for m in re.finditer(r"([sv])gpr(\d+)", t):
                regs.append(f"{m.group(1)}{m.group(2)}")
            mr = re.search(r"\$([sv])gpr(\d+)_\1gpr(\d+)(?:_\1gpr(\d+))*", t)
            if mr:                                            # a tuple names its first and last (or every) register: fill the range
                nums = [int(x) for x in re.findall(r"gpr(\d+)", t)]
                regs = [f"{mr.group(1)}{i}" for i in range(min(nums), max(nums) + 1)]
            if "$vcc" in t:
                regs.append("vcc")
            if regs:
                blocks[-1][2].append(".implicit_def " + ", ".join(regs))
            continue
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


# ---------------------------------------------------------------------------------------------------------------- secret-flow (taint) analysis
# check() and check_after_secret_load() are SHAPE checks: they know which loop is the bit loop and refuse every lane-mask branch.  A kernel like
# k_ecdsa_sign_scalars has public lane-dependent control flow (`if (e >= n) break` on the element index) beside its secrets, so the question
# there is not "is there an exec branch" but "can a SECRET reach one".  check_secret_flow answers that on the ISA: a forward data-flow analysis
# over the function's control-flow graph to a fixed point.
#   sources    every global load whose address derives from a kernel argument named secret (by position; pointer provenance is tracked from the
#              s_load of the kernarg segment through the address arithmetic), scratch reloads once a secret was spilled, LDS reads once a secret
#              was written to LDS;
#   propagates through every instruction: the destinations take the union of the sources' tags (SCC, VCC and EXEC are registers like any other;
#              s_cselect / s_cmov / s_addc / s_subb read SCC);
#   sinks      the condition of a branch (SCC, VCC, EXEC), the address of a global / scratch / LDS access, and -- while the lane mask EXEC itself is
#              made of a secret (field.cuh's conditional corrections are moves under EXEC: same issue slots whatever the mask) -- every memory
#              access, lane read and EXEC branch, whose cost or address set WOULD depend on the mask.  A secret there is a Violation.
# What it does not model: memory (a secret stored through a pointer NOT named secret and loaded back is lost -- name scratch buffers secret), and
# instruction timing itself (gfx9 has no data-dependent-latency integer instruction; divides and transcendental ops are refused outright).
def kernel_args(text, want):
    """[(offset, size, value_kind)] of the first kernel whose name contains `want`, from the .amdgpu_metadata YAML of the unit."""
    meta = text[text.index(".amdgpu_metadata"):]
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        m = re.search(r"\.name:\s+(\S+)", blk)
        if m and want in m.group(1):
            return [(int(o), int(z), k) for o, z, k in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+\.value_kind:\s+(\w+)", blk)]
    raise Violation(f"no kernel metadata for {want!r}")


def kernarg_pointer(text, want):
    """The SGPR pair that holds the kernarg segment pointer on entry, from the kernel descriptor's user-SGPR directives."""
    m = re.search(r"\.amdhsa_kernel\s+(\S*" + re.escape(want) + r"\S*)(.*?)\.end_amdhsa_kernel", text, re.S)
    if not m:
        raise Violation(f"no kernel descriptor for {want!r}")
    d = dict(re.findall(r"\.amdhsa_user_sgpr_(\w+)\s+(\d+)", m.group(2)))
    first = 0
    for name, width in (("private_segment_buffer", 4), ("dispatch_ptr", 2), ("queue_ptr", 2)):
        if int(d.get(name, "0")):
            first += width
    if not int(d.get("kernarg_segment_ptr", "0")):
        raise Violation("the kernel takes no kernarg segment pointer")
    return first


_IMPLICIT_SCC_READ = re.compile(r"^s_(cselect|cmov|addc|subb|cbranch_scc)")
_NO_DATA = re.compile(r"^(s_waitcnt|s_nop|s_endpgm|s_barrier|s_setprio|s_sethalt|s_dcache|s_icache|s_code_end|s_ttracedata|s_inst_prefetch|s_clause|s_setreg|s_getreg|s_memtime|s_memrealtime|s_trap|s_branch)")
_REFUSED = re.compile(r"^(v_rcp|v_rsq|v_sqrt|v_div|v_exp|v_log|v_sin|v_cos|s_sleep|s_swappc|s_call|s_cbranch_cd|s_cbranch_g_fork|s_cbranch_i_fork|s_cbranch_join|global_atomic|flat_|buffer_|ds_bpermute|ds_permute|ds_swizzle|image_)")


def _operand_regs(tok):
    """Registers an operand token names, modifiers (neg(), |x|, sext(), op_sel, offsets) stripped; vcc_lo / vcc_hi / exec_lo / exec_hi fold into vcc / exec."""
    tok = re.sub(r"\b(offset|offset0|offset1|op_sel|op_sel_hi|neg_lo|neg_hi|clamp|omod|dst_sel|src0_sel|src1_sel|row_\w+|quad_perm|bank_mask|row_mask|bound_ctrl|sc\d|nt|glc|slc)\b[:\[\]\w,]*", " ", tok)
    out = []
    for m in re.finditer(r"\b([sv])\[(\d+):(\d+)\]|\b([sv])(\d+)\b|\b(vcc|exec|scc|m0)(?:_lo|_hi)?\b", tok):
        if m.group(1):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        elif m.group(4):
            out.append((m.group(4), int(m.group(5))))
        else:
            out.append(m.group(6))
    return out


def check_secret_flow(text, want, secret_args, public_loads_from=()):
    """Taint analysis of the first function whose name contains `want`; `secret_args` = positions (0-based) of the pointer arguments whose contents
    are secret.  Returns a report dict; raises Violation where a secret reaches a sink."""
    blocks = parse_function(text, want, keep_implicit_defs=True)
    args = kernel_args(text, want)
    kptr = kernarg_pointer(text, want)
    secret_ranges = []
    for a in secret_args:
        off, size, kind = args[a]
        if kind != "global_buffer":
            raise Violation(f"argument {a} is not a pointer ({kind})")
        secret_ranges.append((off, off + size, a))
    labels = {lab: n for n, (lab, _, _) in enumerate(blocks)}
    succs = {n: set() for n in range(len(blocks))}
    for n, (lab, _, insts) in enumerate(blocks):
        falls = True
        for k, inst in enumerate(insts):
            op, ops = split_ops(inst)
            if op.startswith("s_cbranch") and ops and ops[-1] in labels:
                succs[n].add(labels[ops[-1]])
            if op == "s_branch":
                if ops and ops[0] in labels:
                    succs[n].add(labels[ops[0]])
                falls = False
            if op == "s_setpc_b64":
                tgt = [m.group(1) for j in range(max(0, k - 3), k) for m in [re.search(r"\((\.LBB\w+)-", insts[j])] if m]
                if not tgt or tgt[0] not in labels:
                    raise Violation(f"{lab}: `{inst}` is not a relaxed branch to a label of this function")
                succs[n].add(labels[tgt[0]])
                falls = False
            if op == "s_endpgm":
                falls = False
        if falls and n + 1 < len(blocks):
            succs[n].add(n + 1)
    preds = {n: [m for m in succs if n in succs[m]] for n in range(len(blocks))}
    S = "S"
    report = {"kernel": want, "instructions": sum(1 for b in blocks for i in b[2] if not i.startswith(".implicit_def")), "secret_loads": 0, "public_branches": 0, "lane_mask_updates": 0,
              "secret_scratch": False, "secret_lds": False, "kernarg_sgpr": kptr}
    kp = {("s", kptr), ("s", kptr + 1)}

    def transfer(n, state, final):
        """state: {reg: frozenset(tags)} plus the pseudo-registers 'scratch' / 'lds'.  Runs block n; with `final` set it raises on a violation."""
        st = dict(state)
        tags = lambda r: st.get(r, frozenset())
        lab = blocks[n][0]
        for inst in blocks[n][2]:
            op, ops = split_ops(inst)
            if _NO_DATA.match(op):
                continue
            if op == ".implicit_def":                            # undefined from here on: stale contents cannot matter
                for r in [r for o in ops for r in _operand_regs(o)]:
                    st[r] = frozenset()
                continue
            if _REFUSED.match(op):
                raise Violation(f"{lab}: `{inst}` is not allowed in a kernel that handles secrets")
            nd = dest_count(op)
            dst = [r for o in ops[:nd] for r in _operand_regs(o)]
            src = [r for o in ops[nd:] for r in _operand_regs(o)]
            if _IMPLICIT_SCC_READ.match(op):
                src.append("scc")
            t = frozenset().union(*[tags(r) for r in src]) if src else frozenset()
            bad = lambda what: (_ for _ in ()).throw(Violation(f"{lab}: a secret reaches {what} of `{inst}`")) if final else None
            masked = S in tags("exec")                           # the set of active lanes depends on a secret: moves under EXEC are fine (same issue
            if masked and op.startswith(("global_", "scratch_", "ds_", "s_load", "v_readfirstlane", "v_readlane", "v_writelane", "s_cbranch_exec")):
                bad("the lane mask (EXEC) in force at")          # slot whatever the mask), anything whose cost or address set depends on it is not
            if masked and op.startswith("v_"):
                t = t | tags("exec") | frozenset().union(*[tags(r) for r in dst if isinstance(r, tuple)])    # inactive lanes keep the old value
            if op.startswith("s_load_dword"):
                base = set(_operand_regs(ops[1]))
                if base == kp and re.fullmatch(r"(0x[0-9a-f]+|\d+)", ops[2].split()[0]):
                    first = int(ops[2].split()[0], 0)
                    for j, r in enumerate(dst):
                        b = first + 4 * j
                        hit = [a for lo, hi, a in secret_ranges if lo <= b < hi]
                        st[r] = frozenset({f"P{hit[0]}"}) if hit else frozenset()
                else:                                            # a scalar load through some other pointer: secret if the pointer is
                    if S in t:
                        bad("the address")
                    sec = any(x.startswith("P") for x in t)
                    for r in dst:
                        st[r] = frozenset({S}) if sec else frozenset()
                    report["secret_loads"] += int(sec and final)
                continue
            if op.startswith("global_load"):
                if S in t:
                    bad("the address")
                sec = any(x.startswith("P") for x in t)
                for r in dst:
                    st[r] = frozenset({S}) if sec else frozenset()
                report["secret_loads"] += int(sec and final)
                continue
            if op.startswith("global_store"):
                addr = frozenset().union(*[tags(r) for o in (ops[0], ops[2]) for r in _operand_regs(o)])
                if S in addr:
                    bad("the address")
                continue
            if op.startswith("scratch_load") or op.startswith("scratch_store"):
                # spill slots are constant-addressed (`off ... offset:N`): tracked per dword, so that a spilled POINTER does not inherit the taint of a
                # spilled field word (k_gc_scalar_mult<32, *> spills both); any other addressing form falls back to one location for all of scratch
                addr_ops = ops[1:] if op.startswith("scratch_load") else [ops[0]] + ops[2:]
                if S in frozenset().union(*[tags(r) for o in addr_ops for r in _operand_regs(o)]):
                    bad("the address")
                mo = re.search(r"offset:(\d+)", inst)
                mw = re.search(r"dwordx(\d)", op)
                width = int(mw.group(1)) if mw else 1
                constant = all(o.split()[0] == "off" or re.fullmatch(r"s\d+", o.split()[0]) for o in addr_ops)
                slots = [f"scratch@{(int(mo.group(1)) if mo else 0) + 4 * j}" for j in range(width)] if constant else None
                if op.startswith("scratch_load"):
                    for j, r in enumerate(dst):
                        st[r] = tags("scratch") | (tags(slots[j]) if slots and j < len(slots) else frozenset())
                else:
                    data = _operand_regs(ops[1])
                    if slots:
                        for j, r in enumerate(data):
                            if j < len(slots):
                                st[slots[j]] = tags(r)
                    else:
                        st["scratch"] = tags("scratch") | frozenset().union(*[tags(r) for r in data])
                continue
            if op.startswith("ds_read") or op.startswith("ds_load"):
                if S in frozenset().union(*[tags(r) for r in _operand_regs(ops[1])]):
                    bad("the LDS address")
                for r in dst:
                    st[r] = tags("lds")
                continue
            if op.startswith("ds_write") or op.startswith("ds_store"):
                if S in frozenset().union(*[tags(r) for r in _operand_regs(ops[0])]):
                    bad("the LDS address")
                st["lds"] = tags("lds") | frozenset().union(*[tags(r) for o in ops[1:] for r in _operand_regs(o)])
                continue
            if op.startswith("s_cbranch_scc"):
                if S in tags("scc"):
                    bad("the condition (SCC)")
                report["public_branches"] += int(final)
                continue
            if op.startswith("s_cbranch_vcc"):
                if S in tags("vcc"):
                    bad("the condition (VCC)")
                report["public_branches"] += int(final)
                continue
            if op.startswith("s_cbranch_exec"):
                if S in tags("exec"):
                    bad("the condition (EXEC)")
                report["public_branches"] += int(final)
                continue
            if op == "s_setpc_b64":
                continue                                         # (checked above: a relaxed branch to a label)
            if op.startswith("s_getpc"):
                for r in dst:
                    st[r] = frozenset()
                continue
            if "saveexec" in op:                                 # sdst = EXEC; EXEC = EXEC op src: a lane mask made of a secret is allowed (see `masked`)
                for r in dst:
                    st[r] = tags("exec")
                st["exec"] = tags("exec") | t
                st["scc"] = st["exec"]
                report["lane_mask_updates"] += int(final)
                continue
            if op.startswith("v_cmpx"):
                st["exec"] = tags("exec") | t
                for r in dst:
                    st[r] = t
                report["lane_mask_updates"] += int(final)
                continue
            if "exec" in dst:
                # `s_or_b64 exec, exec, saved` closes a structured region: EXEC is a subset of the saved mask, the result IS the saved mask
                if op == "s_or_b64" and _operand_regs(ops[1]) == ["exec"]:
                    t = frozenset().union(*[tags(r) for r in _operand_regs(ops[2])])
                st["exec"] = t
                if op.startswith("s_") and SCC_WRITERS.match(op):
                    st["scc"] = t
                report["lane_mask_updates"] += int(final)
                continue
            if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64", "v_lshlrev_b64", "s_lshl_b64")) and len(dst) >= 2:
                # dword-precise: the LOW half of a 64-bit multiply-add / shift-left / add does not depend on the high half of its 64-bit operand (the
                # compiler forms 64-bit element offsets this way and leaves a stale -- possibly secret -- register in the unused high half)
                pair = lambda o: (lambda rs: (rs[:1], rs[1:]) if len(rs) == 2 else (rs, rs))(_operand_regs(o))
                un = lambda rs: frozenset().union(*[tags(r) for r in rs]) if rs else frozenset()
                if op.startswith("v_mad"):
                    lo = un(_operand_regs(ops[2])) | un(_operand_regs(ops[3])) | un(pair(ops[4])[0])
                elif op.startswith("v_lshl_add_u64"):
                    lo = un(pair(ops[1])[0]) | un(_operand_regs(ops[2])) | un(pair(ops[3])[0])
                else:
                    sh, val = (ops[1], ops[2]) if op.startswith("v_lshlrev") else (ops[2], ops[1])
                    lo = un(_operand_regs(sh)) | un(pair(val)[0])
                st[dst[0]] = lo
                for r in dst[1:]:
                    st[r] = t
                if op.startswith("s_"):
                    st["scc"] = t
                continue
            for r in dst:
                st[r] = t
            if op.startswith("s_") and SCC_WRITERS.match(op):
                st["scc"] = t
        return st

    def join(states):
        out = {}
        for s_ in states:
            for r, t in s_.items():
                out[r] = out.get(r, frozenset()) | t
        return out

    outs = [None] * len(blocks)
    changed = True
    rounds = 0
    while changed:
        changed = False
        rounds += 1
        for n in range(len(blocks)):
            ins = join([outs[m] for m in preds[n] if outs[m] is not None])
            o = transfer(n, ins, False)
            if o != outs[n]:
                outs[n] = o
                changed = True
        if rounds > 200:
            raise Violation("the analysis did not converge")
    for n in range(len(blocks)):
        o = transfer(n, join([outs[m] for m in preds[n] if outs[m] is not None]), True)
        report["secret_scratch"] |= any(S in t_ for k_, t_ in o.items() if isinstance(k_, str) and k_.startswith("scratch"))
        report["secret_lds"] |= S in o.get("lds", frozenset())
    if report["secret_loads"] == 0:
        raise Violation("no load through a secret pointer was found: the analysis did not see the secrets")
    return report


if __name__ == "__main__":
    try:
        rep = check(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else "k_scalar_mult")
    except Violation as exc:
        print("VIOLATION:", exc)
        sys.exit(1)
    for key, val in rep.items():
        print(f"{key}: {val}")
