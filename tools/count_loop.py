#!/usr/bin/env python3
"""Count the instructions of the biggest loop of a gfx950 assembly file (hipcc -S --cuda-device-only).
Usage: count_loop.py file.s [kernel-name-substring]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
# split into functions
funcs, cur, name = {}, None, None
for ln in lines:
    m = re.match(r"^([A-Za-z_][\w.$]*):", ln)
    if m and not ln.startswith(".L"):
        name = m.group(1); cur = funcs.setdefault(name, [])
    elif cur is not None:
        cur.append(ln)
for name, body in funcs.items():
    if want not in name or len(body) < 200:
        continue
    labels = {}
    insts = []
    for ln in body:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        insts.append(t.split(";")[0].strip())
    best = None
    for i, t in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            span = (labels[m.group(1)], i + 1)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if best is None:
        print(name, "no loop"); continue
    loop = insts[best[0]:best[1]]
    c = collections.Counter(t.split()[0] for t in loop)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"{name}: loop of {len(loop)} instructions, {valu} VALU, {sum(v for k, v in c.items() if k.startswith('s_'))} scalar")
    for k, v in c.most_common(24):
        print(f"   {k:28s} {v}")
