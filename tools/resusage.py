#!/usr/bin/env python3
"""Condense `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin) into one line per kernel."""
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        try: name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip(); m2 = re.search(r"(k_\w+(<[^>]*>)?)", name); name = m2.group(1) if m2 else name
        except Exception: pass
        cur = {"name": name}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
    if "error" in line: print(line.rstrip())
for r in rows:
    print(f"{r['name'][:70]:70s} vgpr={r.get('VGPRs',-1):3d} agpr={r.get('AGPRs',0):3d} sgpr={r.get('TotalSGPRs',-1):3d} scratch={r.get('ScratchSize',0):4d} occ={r.get('Occupancy',-1)} spill={r.get('VGPRs Spill',0)}")
