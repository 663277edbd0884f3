#!/usr/bin/env python3
"""Developer tool: list the scratch spills / reloads and SGPR-to-VGPR-lane spills of one kernel with the
instruction that defined each spilled value (prologue vs loop).  usage: asm_spills.py file.s kernel-substring"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
hi = next((i for i, l in enumerate(body) if "Loop Header" in l), len(body))
lastdef = {}
nrl = 0
for k, l in enumerate(body):
    t = l.split(";")[0].strip()
    m = re.match(r"(\S+)\s+(v\d+|v\[\d+:\d+\])\s*,(.*)", t)
    if t.startswith("scratch_"):
        r = re.search(r"(v\[\d+:\d+\]|v\d+)", t).group(1)
        print(k, "LOOP" if k > hi else "PRO ", t[:58], ("<= " + lastdef.get(r, "?")[:70]) if t.startswith("scratch_store") else "")
    elif m and not t.startswith(("global_store", "ds_write")):
        lastdef[m.group(2)] = t
    if t.startswith("v_readlane") and k > hi: nrl += 1
print("v_readlane in loop:", nrl)
