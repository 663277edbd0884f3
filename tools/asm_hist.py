#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -save-temps .s file (developer tool).
usage: asm_hist.py file.s substring-of-mangled-name"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(("function", ":")) or (l.startswith("_ZN") and key in l and ":" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = collections.Counter()
for l in lines[start:end + 1]:
    m = re.match(r"^\s+((?:v|s|ds|global|buffer|flat|scratch)_[a-z0-9_]+)", l)
    if m:
        ins[m.group(1)] += 1
tot = sum(ins.values())
valu = sum(v for k, v in ins.items() if k.startswith("v_"))
print(f"{key}: {tot} instructions, {valu} VALU, {end - start} lines")
FAST = ("v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_cndmask_b32", "v_accvgpr")
fast = sum(v for k, v in ins.items() if k.startswith(FAST))
print(f"  2-cycle-class VALU: {fast}, 4-cycle-class VALU: {valu - fast}, est. issue cycles/wave: {fast * 2 + (valu - fast) * 4}")
for k, v in ins.most_common(45):
    print(f"  {k:30s} {v}")
