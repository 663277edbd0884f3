#!/usr/bin/env python3
"""Developer tool: VALU issue-cycle estimate per marked region of one kernel.
Build the .s with -DTN_MARKS (kernels.hip emits '; TNMARK name' comments), then:
    asm_regions.py file.s substring-of-mangled-name"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
FAST = ("v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32",
        "v_lshlrev_b32", "v_cndmask_b32", "v_accvgpr", "v_not_b32")
reg = "prologue"
cyc, cnt, mul, order = collections.Counter(), collections.Counter(), collections.Counter(), []
for l in lines[start:end + 1]:
    m = re.search(r"; TNMARK (\w+)", l)
    if m:
        reg = m.group(1)
        continue
    m = re.match(r"^\s+(v_[a-z0-9_]+)", l)
    if m:
        op = m.group(1)
        if reg not in cnt: order.append(reg)
        cnt[reg] += 1
        cyc[reg] += 2 if op.startswith(FAST) else 4
        if op.startswith(("v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32")): mul[reg] += 1
tot = sum(cyc.values())
print(f"{key}: {sum(cnt.values())} VALU, est {tot} issue cycles/wave")
for r in order:
    print(f"  {r:18s} {cnt[r]:5d} VALU  {mul[r]:5d} mul  {cyc[r]:6d} cycles  {100.0 * cyc[r] / tot:5.1f}%")
