#!/usr/bin/env python3
"""Approximate VGPR liveness over one kernel's (mostly straight-line) asm: prints live-count profile and the peak region.
usage: asm_liveness.py file.s kernel-substring"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l for l in lines[start + 1:end] if re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None: out.append(int(m.group(3)))
        else: out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
ins = []
for l in body:
    t = l.split(";")[0].strip()
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    if op.startswith(("s_", "buffer_wb", "buffer_inv")) and not op.startswith("s_load"):
        d, u = [], sum((regs(o) for o in ops), [])
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch")): d, u = [], []
        ins.append((op, d, u, t)); continue
    if op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store")):
        d, u = [], sum((regs(o) for o in ops), [])
    elif op.startswith(("v_cmp", "v_cmpx")):
        d, u = [], sum((regs(o) for o in ops), [])
    else:
        d = regs(ops[0]) if ops else []
        u = sum((regs(o) for o in ops[1:]), [])
        if op in ("v_mac_f32", "v_fmac_f32"): u += d
    ins.append((op, d, u, t))
live = set(); prof = [0] * len(ins)
for i in range(len(ins) - 1, -1, -1):
    op, d, u, t = ins[i]
    for r in d: live.discard(r)
    for r in u: live.add(r)
    prof[i] = len(live)
peak = max(prof); pi = prof.index(peak)
print(f"{len(ins)} instrs, peak live VGPRs {peak} at instr {pi}")
step = max(1, len(ins) // 60)
marks = {i: t for i, (op, d, u, t) in enumerate(ins) if op in ("s_barrier",) or op.startswith(("global_load", "global_store", "scratch_"))}
for i in range(0, len(ins), step):
    seg = ins[i:i + step]
    tags = set()
    for op, d, u, t in seg:
        if op == "s_barrier": tags.add("BAR")
        if op.startswith("global_load"): tags.add("gld")
        if op.startswith("global_store"): tags.add("gst")
        if op.startswith("scratch_"): tags.add("SCR")
        if op.startswith("ds_"): tags.add("lds")
        if op.startswith("s_load"): tags.add("sld")
    print(f"{i:6d} live={max(prof[i:i+step]):4d} {' '.join(sorted(tags))}")
