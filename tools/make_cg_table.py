"""Table of the constant-geometry sweep's PMC passes: tools/make_cg_table.py gpurun_out/<tag>  (reads <tag>/summary.txt written by
gpu_cg_profile.sh; one row per cg_kernel<unsigned long, GROUP, LAYOUT, arithmetic, big, 12> instantiation = one point of the sweep)."""
import re, sys
d = sys.argv[1]
rows = {}
for ln in open(f"{d}/summary.txt"):
    m = re.search(r"cg_kernel<unsigned long, (\d+), (\d), \d, (?:false|true), 12>\S*\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", ln)
    if m:
        rows.setdefault((int(m.group(1)), int(m.group(2))), {})[m.group(3)] = float(m.group(4))
lay = {0: "linear", 1: "padded", 2: "swizzled"}
B = 65536
print("# BASELINE config 5: constant-geometry kernels (multi-stage trips, cg_kernel_impl.h), n=4096, q=2^60-2^14+1, 65,536 products per launch (tools/gpu_cg_profile.sh, rocprofv3 --pmc, separate passes)")
print("# conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; FETCH_SIZE x2 on gfx950 (KiB); algorithmic traffic: 4,194,304 KiB read, 2,097,152 KiB written")
print("# stages/trip = log2(2 GROUP); LDS transposes per product = 3 (trips - 1); workgroup barriers per product = 2 per transpose (round 2: 36 + 3 at every GROUP)")
print("GROUP layout    conflict  LDS_IDX_ACTIVE  SQ_INSTS_VALU  VALU/product  SQ_INSTS_LDS  barriers/product  FETCH_SIZE x2 (KiB)  WRITE_SIZE(KiB)")
for (g, l), c in sorted(rows.items()):
    idx = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    L = (2 * g).bit_length() - 1
    trips = (12 + L - 1) // L
    print(f"{g:5d} {lay[l]:10s}  {c.get('SQ_LDS_BANK_CONFLICT', 0.0) / idx if idx else 0:.4f}  {idx:.3e}       {c.get('SQ_INSTS_VALU', 0):.3e}     "
          f"{c.get('SQ_INSTS_VALU', 0) / B:9.0f}   {c.get('SQ_INSTS_LDS', 0):.3e}    {6 * (trips - 1):3d}               {2 * c.get('FETCH_SIZE', 0):.0f}              {c.get('WRITE_SIZE', 0):.0f}")
