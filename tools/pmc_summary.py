#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one dir per pass) for the fused kernel: mean per dispatch."""
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "polymul_fused_kernel" not in name and "cg_kernel" not in name:
                continue
            agg[(name.split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:62s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
