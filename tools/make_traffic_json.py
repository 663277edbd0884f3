#!/usr/bin/env python3
"""Write profiles/traffic_latest.json from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc.sh (run on the GPU box, so
that the library build id recorded is the one the counters were taken on).  usage: make_traffic_json.py <pmc dir> [rows] [config]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tiny_ntt_amd import engine
out = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
config = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "polymul_fused_kernel" in row.get("Kernel_Name", "") and row["Counter_Name"] in vals:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
n, w = (4096, 8) if config == "cfg3" else (1024, 4)
j = {"rows": rows, "config": config, "kernel": "polymul_fused_kernel", "lib_build_id": engine.build_id(),
     "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
     "fetch_size_kib_raw": fetch, "write_size_kib_raw": write, "dispatches": len(vals["FETCH_SIZE"]),
     "correction": "gfx950: FETCH_SIZE counts 64 B per 128 B streaming request -> doubled (MI355X_MICROARCH.md HBM section; calibrated on this "
                   "kernel's own access pattern: 8-byte-per-lane non-temporal loads of a known 2 x rows x n x w bytes read exactly half); WRITE_SIZE exact "
                   "(includes the first-iteration placeholder stores of the software-pipelined store, one row per resident workgroup). "
                   "Separate --pmc passes (tools/gpu_pmc.sh).",
     "algorithmic_bytes_per_launch": rows * 3 * n * w}
with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as f:
    json.dump(j, f, indent=1)
print(json.dumps(j))
