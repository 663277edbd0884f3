"""Headline ledger (round-2 VERDICT item 2): same-box A/B of the fused product kernel with and without the
TN_PLAN_CANONICAL_INPUTS promise (the two instantiations live in ONE library: two plans), alternated.
usage: gpu_ledger.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p0 = engine.Plan(n, q, psi)
p1 = engine.Plan(n, q, psi, flags=engine.PLAN_CANONICAL_INPUTS)
a = p0.fill_lcg(B, 1, 2); b = p0.fill_lcg(B, 2, 2); c0 = torch.empty_like(a); c1 = torch.empty_like(a)
for p, c in ((p0, c0), (p1, c1)):
    p.time_poly_mult(a, b, c, 60, "fused")
print("results equal:", bool(torch.equal(c0, c1)), " row-0 checksum:", int(p1.checksum_rows(c1[:1])[0]))
for rep in range(4):
    m0 = p0.time_poly_mult(a, b, c0, 200, "fused")
    m1 = p1.time_poly_mult(a, b, c1, 200, "fused")
    print(f"rep {rep}: any-word inputs {m0:.4f} ms (frac {B*3*n*8/m0/1e6/8000:.4f})   promised-canonical inputs {m1:.4f} ms (frac {B*3*n*8/m1/1e6/8000:.4f})   ratio {m1/m0:.4f}", flush=True)
