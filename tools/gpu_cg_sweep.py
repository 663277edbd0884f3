"""BASELINE config 5: constant-geometry (stage-sweep) kernels at n=4096, 60-bit — lane grouping / LDS padding sweep.
Run under rocprofv3 (--kernel-trace --stats, then --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
plan = engine.Plan(n, q, psi)
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
ref = plan.poly_mult(a, b)
for v in ("cg", "cg8", "cg8_padded"):
    plan.time_poly_mult(a, b, c, 2, v)
    ms = min(plan.time_poly_mult(a, b, c, 5, v) for _ in range(2))
    ok = torch.equal(c, ref)
    print(f"{v:11s} {ms:8.3f} ms  {B/ms*1e3/1e6:7.3f} M polymul/s  {B*3*n*8/ms/1e6:8.1f} GB/s algorithmic  bit-exact vs fused: {ok}", flush=True)
