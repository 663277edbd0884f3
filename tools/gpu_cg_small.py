"""Developer tool: constant-geometry product kernels at short rows / other shapes (row hand-out A/B: chunks vs one atomic per row vs fixed stride).
usage: gpu_cg_small.py [variant ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
variants = sys.argv[1:] or ["cg8", "cg4", "cg"]
for name, n, q, psi, B in (("n=256 24-bit", 256, 8380417, 1239911, 1048576), ("n=1024 24-bit", 1024, 8380417, 5548360, 262144),
                           ("n=4096 24-bit", 4096, 8380417, 283817, 65536), ("n=256 60-bit", 256, 1152921504606830593, pow(431606828070683274, 16, 1152921504606830593), 524288),
                           ("n=1024 60-bit", 1024, 1152921504606830593, pow(431606828070683274, 4, 1152921504606830593), 131072)):
    plan = engine.Plan(n, q, psi)
    a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
    ref = plan.poly_mult(a, b)
    for v in variants:
        plan.time_poly_mult(a, b, c, 1, v)
        ms = min(plan.time_poly_mult(a, b, c, 3, v) for _ in range(2))
        ok = torch.equal(c, ref)
        print(f"{name:14s} B={B:8d} {v:13s} {ms:8.3f} ms  {B/ms*1e3/1e6:8.2f} M polymul/s  frac {B*3*n*plan.elem_bytes/ms/1e6/8000:.4f}  equal to fused: {ok}", flush=True)
    plan.close()
