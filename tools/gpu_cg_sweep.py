"""BASELINE config 5: constant-geometry (stage-sweep) kernels at n=4096, 60-bit — lane grouping x LDS layout sweep.
Run under rocprofv3 (--kernel-trace --stats, then --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ...): each variant is its
own kernel instantiation (cg_kernel<unsigned long, GROUP, LAYOUT>), so the per-kernel rows of the profile are the sweep.
usage: gpu_cg_sweep.py [rows] [variant ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variants = sys.argv[2:] or ["cg", "cg_swizzled", "cg2", "cg2_padded", "cg2_swizzled", "cg4", "cg4_padded", "cg4_swizzled", "cg8", "cg8_padded", "cg8_swizzled"]
SPIN_S = float(os.environ.get("SPIN_S", "0.15"))
plan = engine.Plan(n, q, psi)
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
ref = plan.poly_mult(a, b)
for v in variants:
    plan.time_poly_mult(a, b, c, 1, v)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < SPIN_S:                       # the shader clock settles ~0.1 s after idle / after a lighter kernel
        plan.time_poly_mult(a, b, c, 8, v)
    ms = min(plan.time_poly_mult(a, b, c, 10, v) for _ in range(2))
    ok = torch.equal(c, ref)
    print(f"{v:13s} {ms:8.3f} ms  {B/ms*1e3/1e6:7.3f} M polymul/s  {B*3*n*8/ms/1e6:8.1f} GB/s algorithmic  frac {B*3*n*8/ms/1e6/8000:.4f}  bit-exact vs fused: {ok}", flush=True)
