"""Kernel time at n = 8192, q = 2^60 - 2^14 + 1 (fused product, standalone transforms, stage-sweep kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 8192, 1152921504606830593, 458558429756866
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
plan = engine.Plan(n, q, psi)
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
for v in ("fused", "cg", "cg2_padded", "cg4_padded", "cg8_padded", "cg8_swizzled"):
    plan.time_poly_mult(a, b, c, 3, v)
    ms = min(plan.time_poly_mult(a, b, c, 10, v) for _ in range(2))
    print(f"n=8192 {v:12s} poly_mult: {ms:8.3f} ms  {B/ms*1e3/1e6:7.3f} M/s  {B*3*n*8/ms/1e6:7.0f} GB/s  frac {B*3*n*8/ms/1e6/8000:.3f}", flush=True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("cg_ntt", plan.ntt_forward), ("cg_intt", plan.ntt_inverse), ("twist+ntt", plan.twisted_ntt_forward)):
    fn(a, out=c); torch.cuda.synchronize()
    ev0.record()
    for _ in range(10): fn(a, out=c)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    print(f"n=8192 fused {name:10s}: {ms:8.3f} ms  {B/ms*1e3/1e6:7.3f} M/s  {B*2*n*8/ms/1e6:7.0f} GB/s  frac {B*2*n*8/ms/1e6/8000:.3f}", flush=True)
