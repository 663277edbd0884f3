"""Developer tool: fused kernel time for the 24-bit shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
for name, n, psi, B in (("n=1024", 1024, 5548360, 262144), ("n=4096", 4096, 283817, 65536), ("n=256", 256, 1239911, 1048576)):
    plan = engine.Plan(n, 8380417, psi)
    a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
    plan.time_poly_mult(a, b, c, 3)
    ms = min(plan.time_poly_mult(a, b, c, 10) for _ in range(3))
    ok = torch.equal(plan.poly_mult(a[:256], b[:256], variant="cg"), c[:256])
    print(f"{name} 24-bit B={B}: {ms:.3f} ms  {B/ms*1e3/1e6:.1f} M polymul/s  {B*3*n*4/ms/1e6:.0f} GB/s frac {B*3*n*4/ms/1e6/8000:.3f} parity_vs_cg={ok}", flush=True)
    plan.close()
