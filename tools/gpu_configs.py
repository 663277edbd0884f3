"""Developer tool: kernel time of every BASELINE configuration that runs on one GPU (parity is covered by tests)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
CFG = [("cfg2 n=1024 24-bit batch=4096", 1024, 8380417, 5548360, 4096, 4),
       ("n=1024 24-bit batch=262144", 1024, 8380417, 5548360, 262144, 4),
       ("n=4096 24-bit batch=65536", 4096, 8380417, 283817, 65536, 4),
       ("n=256 24-bit batch=1048576", 256, 8380417, 1239911, 1048576, 4),
       ("cfg3 n=4096 60-bit batch=65536", 4096, 1152921504606830593, 431606828070683274, 65536, 8),
       ("cfg4-share n=4096 60-bit batch=131072 (one GPU's share of 1M/8)", 4096, 1152921504606830593, 431606828070683274, 131072, 8)]
for name, n, q, psi, B, w in CFG:
    plan = engine.Plan(n, q, psi)
    a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
    plan.time_poly_mult(a, b, c, 3)
    ms = min(plan.time_poly_mult(a, b, c, 10) for _ in range(3))
    print(f"{name:62s} {ms:8.3f} ms  {B/ms*1e3/1e6:8.2f} M polymul/s  {B*3*n*w/ms/1e6:7.0f} GB/s  frac {B*3*n*w/ms/1e6/8000:.3f}  lazy={plan.is_lazy}", flush=True)
    del a, b, c; plan.close(); torch.cuda.empty_cache()
