"""Kernel time of the fused product at one shape over a range of batch sizes (launch-bound region -> throughput region).
usage: gpu_batch_sweep.py cfg2|cfg3 [batch ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n, q, psi, eb = (1024, 8380417, 5548360, 4) if cfg == "cfg2" else (4096, 1152921504606830593, 431606828070683274, 8)
plan = engine.Plan(n, q, psi)
for B in [int(x) for x in sys.argv[2:]] or [64, 256, 1024, 4096, 16384, 65536, 262144]:
    a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
    t0 = time.perf_counter(); est = plan.time_poly_mult(a, b, c, 20, "fused")
    while time.perf_counter() - t0 < 0.15:                         # the shader clock settles ~0.1 s after idle
        plan.time_poly_mult(a, b, c, 64, "fused")
    ms = min(plan.time_poly_mult(a, b, c, max(50, min(2000, int(10.0 / est))), "fused") for _ in range(3))
    print(f"{cfg} batch {B:7d}: {ms*1e3:9.2f} us  {B/ms*1e3/1e6:8.2f} M/s  frac {B*3*n*eb/ms/1e6/8000:.3f}", flush=True)
