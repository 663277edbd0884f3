"""Developer tool: keep the fused kernel running for a few seconds (for clock / power sampling with rocm-smi)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
plan = engine.Plan(n, q, psi)
B = 65536
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
plan.time_poly_mult(a, b, c, 3, "fused")
open(sys.argv[1], "w").write("go\n")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    ms = plan.time_poly_mult(a, b, c, 500, "fused")
    print(f"spin: {ms:.3f} ms/launch", flush=True)
os.remove(sys.argv[1])
