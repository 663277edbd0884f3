"""Developer tool: keep the fused kernel running for a few seconds (for clock / power sampling with rocm-smi)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
# SPIN_SHAPE=p1024 / p4096 / p256: the 24-bit shapes (batch chosen to move the same bytes); default: n = 4096 / 60-bit
SHAPES = {"": (4096, 1152921504606830593, 431606828070683274, 65536), "p1024": (1024, 8380417, 5548360, 262144),
          "p4096": (4096, 8380417, 283817, 65536), "p256": (256, 8380417, 1239911, 1048576)}
n, q, psi, B = SHAPES[os.environ.get("SPIN_SHAPE", "")]
plan = engine.Plan(n, q, psi)
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
V = os.environ.get("SPIN_VARIANT", "fused")           # SPIN_VARIANT=cg8_padded ...: any engine.VARIANTS name
plan.time_poly_mult(a, b, c, 3, V)
open(sys.argv[1], "w").write("go\n")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    ms = plan.time_poly_mult(a, b, c, 500, V)
    print(f"spin: {ms:.3f} ms/launch", flush=True)
os.remove(sys.argv[1])
