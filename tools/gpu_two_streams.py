"""Developer tool: do consecutive launches overlap their tails when they alternate between two streams?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
plan = engine.Plan(n, q, psi)
B = 65536
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2)
c = [torch.empty_like(a), torch.empty_like(a)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
def run(k, two):
    for i in range(10):
        plan.poly_mult(a, b, out=c[i & 1], stream=s[(i & 1) if two else 0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        plan.poly_mult(a, b, out=c[i & 1], stream=s[(i & 1) if two else 0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
for rep in range(3):
    print(f"one stream: {run(100, False):.4f} ms/step   two streams alternating: {run(100, True):.4f} ms/step", flush=True)
print("equal outputs:", torch.equal(c[0], c[1]))
