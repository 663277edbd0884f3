"""hipGraph replay of back-to-back product launches against plain launches (launch-bound shapes).
usage: gpu_graph.py cfg2|cfg3 [batch] [launches per graph]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n, q, psi = (1024, 8380417, 5548360) if cfg == "cfg2" else (4096, 1152921504606830593, 431606828070683274)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
plan = engine.Plan(n, q, psi)
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2)
ref = plan.poly_mult(a, b); torch.cuda.synchronize()
s = torch.cuda.Stream()
c = torch.zeros_like(a)
with torch.cuda.stream(s):
    for _ in range(5): plan.poly_mult(a, b, out=c, stream=s)
s.synchronize()
def timed(fn, reps):
    s.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    s.synchronize(); return (time.perf_counter() - t0) / reps
def plain():
    for _ in range(K): plan.poly_mult(a, b, out=c, stream=s)
t_plain = min(timed(plain, 5) for _ in range(3)) / K
g = torch.cuda.CUDAGraph()
c.zero_()
with torch.cuda.graph(g, stream=s):
    for _ in range(K): plan.poly_mult(a, b, out=c, stream=s)
def replay():
    with torch.cuda.stream(s): g.replay()
replay(); s.synchronize()
ok = bool(torch.equal(c, ref))
t_graph = min(timed(replay, 5) for _ in range(3)) / K
print(f"{cfg} batch {B}: plain {t_plain*1e6:.2f} us/launch, graph of {K}: {t_graph*1e6:.2f} us/launch, results equal: {ok}", flush=True)
