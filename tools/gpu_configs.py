"""Developer tool: kernel time of every BASELINE configuration that runs on one GPU (parity is covered by tests)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import time
import torch
from tiny_ntt_amd import engine
SPIN_S = 0.15        # launches before any timing: the shader clock settles ~0.1 s after idle (short launches would otherwise be timed on the ramp)
CFG = [("cfg2 n=1024 24-bit batch=4096", 1024, 8380417, 5548360, 4096, 4),
       ("n=1024 24-bit batch=262144", 1024, 8380417, 5548360, 262144, 4),
       ("n=4096 24-bit batch=65536", 4096, 8380417, 283817, 65536, 4),
       ("n=256 24-bit batch=1048576", 256, 8380417, 1239911, 1048576, 4),
       ("cfg3 n=4096 60-bit batch=65536", 4096, 1152921504606830593, 431606828070683274, 65536, 8),
       ("cfg4-share n=4096 60-bit batch=131072 (one GPU's share of 1M/8)", 4096, 1152921504606830593, 431606828070683274, 131072, 8)]
for name, n, q, psi, B, w in CFG:
    plan = engine.Plan(n, q, psi)
    a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
    t0 = time.perf_counter(); est = plan.time_poly_mult(a, b, c, 3)
    while time.perf_counter() - t0 < SPIN_S:
        plan.time_poly_mult(a, b, c, 32)
    iters = max(10, min(2000, int(20.0 / est)))                       # >= 20 ms per timing
    ms = min(plan.time_poly_mult(a, b, c, iters) for _ in range(3))
    print(f"{name:62s} {ms:8.3f} ms  {B/ms*1e3/1e6:8.2f} M polymul/s  {B*3*n*w/ms/1e6:7.0f} GB/s  frac {B*3*n*w/ms/1e6/8000:.3f}  lazy={plan.is_lazy}", flush=True)
    del a, b, c; plan.close(); torch.cuda.empty_cache()

# standalone transforms (NTTs/s) at the benchmark shape
n, q, psi = 4096, 1152921504606830593, 431606828070683274
plan = engine.Plan(n, q, psi)
B = 65536
x = plan.fill_lcg(B, 1, 2); y = torch.empty_like(x)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("cg_ntt fused", lambda: plan.ntt_forward(x, variant="fused", out=y)), ("cg_intt fused", lambda: plan.ntt_inverse(x, variant="fused", out=y)),
                 ("twist+ntt fused", lambda: plan.twisted_ntt_forward(x, variant="fused", out=y)), ("cg_ntt cg", lambda: plan.ntt_forward(x[:8192], variant="cg", out=y[:8192]))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < SPIN_S:
        for _ in range(8): fn()
        torch.cuda.synchronize()
    ev0.record()
    for _ in range(40): fn()
    ev1.record(); torch.cuda.synchronize()
    rows = 8192 if name.endswith(" cg") else B
    ms = ev0.elapsed_time(ev1) / 40
    print(f"{name:18s} {ms:8.3f} ms  {rows/ms*1e3/1e6:8.2f} M NTT/s  {rows*2*n*8/ms/1e6:7.0f} GB/s (2nw bytes)", flush=True)

# measured HBM copy bandwidth on this box ("also record measured copy bandwidth"): 2 GiB device-to-device
src = torch.empty(2 << 30, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
dst.copy_(src); torch.cuda.synchronize()
ev0.record()
for _ in range(10): dst.copy_(src)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 10
print(f"device copy 2 GiB   {ms:8.3f} ms  {2 * (2 << 30) / ms / 1e6:7.0f} GB/s read+write  ({2 * (2 << 30) / ms / 1e6 / 8000:.2f} of the 8 TB/s peak)", flush=True)
