"""Developer tool: PCIe-inclusive rate of tn_poly_mult_host (host buffers in, host buffers out) at the bench shape,
pageable vs pinned host memory, pipelined chunks vs one chunk."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
plan = engine.Plan(n, q, psi)
da = plan.fill_lcg(B, 1, 2); db = plan.fill_lcg(B, 2, 2)
a, b = plan.to_host(da), plan.to_host(db)
dc = plan.poly_mult(da, db); plan.synchronize(); ref = plan.to_host(dc)
del da, db, dc
pa, pb = torch.from_numpy(a).pin_memory().numpy(), torch.from_numpy(b).pin_memory().numpy()
pc = torch.empty((B, n), dtype=torch.int64).pin_memory().numpy().view(np.uint64)
c = np.empty_like(a)
gb = 3 * B * n * 8 / 1e9
for name, xa, xb, xc in (("pageable", a, b, c), ("pinned", pa, pb, pc)):
    for rows in (B, 0):
        plan.set_host_chunk_rows(rows)
        plan.poly_mult(xa, xb, out=xc)
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); plan.poly_mult(xa, xb, out=xc); t.append(time.perf_counter() - t0)
        ok = np.array_equal(xc, ref)
        dt = min(t)
        print(f"{name:9s} chunk_rows={'all' if rows == B else 'auto(1024)':10s} {dt*1e3:8.1f} ms  {B/dt/1e6:6.3f} M products/s  {gb/dt:6.1f} GB/s over PCIe (a+b in, c out)  bit-exact={ok}", flush=True)
