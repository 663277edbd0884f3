import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tiny_ntt_amd import engine
for name, n, q, psi, B in (("n=1024 24-bit", 1024, 8380417, 5548360, 262144), ("n=1024 24-bit", 1024, 8380417, 5548360, 4096), ("n=1024 60-bit", 1024, 1152921504606830593, pow(431606828070683274, 4, 1152921504606830593), 131072)):
    plan = engine.Plan(n, q, psi)
    a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
    ref = plan.poly_mult(a[:512], b[:512], variant="cg")
    t0 = time.perf_counter(); est = plan.time_poly_mult(a, b, c, 5, "fused")
    while time.perf_counter() - t0 < 0.15: plan.time_poly_mult(a, b, c, 32, "fused")
    ms = min(plan.time_poly_mult(a, b, c, max(20, min(2000, int(10 / est))), "fused") for _ in range(3))
    ok = torch.equal(c[:512], ref)
    print(f"{name} B={B}: {ms*1e3:9.2f} us  {B/ms*1e3/1e6:8.2f} M/s frac {B*3*n*plan.elem_bytes/ms/1e6/8000:.3f} parity_vs_cg={ok}", flush=True)
    plan.close()
