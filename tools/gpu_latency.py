"""Single-polynomial latency beside the reference's published numbers (reports/final-report.tex:1364-1392 CPU 433-709 us;
:1339-1342 RTL 153-383 us): mean time per launch on small batches (HIP events on the plan's stream, back-to-back launches),
the launch floor (a trivial kernel through the same path), and whole host-buffer calls.  usage: gpu_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from tiny_ntt_amd import engine
CFG = {"cfg3 n=4096 60-bit": (4096, 1152921504606830593, 431606828070683274), "cfg2 n=1024 24-bit": (1024, 8380417, 5548360)}
def per_launch(fn, iters=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6
for name, (n, q, psi) in CFG.items():
    plan = engine.Plan(n, q, psi)
    a = plan.fill_lcg(256, 1, 2); b = plan.fill_lcg(256, 2, 2); c = torch.empty_like(a)
    print(f"== {name}")
    for v in ("fused", "cg8", "cg"):
        for k in (1, 16, 256):
            plan.time_poly_mult(a[:k], b[:k], c[:k], 20, v)
            cold = plan.time_poly_mult(a[:k], b[:k], c[:k], 300, v) * 1e3           # a few ms after idle: the shader clock is still on its ramp
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.15:
                plan.time_poly_mult(a[:k], b[:k], c[:k], 256, v)
            us = plan.time_poly_mult(a[:k], b[:k], c[:k], 1000, v) * 1e3
            print(f"  poly_mult {v:5s} batch {k:3d}: {us:8.2f} us per launch after 0.15 s of such launches, {cold:8.2f} us in the first milliseconds after idle (device-resident, HIP events)")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        print(f"  launch floor (pointwise_mul of ONE row, same entry path, wall clock over back-to-back launches): {per_launch(lambda: plan.pointwise_mul(a[:1], b[:1], out=c[:1], stream=s)):.2f} us")
        print(f"  ntt_forward fused batch 1 (wall clock, back-to-back): {per_launch(lambda: plan.ntt_forward(a[:1], out=c[:1], stream=s)):.2f} us")
        print(f"  poly_mult fused batch 1 (wall clock, back-to-back):   {per_launch(lambda: plan.poly_mult(a[:1], b[:1], out=c[:1], stream=s)):.2f} us")
    ha, hb = plan.to_host(a[:1]).copy(), plan.to_host(b[:1]).copy(); hc = np.empty_like(ha)
    for _ in range(5): plan.poly_mult(ha, hb, out=hc)
    t0 = time.perf_counter()
    for _ in range(100): plan.poly_mult(ha, hb, out=hc)
    print(f"  host-buffer call, one pair (H2D + kernel + D2H + sync): {(time.perf_counter() - t0) / 100 * 1e6:.1f} us")
