"""Developer tool: correctness spot-check + kernel time of the fused kernel at the bench shape."""
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
plan = engine.Plan(n, q, psi)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
plan.poly_mult(a, b, out=c)
cs = plan.checksum_rows(c[:4])
print("row0 checksum", int(cs[0]), "OK" if int(cs[0]) == 2710933653778106521 else "MISMATCH")
orc = ctypes.CDLL(os.path.join(ROOT, "oracle/_build/liboracle.so")); P = ctypes.POINTER(ctypes.c_uint64)
orc.tn_oracle_nwc_poly_mult_batch.argtypes = [P, P, P, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]
idx = list(range(16)) + list(range(B - 16, B))
ha = np.ascontiguousarray(plan.to_host(a[idx])); hb = np.ascontiguousarray(plan.to_host(b[idx])); ref = np.empty_like(ha)
orc.tn_oracle_nwc_poly_mult_batch(ha.ctypes.data_as(P), hb.ctypes.data_as(P), ref.ctypes.data_as(P), len(idx), n, q, psi)
print("sample rows vs oracle:", "OK" if np.array_equal(plan.to_host(c[idx]), ref) else "MISMATCH")
for v in sys.argv[2:] or ["fused"]:
    plan.time_poly_mult(a, b, c, 3, v)
    ms = min(plan.time_poly_mult(a, b, c, 10, v) for _ in range(3))
    print(f"{v}: {ms:.3f} ms  {B/ms*1e3/1e6:.2f} M polymul/s  {B*3*n*8/ms/1e6:.0f} GB/s  frac {B*3*n*8/ms/1e6/8000:.3f}")
