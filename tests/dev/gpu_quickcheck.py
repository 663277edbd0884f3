"""Ad-hoc GPU check (developer tool): every variant vs the oracle on all parameter sets + rough timings."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tiny_ntt_amd import engine
orc = ctypes.CDLL(os.path.join(ROOT, "oracle/_build/liboracle.so"))
P = ctypes.POINTER(ctypes.c_uint64)
orc.tn_oracle_nwc_poly_mult_batch.argtypes = [P, P, P, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]
params = {"P256": (256, 8380417, 1239911), "P1024": (1024, 8380417, 5548360), "P4096": (4096, 8380417, 283817),
          "P4096_60": (4096, 1152921504606830593, 431606828070683274)}
ok = True
for tag, (n, q, psi) in params.items():
    for flags in (0, engine.PLAN_FORCE_CANONICAL):
        plan = engine.Plan(n, q, psi, 0, flags)
        rng = np.random.default_rng(3)
        B = 16
        a = rng.integers(0, q, size=(B, n), dtype=np.uint64); b = rng.integers(0, q, size=(B, n), dtype=np.uint64)
        a[0] = q - 1; b[0] = q - 1
        ref = np.empty_like(a)
        orc.tn_oracle_nwc_poly_mult_batch(a.ctypes.data_as(P), b.ctypes.data_as(P), ref.ctypes.data_as(P), B, n, q, psi)
        for v in (["fused", "cg", "cg8", "cg8_padded"] if flags == 0 else ["fused"]):
            c = plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v).astype(np.uint64)
            good = np.array_equal(c, ref); ok &= good
            print(tag, "lazy" if plan.is_lazy else "canon", v, "OK" if good else f"MISMATCH {np.count_nonzero(c != ref)}", flush=True)
        plan.close()
# timing at bench shape
import torch
n, q, psi = params["P4096_60"]
plan = engine.Plan(n, q, psi)
for B in (4096, 65536):
    a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
    for v in ("fused", "cg", "cg8", "cg8_padded"):
        if v != "fused" and B > 4096: continue
        plan.time_poly_mult(a, b, c, 2, v)
        ms = plan.time_poly_mult(a, b, c, 5, v)
        print(f"B={B} {v}: {ms:.3f} ms/launch  {B/ms*1e3/1e6:.3f} M polymul/s  {B*3*n*8/ms/1e6:.1f} GB/s", flush=True)
    cs_a = plan.checksum_rows(a[:1]); cs_c = plan.checksum_rows(c[:1])
    print("row0 checksum c:", cs_c[0], "(expect 2710933653778106521)")
print("ALL OK" if ok else "FAILURES")
