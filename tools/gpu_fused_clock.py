"""Developer tool (diagnostic build only): the in-kernel clock of the fused product kernel (MI355X_MICROARCH.md, DVFS give-back item 6).
Build:  CG_PARTS="" tools/build_variant.sh fstamps -DTN_FUSED_STAMPS=1     Run:  TINYNTT_LIB=.../libtinyntt_fstamps.so python tools/gpu_fused_clock.py
After >= 2 s of back-to-back launches on the LCG rows, every workgroup's (s_memtime, s_memrealtime) pair before and after its row loop:
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups; beside GRBM-style launch time by HIP events."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
plan = engine.Plan(n, q, psi)
lib = engine.load_library()
lib.tn_debug_fused_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]; lib.tn_debug_fused_stamps.restype = ctypes.c_size_t
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
t0 = time.time()
while time.time() - t0 < 2.5:                      # >= 2 s of back-to-back launches on random data
    ms = plan.time_poly_mult(a, b, c, 100, "fused")
buf = np.zeros(4 * 4096, dtype=np.uint64)
nb = lib.tn_debug_fused_stamps(buf.ctypes.data, buf.nbytes)
st = buf.reshape(-1, 4); st = st[st[:, 2] > 0]
dt = (st[:, 2] - st[:, 0]).astype(np.float64); dr = (st[:, 3] - st[:, 1]).astype(np.float64)
clk = dt / dr * 0.1                                # GHz
print(f"fused product, batch {B}: {ms:.4f} ms per launch (HIP events), {st.shape[0]} workgroups stamped")
print(f"in-kernel clock: median {np.median(clk):.3f} GHz, min {clk.min():.3f}, max {clk.max():.3f}  (d s_memtime / d s_memrealtime x 100 MHz)")
print(f"workgroup row-loop time: median {np.median(dr) * 10e-6:.4f} ms, max {dr.max() * 10e-6:.4f} ms; shader cycles per workgroup: median {np.median(dt):.4e}")
