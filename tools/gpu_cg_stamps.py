"""Developer tool (diagnostic build only): where a wave of the constant-geometry product kernel spends its cycles.
Build:  CG_PARTS="4 5 6" tools/build_variant.sh stamps -DTN_CG_STAMPS=1     Run:  TINYNTT_LIB=.../libtinyntt_stamps.so python tools/gpu_cg_stamps.py [rows] [variant ...]
Per wave the kernel accumulates s_memtime deltas per phase of a product row and the time spent at workgroup barriers (cg_kernel_impl.h,
TN_CG_STAMPS) into a buffer of its own; this prints the median over waves, per row."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from tiny_ntt_amd import engine
n, q, psi = 4096, 1152921504606830593, 431606828070683274
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variants = sys.argv[2:] or ["cg8_padded", "cg4_padded"]
plan = engine.Plan(n, q, psi)
lib = engine.load_library()
lib.tn_debug_cg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]; lib.tn_debug_cg_stamps.restype = ctypes.c_size_t
a = plan.fill_lcg(B, 1, 2); b = plan.fill_lcg(B, 2, 2); c = torch.empty_like(a)
names = ["twist a,b (records from L2)", "transform a", "transform b", "pointwise + prefetch issue", "inverse transform", "untwist + store", "-", "loop top"]
for v in variants:
    for _ in range(30):                                   # settle the clock
        plan.poly_mult(a, b, variant=v, out=c, stream="plan")
    plan.synchronize()
    ms = plan.time_poly_mult(a, b, c, 5, v)
    buf = np.zeros(1 << 22, dtype=np.uint64)
    nb = lib.tn_debug_cg_stamps(buf.ctypes.data, buf.nbytes)
    st = buf[: nb // 8].reshape(-1, 12)
    st = st[st[:, 9] > 0]
    waves = st.shape[0]
    rows_per_wave = B / (waves / (4 if v.startswith("cg8") else 8 if v.startswith("cg4") else 16))
    tot = np.median(st[:, 9])
    print(f"== {v}: {ms:.3f} ms per launch, {waves} waves, ~{rows_per_wave:.0f} rows per workgroup, wave lifetime {tot:.3e} cycles (median)")
    for k in (7, 0, 1, 2, 3, 4, 5):
        m = np.median(st[:, k])
        print(f"   {names[k]:32s} {m / rows_per_wave:9.0f} cycles/row  {100 * m / tot:5.1f} %")
    bar = np.median(st[:, 8])
    print(f"   {'of which: waiting at barriers':32s} {bar / rows_per_wave:9.0f} cycles/row  {100 * bar / tot:5.1f} %")
    # in-kernel clock: s_memtime counts shader cycles, s_memrealtime the constant 100 MHz clock, both over the wave's lifetime
    clk = np.median(st[:, 9] / np.maximum(st[:, 10], 1)) * 100e6
    life = np.median(st[:, 10]) / 100e6
    print(f"   in-kernel clock {clk / 1e9:.3f} GHz (s_memtime / s_memrealtime over a wave's lifetime, median); a wave lives {life * 1e3:.3f} ms of the {ms:.3f} ms launch")
