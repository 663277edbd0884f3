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
    lt = np.sort(st[:, 10]) / 100e6 * 1e3
    pct = lambda f: lt[min(len(lt) - 1, int(f * len(lt)))]
    print(f"   wave lifetimes (ms): min {lt[0]:.3f}  10% {pct(.1):.3f}  50% {pct(.5):.3f}  90% {pct(.9):.3f}  99% {pct(.99):.3f}  max {lt[-1]:.3f}")
    # where each wave ran: HW_ID (wave [3:0], simd [5:4], cu [11:8], sh [12], se [15:13]) and XCC_ID [3:0]
    hw = st[:, 11] & 0xffffffff; xcc = (st[:, 11] >> 32) & 0xf
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    life = st[:, 10] / 100e6 * 1e3
    ids, counts = np.unique(cu, return_counts=True)
    print(f"   {len(ids)} distinct CUs ran waves; waves per CU: " + ", ".join(f"{k} waves on {int((counts == k).sum())} CUs" for k in np.unique(counts)))
    for k in np.unique(counts):
        sel = np.isin(cu, ids[counts == k])
        print(f"   CUs with {k} waves: wave lifetime mean {life[sel].mean():.3f} ms (min {life[sel].min():.3f}, max {life[sel].max():.3f})")
    print("   mean lifetime per XCD: " + "  ".join(f"{int(x)}: {life[xcc == x].mean():.3f}" for x in np.unique(xcc)))
    simd = (hw >> 4) & 3
    per = {}
    for i in range(len(cu)): per.setdefault((int(cu[i]), int(simd[i])), []).append(life[i])
    ns = np.array([len(v) for v in per.values()]); 
    print("   waves per (CU, SIMD): " + ", ".join(f"{k}: {int((ns == k).sum())}" for k in np.unique(ns)) +
          ";  mean lifetime by that count: " + ", ".join(f"{k}: {np.mean([np.mean(v) for v in per.values() if len(v) == k]):.3f}" for k in np.unique(ns)))
    print(f"   in-kernel clock {clk / 1e9:.3f} GHz (s_memtime / s_memrealtime over a wave's lifetime, median); a wave lives {np.median(life):.3f} ms (median) of the {ms:.3f} ms launch")
