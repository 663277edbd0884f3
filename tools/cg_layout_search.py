#!/usr/bin/env python3
"""Developer tool: LDS layouts of the constant-geometry kernels (BASELINE config 5) scored in the gfx950 banking model of
tests/test_lds_banks.py.

Accesses of one transform at n = 4096 / 64-bit lanes, GROUP butterflies per lane-step (cg_core.h, cg_kernel_impl.h):
  W1  first-trip column writes   element pos(r1, brv(t), e)             ds_write_b64
  RD  a lane-step's R neighbours elements R t + 2p, 2p + 1              ds_read_b128
  W2  later-trip column writes   element t + e n/R                      ds_write_b64
  TW  twiddle records            record h (n >> (j+1)) + (base0 >> j)   ds_read_b128   (forward, and n/2 - that for the inverse)
Score = LDS-array cycles per wave and transform (conflict degree x cycles of the instruction), summed over the accesses.

usage: cg_layout_search.py            # scores of the shipped layouts (CgMap / cg_twmap)
       cg_layout_search.py search     # exhaustive search over padding / XOR candidates per GROUP"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_lds_banks import conflict_degree  # noqa: E402

LOGN, EB, TWB = 12, 8, 16
N = 1 << LOGN


def brv(x, bits):
    r = 0
    for i in range(bits):
        r = (r << 1) | ((x >> i) & 1)
    return r


def geom(G):
    R = 2 * G
    L = R.bit_length() - 1
    ntrips = (LOGN + L - 1) // L
    return R, L, ntrips, LOGN - (ntrips - 1) * L


def pos(G, j, T, e):
    R, L, _, _ = geom(G)
    return ((R * T) >> j) + (e & ((R >> j) - 1)) + ((e >> (L - j)) << (LOGN - j))


# ---- the shipped layouts (must mirror CgMap::at / cg_twmap in cg_core.h; tests/test_lds_banks.py checks that they do) ----
SWZ = {8: (4, 15, 9, 7, 0, 0, 16, 0, 4, 15), 4: (3, 1, 4, 7, 8, 7, 32, 0, 4, 7), 2: (2, 1, 3, 7, 7, 7, 64, 0, 4, 3),
       1: (2, 1, 4, 7, 7, 7, 64, 256, 1, 15)}          # S1 M1 S2 M2 S3 M3 CH CH2 TS TM: CgSwz<GROUP> in cg_core.h


def image_map(G, layout):
    S1, M1, S2, M2, S3, M3, CH, CH2, _, _ = SWZ[G]
    if layout == 1:
        return lambda x: x + (x // CH) * 2 + ((x // CH2) * 2 if CH2 else 0)
    if layout == 2:
        return lambda x: x ^ (((x >> S1) & M1) << 1) ^ (((x >> S2) & M2) << 1) ^ (((x >> S3) & M3) << 1)
    return lambda x: x


def table_map(G, layout):
    TS, TM = SWZ[G][8:]
    if layout == 0:
        return lambda j: j
    return lambda j: j ^ ((j >> TS) & TM)


def accesses(G, at, tw, waves=None):
    """[(kind, width, [byte address per lane], tag, cycles when conflict free)] of one forward + one inverse-style transform"""
    R, L, ntrips, r1 = geom(G)
    TP = N // R
    threads = min(TP, 1024)
    out = []
    nw = threads // 64
    for wave in (waves if waves is not None else sorted({0, 1, nw // 2, nw - 1})):
        for it in range(TP // threads):
            lanes = [wave * 64 + l + it * threads for l in range(64)]
            for e in range(R):
                out.append(("write", 8, [at(pos(G, r1, brv(t, LOGN - L), e)) * EB for t in lanes], "W1", 4))
                out.append(("write", 8, [at(t + e * TP) * EB for t in lanes], "W2", 4))
            for p in range(G):
                out.append(("read", 16, [at(R * t + 2 * p) * EB for t in lanes], "RD", 4))
            s0 = r1
            for trip in range(1, ntrips):
                if LOGN - s0 - L < 6:                      # lane-dependent twiddles: LDS table
                    k0 = N >> (s0 + 1)
                    for j in range(L):
                        for h in range(1 << j):
                            idx = [h * (N >> (j + 1)) + (((t << (L - 1)) & ~(k0 - 1)) >> j) for t in lanes]
                            out.append(("read", 16, [tw(i) * TWB for i in idx], f"TWf{trip}", 4))
                            out.append(("read", 16, [tw(N // 2 - i) * TWB for i in idx], f"TWr{trip}", 4))
                s0 += L
    return out


def score(G, at, tw, detail=False):
    tot, ideal, by = 0, 0, {}
    for kind, w, addrs, tag, cyc in accesses(G, at, tw):
        d = conflict_degree(addrs, kind, w)
        tot += d * cyc; ideal += cyc
        by[tag] = max(by.get(tag, 1), d)
    return (tot, ideal, by) if detail else tot


def injective(f, count):
    s = {f(x) for x in range(count)}
    return len(s) == count


def main():
    if len(sys.argv) < 2:
        for G in (1, 2, 4, 8):
            for layout, name in ((0, "linear"), (1, "padded"), (2, "swizzled")):
                tot, ideal, by = score(G, image_map(G, layout), table_map(G, layout), True)
                print(f"GROUP {G} {name:9s} array cycles {tot:6d} (conflict free: {ideal})  worst degree per access: {by}")
        return
    for G in (1, 2, 4, 8):
        R = 2 * G
        best = None
        cands = []
        for ch in (4, 8, 16, 32, 64, 128):
            for pad in (2, 4, 6, 8):
                cands.append((f"pad {pad} per {ch}", lambda x, ch=ch, pad=pad: x + (x // ch) * pad))
                for ch2 in (256, 512, 1024):
                    for pad2 in (2, 4, 8):
                        cands.append((f"pad {pad} per {ch} + {pad2} per {ch2}", lambda x, ch=ch, pad=pad, ch2=ch2, pad2=pad2: x + (x // ch) * pad + (x // ch2) * pad2))
        for s in range(2, 10):
            for m in (3, 7, 15, 31):
                cands.append((f"x ^ ((x>>{s})&{m})<<1", lambda x, s=s, m=m: x ^ (((x >> s) & m) << 1)))
                for s2 in range(s + 1, 11):
                    for m2 in (1, 3, 7, 15):
                        cands.append((f"x ^ ((x>>{s})&{m})<<1 ^ ((x>>{s2})&{m2})<<1",
                                      lambda x, s=s, m=m, s2=s2, m2=m2: x ^ (((x >> s) & m) << 1) ^ (((x >> s2) & m2) << 1)))
        tws = [("j", lambda j: j)]
        for s in range(1, 8):
            for m in (1, 3, 7, 15):
                tws.append((f"j ^ ((j>>{s})&{m})", lambda j, s=s, m=m: j ^ ((j >> s) & m)))
        # the image and the table are independent: search them separately
        bi = None
        for name, f in cands:
            if not injective(f, N) or max(f(x) for x in range(0, N, 1)) >= 2 * N:
                continue
            sc = sum(conflict_degree(a, k, w) * c for k, w, a, tag, c in accesses(G, f, lambda j: j) if not tag.startswith("TW"))
            if bi is None or sc < bi[0]:
                bi = (sc, name); print("GROUP", G, "image", bi, flush=True)
        bt = None
        for name, f in tws:
            if not injective(f, N // 2 + 1):
                continue
            sc = sum(conflict_degree(a, k, w) * c for k, w, a, tag, c in accesses(G, lambda x: x, f) if tag.startswith("TW"))
            if bt is None or sc < bt[0]:
                bt = (sc, name); print("GROUP", G, "table", bt, flush=True)


if __name__ == "__main__":
    main()
