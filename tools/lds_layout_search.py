#!/usr/bin/env python3
"""Developer tool: search for ADDITIVE (base + register offset) LDS layouts of the fused kernels' last transpose that are
bank-conflict free in the gfx950 banking model of tests/test_lds_banks.py.

Within one wave the transpose between the second-to-last phase (register index = coefficient bits [3,6), lanes = bits
[0,3) and [6,9)) and the last phase (register index = bits [0,3), lanes = bits [3,9)) sees addresses
    A(j) = j0 + a1 * j1 + h(j2),      j = j0 + 8 j1 + 64 j2
(8-byte units).  The script enumerates a1 and h and prints the cheapest layouts with 128-bit accesses on the owning
side and with 64-bit accesses on both sides.  Result used by FusedCfg::ex_pad: a1 = 9, h = 72 j2 (one pad element
per 8 coefficients), 64-bit accesses, no conflicts."""
import itertools, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_lds_banks import conflict_degree

def cost(a1,h,b128=True):
    S=set()
    for j2 in range(8):
        for j1 in range(8):
            for j0 in range(8): S.add(j0+a1*j1+h[j2])
    if len(S)!=512: return None
    c={'w2':0,'r2':0,'w3':0,'r3':0}
    for j1 in range(8):
        addrs=[(( (l&7)+a1*j1+h[l>>3])*8) for l in range(64)]
        c['w2']+=4*conflict_degree(addrs,"write",8); c['r2']+=4*conflict_degree(addrs,"read",8)
    if b128:
        for r0 in (0,2,4,6):
            addrs=[((r0+a1*(l&7)+h[l>>3])*8) for l in range(64)]
            if any(a%16 for a in addrs): return None
            c['w3']+=8*conflict_degree(addrs,"write",16); c['r3']+=8*conflict_degree(addrs,"read",16)
    else:
        for r0 in range(8):
            addrs=[((r0+a1*(l&7)+h[l>>3])*8) for l in range(64)]
            c['w3']+=4*conflict_degree(addrs,"write",8); c['r3']+=4*conflict_degree(addrs,"read",8)
    return sum(c.values()),c,max(S)+1
for b128 in (True,False):
    best=None
    for a1 in range(8,34):
        for stride in range(7+7*a1+1, 7+7*a1+40):
            for extra in itertools.product(range(0,18,2) if b128 else range(0,17),repeat=2):
                h=[stride*j2+extra[0]*(j2&1)+extra[1]*((j2>>1)&1) for j2 in range(8)]
                r=cost(a1,h,b128)
                if r and (best is None or (r[0],r[2])<(best[0][0],best[0][2])): best=(r,a1,stride,extra)
    print("b128" if b128 else "b64",best)
