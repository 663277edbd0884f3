#!/usr/bin/env python3
"""Developer tool: exhaustive search for the XOR swizzle of the constant-geometry kernels' LDS image (BASELINE config 5).

Candidates x ^ (((x >> s) & m) << 1) [^ (((x >> s2) & m2) << 1)] keep the pairs (2i, 2i+1) adjacent and 16-byte aligned;
each is scored on every access pattern of the sweep in the gfx950 banking model of tests/test_lds_banks.py: the
128-bit pair reads, the output streams i / i + n/2 (64-bit at GROUP 1, 128-bit stores of two neighbours at GROUP >= 2),
the bit-reversed 128-bit scatter of the loads and the 128-bit linear read-out, for GROUP in {1, 2, 4, 8}.
Result used by CgMap (kernels.hip): s = 4, m = 15, s2 = 9, m2 = 7 — conflict free except GROUP 2's pair reads (2-way)."""
import itertools, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_lds_banks import conflict_degree
n=4096; logn=12; EB=8
def brv(x,bits):
    r=0
    for i in range(bits): r=(r<<1)|((x>>i)&1)
    return r
def accesses2(G, sw):
    pairs=n//2; out=[]
    for wave in (0,3,5):
        lanes=[wave*64+l for l in range(64)]
        for g in range(G):
            out.append(("read",16,[sw(2*(t*G+g))*EB for t in lanes],"pair_read"))
        if G==1:
            out.append(("write",8,[sw(t)*EB for t in lanes],"w_lo")); out.append(("write",8,[sw(t+pairs)*EB for t in lanes],"w_hi"))
        else:
            for g in range(0,G,2):
                out.append(("write",16,[sw(t*G+g)*EB for t in lanes],"w_lo128")); out.append(("write",16,[sw(t*G+g+pairs)*EB for t in lanes],"w_hi128"))
        out.append(("write",16,[sw(2*brv(t,logn-1))*EB for t in lanes],"scatter128"))
        out.append(("read",16,[sw(2*t)*EB for t in lanes],"lin_read128"))
    return out
best=None
for s in range(3,10):
    for m in (7,15,31,63):
        for s2 in (None,6,7,8,9,10):
          for m2 in (7,15):
            def sw(x,s=s,m=m,s2=s2,m2=m2):
                y=x ^ (((x>>s)&m)<<1)
                if s2: y ^= (((x>>s2)&m2)<<1)
                return y
            if len({sw(x) for x in range(n)})!=n or max(sw(x) for x in range(n))>=n: continue
            res={}
            for G in (1,2,4,8):
                for kind,w,addrs,tag in accesses2(G,sw):
                    res[(G,tag)]=max(res.get((G,tag),1),conflict_degree(addrs,kind,w))
            bad={k:v for k,v in res.items() if v>1}
            score=sum(v-1 for v in bad.values())
            if best is None or score<best[0]:
                best=(score,s,m,s2,m2,bad); print(best)
