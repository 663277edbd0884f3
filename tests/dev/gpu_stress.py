"""Developer tool: randomized soak of the persistent kernels' row hand-out (fused and constant-geometry, dynamic and fixed stride, chunked
and single rows) on several streams at once; every result is compared with the fused kernel's rows computed up front.
usage: gpu_stress.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from tiny_ntt_amd import engine, numtheory
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
Q60, Q24 = 1152921504606830593, 8380417
shapes = [(4096, Q60, 431606828070683274), (2048, Q60, None), (1024, Q24, 5548360), (256, Q24, 1239911)]
batches = {4096: [1, 7, 300, 4097, 9000], 2048: [3, 515, 20001], 1024: [1, 64, 4096, 30011], 256: [5, 1000, 70001]}
variants = ["fused", "cg8_padded", "cg8_swizzled", "cg4_padded", "cg2_swizzled", "cg"]
cases = []
for n, q, psi in shapes:
    plan = engine.Plan(n, q, psi if psi else numtheory.primitive_2n_root(n, q))
    for B in batches[n]:
        a = plan.fill_lcg(B, 11, 2); b = plan.fill_lcg(B, 12, 2)
        ref = plan.poly_mult(a, b, variant="fused"); ref_t = plan.ntt_forward(a, variant="fused")
        cases.append((plan, a, b, ref, ref_t))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(4)]
t0, launches, checks = time.perf_counter(), 0, 0
while time.perf_counter() - t0 < secs:
    pending = []
    for st in streams:
        for _ in range(rng.randint(1, 6)):
            plan, a, b, ref, ref_t = rng.choice(cases)
            v = rng.choice(variants)
            with torch.cuda.stream(st):                     # the zero fill must be ordered before the kernel: same stream
                out = torch.zeros_like(a)
            if rng.random() < 0.75:
                plan.poly_mult(a, b, out=out, variant=v, stream=st); pending.append((out, ref, v, "product", a.shape))
            else:
                plan.ntt_forward(a, variant=v, out=out, stream=st); pending.append((out, ref_t, v, "cg_ntt", a.shape))
            launches += 1
    torch.cuda.synchronize()
    for out, ref, v, what, shape in pending:
        if not torch.equal(out, ref):
            print(f"MISMATCH {what} {v} shape {tuple(shape)} after {launches} launches"); sys.exit(1)
        checks += 1
print(f"gpu_stress: {launches} launches on {len(streams)} streams in {time.perf_counter() - t0:.0f} s, {checks} results compared, 0 mismatches")
