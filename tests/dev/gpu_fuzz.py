"""One-off wide parity fuzz (developer tool, oracle as checker): every power-of-two n the library accepts x moduli of every bit
length 14..62 (the prime just below 2^k, and one far from a power of two) x every kernel variant, random rows incl. unreduced
words, q-1 rows and zero rows, against the CPU oracle.  usage: gpu_fuzz.py [seed] [max_seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import is_prime, Oracle
from tiny_ntt_amd import engine, numtheory

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
orc = Oracle()
t0 = time.time()
rng = np.random.default_rng(seed)
cases = bad = 0
lazy_bits, canon_bits = set(), set()
for n in [int(x) for x in os.environ.get("FUZZ_N", "8192,4096,2048,1024,512,256,64,8").split(",")]:
    ks = range(62, 13, -1) if n <= 1024 else (62, 61, 60, 58, 57, 55, 52, 50, 47, 44, 41, 36, 33, 32, 31, 29, 26, 23, 20)
    for k in ks:
        for limit in (2 ** k, int(0.63 * 2 ** k)):
            if time.time() - t0 > budget: break
            if limit < 4 * n: continue
            q = (limit - 2) // (2 * n) * (2 * n) + 1                  # largest prime = 1 (mod 2n) below limit, if any above 2n
            while q > 2 * n and not is_prime(q): q -= 2 * n
            if q <= 2 * n: continue
            if q < 2 * n or q.bit_length() > 62: continue
            if n == 8192 and q.bit_length() <= 31: pass
            psi = numtheory.primitive_2n_root(n, q)
            try:
                plan = engine.Plan(n, q, psi)
            except Exception as e:
                print("plan refused", n, q, e, flush=True); continue
            (lazy_bits if plan.is_lazy else canon_bits).add((n, q.bit_length()))
            B = 5 if n <= 1024 else 4          # the oracle follows the reference (a modexp per butterfly): ~0.3 s per row at n = 8192
            word = 2 ** (8 * plan.elem_bytes) - 1
            a = rng.integers(0, q, (B, n), dtype=np.uint64); b = rng.integers(0, q, (B, n), dtype=np.uint64)
            a[0], b[0] = q - 1, q - 1
            a[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True); b[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True)
            a[2] = 0
            b[3] = 0; b[3, rng.integers(0, n)] = 1                      # a monomial
            ref = orc.poly_mult(a, b, q, psi)
            variants = [v for v in engine.VARIANTS if v != "auto" and (v != "fused" or plan.has_fused)]
            pick = ["fused"] * plan.has_fused + list(rng.choice([v for v in variants if v != "fused"], size=3, replace=False))
            for v in pick:
                got = plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v).astype(np.uint64)
                cases += 1
                if not np.array_equal(got, ref):
                    bad += 1
                    print("MISMATCH", n, q, v, "lazy" if plan.is_lazy else "canon", int(np.count_nonzero(got != ref)), flush=True)
            if plan.has_fused:
                A = plan.ntt_forward(a.astype(plan.dtype), variant="fused").astype(np.uint64)
                cases += 1
                if not (np.array_equal(A[1], orc.cg_ntt(a[1], plan.omega, q)) and
                        np.array_equal(plan.ntt_inverse(A.astype(plan.dtype), variant="fused").astype(np.uint64), a % np.uint64(q))):
                    bad += 1; print("MISMATCH transforms", n, q, flush=True)
            plan.close()
            print(f"  n={n} q={q} ({q.bit_length()} bits, {'lazy' if plan.is_lazy else 'canonical'}): ok so far ({cases} checks, {bad} bad)", flush=True)
    print(f"n={n}: {cases} checks so far, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz seed {seed}: {cases} checks, {bad} mismatches; lazy (n, bits): {len(lazy_bits)}, canonical: {len(canon_bits)}")
sys.exit(1 if bad else 0)
