#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING the
reference's Python golden model (new_reference/cg_ntt.py, cg_ntt_8butterfly.py).

Run only in the build container, where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing from the reference is copied: the fixtures hold inputs and the outputs
the reference computed for them (numpy .npz, uint64 little-endian).  The GPU box
has no /root/reference; tests read only the committed .npz / .json files.

Parameterisation (SURVEY.md §7 "parameterisation trap"): the module hard-codes
N=256, Q=8380417; cg_ntt/cg_intt read N at call time and take `modulus`
explicitly, so N and Q are overridden on both modules and `modulus` is always
passed.
"""
import ast
import json
import os
import random
import sys

import numpy as np

REF = os.environ.get("TINY_NTT_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "new_reference"))
import cg_ntt as ref          # noqa: E402
import cg_ntt_8butterfly as ref8   # noqa: E402
# second, independent reference model of the UNTWISTED transforms (the RTL testbench's golden model):
# test/refs/ntt_forward_reference.py:38, ntt_inverse_reference.py:9 — parameters are passed per call
sys.path.insert(0, os.path.join(REF, "test"))
from refs.ntt_forward_reference import ntt_forward_reference as rtl_fwd    # noqa: E402
from refs.ntt_inverse_reference import ntt_inverse_reference as rtl_inv    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

PARAMS = {
    # tag: (n, q, psi)        source of the parameter set
    "P4":       (4, 7681, 1925),                                   # test/refs/fast_ntt_negacyclic_convolution.py:161-214
    "P256":     (256, 8380417, 1239911),                           # new_reference/test_cg_ntt.py:7
    "P1024":    (1024, 8380417, 5548360),                          # test/Makefile:268
    "P4096":    (4096, 8380417, 283817),                           # software_benchmark/CMakeLists.txt:5-7
    "P4096_60": (4096, 1152921504606830593, 431606828070683274),   # rtl/ntt_poly_mult.sv:16-24
}


def set_params(n, q):
    ref.N, ref.Q = n, q
    ref8.N, ref8.Q = n, q


def lcg_poly(seed, n, q):
    """Inputs of the reference C++ benchmark (make_poly): benchmark_ntt_60bit.cpp:79-87 / benchmark_ntt.cpp:82-90."""
    x, out = seed, []
    for _ in range(n):
        x = (6364136223846793005 * x + 1442695040888963407) & ((1 << 64) - 1)
        out.append((x >> 17) % q if q < (1 << 32) else x % q)
    return out


def trace_of(fn, a, omega, q):
    """Run the reference with verbose=True and collect its per-stage 'first 16' lines."""
    lines = []
    out = fn(a, omega, q, True, lines.append)
    stages = [ast.literal_eval(s.split("=", 1)[1]) for s in lines if s.startswith("  stage_out(first 16)=")]
    bitrev = [ast.literal_eval(s.split("=", 1)[1]) for s in lines if s.startswith("  bitrev(first 16)=")]
    return out, stages, bitrev[0]


def u64(x):
    return np.asarray(x, dtype=np.uint64)


def gen(tag):
    n, q, psi = PARAMS[tag]
    set_params(n, q)
    omega = pow(psi, 2, q)
    arrays, meta = {}, {"n": n, "q": q, "psi": psi, "omega": omega, "cases": [],
                        "cross_checked_with": ["new_reference/cg_ntt_8butterfly.py", "test/refs/ntt_forward_reference.py", "test/refs/ntt_inverse_reference.py"]}

    def poly_case(name, a, b):
        c = ref.nwc_poly_mult(list(a), list(b), psi)
        c8 = ref8.nwc_poly_mult_8butterfly(list(a), list(b), psi)
        assert c == c8
        arrays[name + "_a"], arrays[name + "_b"], arrays[name + "_c"] = u64(a), u64(b), u64(c)
        meta["cases"].append({"name": name, "kind": "poly_mult"})
        return c

    def ntt_case(name, a):
        A, stages, bitrev16 = trace_of(ref.cg_ntt, list(a), omega, q)
        A8, stages8, _ = trace_of(ref8.cg_ntt_8butterfly, list(a), omega, q)
        assert A == A8 and stages == stages8
        back = ref.cg_intt(list(A), omega, q)
        assert back == [x % q for x in a]
        assert rtl_fwd(list(a), N=n, q=q, psi=psi) == A and rtl_inv(list(A), N=n, q=q, psi=psi) == back   # second oracle agrees
        assert ref8.cg_intt_8butterfly(list(A), omega, q) == back
        arrays[name + "_x"], arrays[name + "_X"] = u64(a), u64(A)
        w = min(16, n)
        arrays[name + "_trace16"] = u64([s[:w] for s in stages])
        arrays[name + "_bitrev16"] = u64(bitrev16[:w])
        meta["cases"].append({"name": name, "kind": "ntt"})

    # seeded random cases, in the exact RNG call order of the reference tests
    # (test_cg_ntt.py:44-52,92-95; test_cg_ntt_8butterfly.py:49-51,60-62,108-111)
    seeds_ntt = [0, 2, 3] if tag == "P256" else [0]
    for s in seeds_ntt:
        random.seed(s)
        ntt_case(f"seed{s}_ntt", [random.randrange(q) for _ in range(n)])
    seeds_mul = [1, 4] if tag == "P256" else [1]
    for s in seeds_mul:
        random.seed(s)
        a = [random.randrange(q) for _ in range(n)]
        b = [random.randrange(q) for _ in range(n)]
        poly_case(f"seed{s}_mul", a, b)

    def sparse(vals):
        return list(vals) + [0] * (n - len(vals))

    if n >= 4:
        # hand KATs: test_cg_ntt.py:55-89, test/cocotb_tests/test_ntt_inverse.py:273-275
        if n >= 8:
            assert poly_case("kat_123x456", sparse([1, 2, 3]), sparse([4, 5, 6]))[:6] == [4, 13, 28, 27, 18, 0]
        c = poly_case("kat_123x51", sparse([1, 2, 3]), sparse([5, 1]))
        if n >= 8:
            assert c[:5] == [5, 11, 17, 3, 0]
        poly_case("kat_151x51", sparse([1, 5, 1]), sparse([5, 1]))
    # shapes of test/cocotb_tests/test_ntt_forward.py:246-453
    ntt_case("zeros_ntt", [0] * n)
    ntt_case("impulse_ntt", sparse([1]))
    ntt_case("ones_ntt", [1] * n)
    ntt_case("counting_ntt", [i % q for i in range(n)])
    ntt_case("qm1_ntt", [q - 1] * n)
    # boundary products: all q-1, and the wrap-around monomial x^(n-1) * x = -1
    poly_case("qm1_mul", [q - 1] * n, [q - 1] * n)
    xm, x1 = [0] * n, [0] * n
    xm[n - 1], x1[1] = 1, 1
    c = poly_case("wrap_mul", xm, x1)
    assert c[0] == q - 1 and not any(c[1:])
    poly_case("zeros_mul", [0] * n, [q - 1] * n)
    # the reference C++ benchmark's own inputs (checksums G1-G3 of SURVEY.md §8c)
    if tag != "P4":
        a, b = lcg_poly(1, n, q), lcg_poly(2, n, q)
        poly_case("lcg12_mul", a, b)
        twisted = [(a[i] * pow(psi, i, q)) % q for i in range(n)]
        arrays["lcg1_fwd"] = u64(ref.cg_ntt(twisted, omega, q))
        meta["cases"].append({"name": "lcg1_fwd", "kind": "forward_ntt_bench"})
    if tag == "P4":
        # literature KAT (fast_ntt_negacyclic_convolution.py:161-214)
        g, h = [1, 2, 3, 4], [5, 6, 7, 8]
        assert poly_case("lit_mul", g, h) == [7625, 7645, 2, 60]
        tw = [(g[i] * pow(psi, i, q)) % q for i in range(n)]
        assert ref.cg_ntt(tw, omega, q) == [1467, 2807, 3471, 7621]

    np.savez_compressed(os.path.join(HERE, f"golden_{tag}.npz"), **arrays)
    with open(os.path.join(HERE, f"golden_{tag}.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(tag, "cases:", len(meta["cases"]), "arrays:", len(arrays))


def hex_digests():
    """Digests (not copies) of the reference's on-disk twiddle tables rtl/twiddle_*.hex: the export
    code in tiny_ntt_amd/twiddles.py must reproduce these files byte for byte."""
    import hashlib
    files = {"twiddle_forward_4096_60bit.hex": ("P4096_60", "fwd"), "twiddle_inverse_4096_60bit.hex": ("P4096_60", "inv"),
             "twiddle_forward_4096.hex": ("P4096", "fwd"), "twiddle_inverse_4096.hex": ("P4096", "inv"),
             "twiddle_forward_1024.hex": ("P1024", "fwd"), "twiddle_inverse_1024.hex": ("P1024", "inv"),
             "twiddle_forward.hex": ("P256", "fwd"), "twiddle_inverse.hex": ("P256", "inv")}
    out = {}
    for name, (tag, kind) in files.items():
        data = open(os.path.join(REF, "rtl", name), "rb").read()
        lines = data.decode().splitlines()
        out[name] = {"tag": tag, "kind": kind, "sha256": hashlib.sha256(data).hexdigest(), "lines": len(lines),
                     "first": lines[:3], "last": lines[-1]}
    with open(os.path.join(HERE, "reference_hex_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("hex digests:", len(out))


def find_psi_outputs():
    """What the reference's parameter finder returns (scripts/find_psi.py:9-43, imported, stdout discarded) for the
    parameter sets of this repo, its own three demo sets (:60-64) and a search bound that excludes the answer."""
    import contextlib
    import io
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import find_psi as ref_find      # noqa: E402
    cases = [(n, q, 10000) for (n, q, _psi) in PARAMS.values() if n >= 256]
    cases += [(256, 7681, 10000), (512, 12289, 10000), (1024, 8380417, 100), (256, 8380417, 1754), (256, 8380417, 1753)]
    out = []
    for n, q, bound in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            psi = ref_find.find_psi(n, q, bound)
        out.append({"n": n, "q": q, "max_search": bound, "psi": psi})
    with open(os.path.join(HERE, "reference_find_psi.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("find_psi cases:", out)


def general_omega_cases():
    """cg_ntt / cg_intt accept ANY omega_n (cg_ntt.py:29-65 only evaluates the butterflies): outputs of the reference for
    omegas that are not primitive n-th roots (a non-root, a root of lower order, 0, 1) and one that is, at n = 256 and
    n = 4096 / 60-bit; and a modulus where 2n does not divide q - 1 (omega has no square root psi)."""
    arrays, meta = {}, []
    rng = random.Random(2024)
    sets = [(256, 8380417, [3, 1239911 ** 4 % 8380417, 0, 1, 1239911 ** 2 % 8380417]),
            (4096, 1152921504606830593, [5, 1, pow(431606828070683274, 2, 1152921504606830593)]),
            (16, 97, [8, 5])]                     # q = 97: 2n = 32 | 96; omega = 8 has order 16 ... and 5 is a non-residue
    for n, q, omegas in sets:
        set_params(n, q)
        x = [rng.randrange(q) for _ in range(n)]
        for w in omegas:
            name = f"n{n}_q{q}_w{w}"
            X = ref.cg_ntt(list(x), w, q)
            arrays[name + "_x"], arrays[name + "_X"] = u64(x), u64(X)
            entry = {"name": name, "n": n, "q": q, "omega": w}
            if w % q and pow(w, q - 1, q) == 1:
                arrays[name + "_inv"] = u64(ref.cg_intt(list(X), w, q))       # cg_intt(cg_ntt(x)) (= x only for primitive roots)
                entry["has_inverse"] = True
            meta.append(entry)
    np.savez_compressed(os.path.join(HERE, "golden_general_omega.npz"), **arrays)
    with open(os.path.join(HERE, "golden_general_omega.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("general omega cases:", len(meta))


def n8192_cases():
    """n = 8192 at the reference's 60-bit modulus (the largest negacyclic size it admits: 2n = 2^14 divides q - 1 exactly):
    N overridden on the reference modules (cg_ntt.py:5), psi = a primitive 16384-th root of unity.  Product, forward
    transform and first-16 stage traces of one seeded pair, plus a sparse wrap-around case."""
    n, q = 8192, 1152921504606830593
    psi = 458558429756866                                   # numtheory.primitive_2n_root(8192, q); psi^8192 == -1
    assert pow(psi, n, q) == q - 1
    omega = psi * psi % q
    set_params(n, q)
    rng = random.Random(8192)
    a = [rng.randrange(q) for _ in range(n)]; b = [rng.randrange(q) for _ in range(n)]
    arrays = {"a": u64(a), "b": u64(b), "c": u64(ref.nwc_poly_mult(a, b, psi)), "c8": u64(ref8.nwc_poly_mult_8butterfly(a, b, psi))}
    assert np.array_equal(arrays["c"], arrays["c8"])
    X, tr, _bitrev = trace_of(ref.cg_ntt, a, omega, q)
    arrays["a_ntt"], arrays["a_trace16"] = u64(X), u64(tr)
    assert ref.cg_intt(X, omega, q) == a
    xm = [0] * n; xm[n - 1] = 1; x1 = [0] * n; x1[1] = 1     # x^(n-1) * x = -1
    arrays["wrap_c"] = u64(ref.nwc_poly_mult(xm, x1, psi))
    assert arrays["wrap_c"][0] == q - 1
    np.savez_compressed(os.path.join(HERE, "golden_P8192_60.npz"), **arrays)
    with open(os.path.join(HERE, "golden_P8192_60.json"), "w") as f:
        json.dump({"n": n, "q": q, "psi": psi, "omega": omega}, f, indent=1)
    print("n=8192 cases:", sorted(arrays))


def general_psi_cases():
    """nwc_poly_mult(a, b, psi_2n) validates neither psi_2n nor the modulus (cg_ntt.py:78-92): outputs of the reference for
    psi that are NOT primitive 2n-th roots (a non-root, a root of too low an order, 0, 1), for composite and even moduli
    (where "modinv" = pow(v, q-2, q) is not an inverse), at n = 256 / 24-bit and n = 4096 / 60-bit and a few small sizes;
    8-butterfly twin asserted equal.  Also cg_ntt / cg_intt with composite and even moduli (forward needs no inverse at all)."""
    arrays, meta = {}, []
    rng = random.Random(31337)
    q24, q60 = 8380417, 1152921504606830593
    sets = [(256, q24, [3, pow(1239911, 4, q24), 0, 1, q24 - 1]),                 # non-root, order n/2, 0, 1, -1
            (4096, q60, [5, pow(431606828070683274, 2, q60)]),                    # non-root, a root of order n (psi^n = +1)
            (256, 8380416, [3, 1239911]),                                         # even composite modulus
            (64, 1000001, [10, 7]),                                               # odd composite (101 * 9901)
            (16, 2 ** 61 - 2, [12345678901234567]),                               # even, needs 64-bit lanes
            (8, 2, [1]), (4, 6, [5])]                                             # tiny moduli
    for n, q, psis in sets:
        set_params(n, q)
        a = [rng.randrange(q) for _ in range(n)]; b = [rng.randrange(q) for _ in range(n)]
        for psi in psis:
            name = f"n{n}_q{q}_psi{psi}"
            c = ref.nwc_poly_mult(list(a), list(b), psi)
            assert c == ref8.nwc_poly_mult_8butterfly(list(a), list(b), psi)
            arrays[name + "_a"], arrays[name + "_b"], arrays[name + "_c"] = u64(a), u64(b), u64(c)
            meta.append({"name": name, "kind": "poly_mult", "n": n, "q": q, "psi": psi})
    for n, q, omegas in [(256, 8380416, [3, 7]), (64, 1000001, [10]), (16, 2 ** 61 - 2, [987654321987654321])]:
        set_params(n, q)
        x = [rng.randrange(q) for _ in range(n)]
        for w in omegas:
            name = f"n{n}_q{q}_w{w}"
            X = ref.cg_ntt(list(x), w, q)
            assert X == ref8.cg_ntt_8butterfly(list(x), w, q)
            arrays[name + "_x"], arrays[name + "_X"] = u64(x), u64(X)
            arrays[name + "_inv"] = u64(ref.cg_intt(list(X), w, q))               # whatever cg_intt makes of it (no true inverse exists)
            meta.append({"name": name, "kind": "ntt", "n": n, "q": q, "omega": w})
    np.savez_compressed(os.path.join(HERE, "golden_general_psi.npz"), **arrays)
    with open(os.path.join(HERE, "golden_general_psi.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("general psi cases:", len(meta))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "n8192":
        n8192_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "general_psi":
        general_psi_cases()
        sys.exit(0)
    for tag in PARAMS:
        gen(tag)
    n8192_cases()
    hex_digests()
    find_psi_outputs()
    general_omega_cases()
    general_psi_cases()
