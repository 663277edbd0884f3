"""CPU stepping of the product kernels' per-thread code (tests/emu) against golden vectors and the
oracle: validates index maps, LDS layouts, table semantics, lazy-reduction bounds and the modular
primitives of tiny_ntt_amd/csrc/*.h without a GPU."""
import os
import random

import numpy as np
import pytest

from conftest import PARAMS

FUSED_TAGS = ["P256", "P1024", "P4096", "P4096_60"]


@pytest.mark.parametrize("tag", FUSED_TAGS)
@pytest.mark.parametrize("canonical", [False, True])
def test_fused_emulation_matches_golden(emu, golden, tag, canonical):
    g = golden(tag)
    for name in g.cases("poly_mult"):
        c = emu.fused(g.n, g.q, g.psi, g[name + "_a"], g[name + "_b"], canonical=canonical)
        assert np.array_equal(c, g[name + "_c"]), name


@pytest.mark.parametrize("tag", FUSED_TAGS)
def test_fused_emulation_random_and_unreduced_vs_oracle(emu, oracle, tag):
    n, q, psi = PARAMS[tag]
    rng = np.random.default_rng(2)
    word = 2 ** 32 - 1 if q < 2 ** 31 else 2 ** 64 - 1
    a = rng.integers(0, q, (6, n), dtype=np.uint64); b = rng.integers(0, q, (6, n), dtype=np.uint64)
    a[0] = q - 1; b[0] = q - 1
    a[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True); b[1] = word      # any word value is taken mod q
    a[2] = word; b[2] = word
    ref = oracle.poly_mult(a, b, q, psi)
    for canonical in (False, True):
        assert np.array_equal(emu.fused(n, q, psi, a, b, canonical=canonical), ref)


def test_fused_emulation_with_promised_canonical_inputs(emu, oracle, golden):
    """TN_PLAN_CANONICAL_INPUTS: the n = 4096 / 60-bit product kernel with the bound schedule started from q (no load folds):
    golden products and the extreme canonical values (all q - 1, all 0, random)."""
    g = golden("P4096_60")
    n, q, psi = g.n, g.q, g.psi
    for name in g.cases("poly_mult"):
        a, b = g[name + "_a"], g[name + "_b"]
        if a.max() < q and b.max() < q:
            assert np.array_equal(emu.fused(n, q, psi, a, b, promised_canonical_inputs=True), g[name + "_c"]), name
    rng = np.random.default_rng(12)
    a = rng.integers(0, q, (4, n), dtype=np.uint64); b = rng.integers(0, q, (4, n), dtype=np.uint64)
    a[0] = q - 1; b[0] = q - 1; a[1] = q - 1; b[2] = 0
    assert np.array_equal(emu.fused(n, q, psi, a, b, promised_canonical_inputs=True), oracle.poly_mult(a, b, q, psi))
    assert emu.fused(256, 8380417, 1239911, a[0, :256] % np.uint64(8380417), a[0, :256] % np.uint64(8380417), promised_canonical_inputs=True) is None


def test_lazy_policy_selected_for_reference_moduli(emu):
    for tag in FUSED_TAGS:
        assert emu.lib.emu_is_lazy(*PARAMS[tag]) == 1, tag
    # a generic 30-bit NTT prime far from a power of two must fall back to canonical arithmetic
    q, n = 754974721, 256                      # 45 * 2^24 + 1
    g = next(pow(x, (q - 1) // (2 * n), q) for x in range(2, 50) if pow(pow(x, (q - 1) // (2 * n), q), n, q) == q - 1)
    assert emu.lib.emu_is_lazy(n, q, g) == 0


def test_fused_emulation_generic_modulus_canonical(emu, oracle):
    q, n = 754974721, 256
    psi = next(pow(x, (q - 1) // (2 * n), q) for x in range(2, 50) if pow(pow(x, (q - 1) // (2 * n), q), n, q) == q - 1)
    rng = np.random.default_rng(9)
    a = rng.integers(0, q, (3, n), dtype=np.uint64); b = rng.integers(0, q, (3, n), dtype=np.uint64)
    assert np.array_equal(emu.fused(n, q, psi, a, b), oracle.poly_mult(a, b, q, psi))
    # 61-bit prime (k * 2^20 + 1): 16q > 2^64, so 64-bit lanes must use the canonical policy
    q = 2305843009196916737
    psi = next(pow(x, (q - 1) // (2 * n), q) for x in range(2, 200) if pow(pow(x, (q - 1) // (2 * n), q), n, q) == q - 1)
    assert emu.lib.emu_is_lazy(n, q, psi) == 0
    a = rng.integers(0, q, (2, n), dtype=np.uint64); b = rng.integers(0, q, (2, n), dtype=np.uint64)
    a[0] = q - 1; b[0] = q - 1
    assert np.array_equal(emu.fused(n, q, psi, a, b), oracle.poly_mult(a, b, q, psi))


@pytest.mark.parametrize("tag", ["P4", "P256", "P1024", "P4096", "P4096_60"])
def test_cg_emulation_matches_golden(emu, golden, tag):
    g = golden(tag)
    for name in g.cases("poly_mult"):
        assert np.array_equal(emu.cg(g.n, g.q, g.psi, 2, g[name + "_a"], g[name + "_b"]), g[name + "_c"]), name
    for name in g.cases("ntt"):
        out, tr = emu.cg(g.n, g.q, g.psi, 0, g[name + "_x"], trace=True)
        assert np.array_equal(out, g[name + "_X"]), name
        assert np.array_equal(tr[:, :min(16, g.n)], g[name + "_trace16"]), name
        assert np.array_equal(emu.cg(g.n, g.q, g.psi, 1, g[name + "_X"]), g[name + "_x"] % np.uint64(g.q)), name
    if tag != "P4":
        assert np.array_equal(emu.cg(g.n, g.q, g.psi, 3, g["lcg12_mul_a"]), g["lcg1_fwd"])


@pytest.mark.parametrize("tag", ["P4", "P256", "P1024", "P4096", "P4096_60"])
def test_cg_trips_emulation_matches_golden_and_oracle(emu, oracle, golden, tag):
    """The constant-geometry kernels' multi-stage trips (cg_core.h: log2(2 GROUP) stages per LDS round trip, first trip fed
    straight from the bit-reversed load, one LDS twiddle table read backwards for the inverse) stepped lane-step by
    lane-step: every lane grouping x LDS layout x arithmetic x twiddle source, products, transforms and per-stage traces."""
    g = golden(tag)
    n, q, psi = g.n, g.q, g.psi
    rng = np.random.default_rng(3)
    word = 2 ** 32 - 1 if q < 2 ** 31 else 2 ** 64 - 1
    a = rng.integers(0, word, n, dtype=np.uint64, endpoint=True); b = np.full(n, word, dtype=np.uint64)       # any word is taken mod q
    ref = oracle.poly_mult(a[None], b[None], q, psi)[0]
    full_trace = {name: emu.cg(n, q, psi, 0, g[name + "_x"], trace=True)[1] for name in g.cases("ntt")}
    ran = 0
    for group in (1, 2, 4, 8):
        for layout in (0, 1, 2):
            for am in (0, 1, 2, 3):
                if emu.cgm(n, q, psi, 0, a, group=group, layout=layout, am=am) is None:
                    continue                                  # log2 n < log2(2 group), or split arithmetic on a Shoup plan
                ran += 1
                for flags in (0, 1, 2, 3):
                    assert np.array_equal(emu.cgm(n, q, psi, 2, a, b, group=group, layout=layout, am=am, flags=flags), ref), (group, layout, am, flags)
                for name in g.cases("poly_mult"):
                    assert np.array_equal(emu.cgm(n, q, psi, 2, g[name + "_a"], g[name + "_b"], group=group, layout=layout, am=am), g[name + "_c"]), name
                for name in g.cases("ntt"):
                    if am < 2:                                # traces need canonical stages
                        out, tr = emu.cgm(n, q, psi, 0, g[name + "_x"], group=group, layout=layout, am=am, flags=1, trace=True)
                        assert np.array_equal(out, g[name + "_X"]) and np.array_equal(tr, full_trace[name]), (name, group, layout, am)
                        assert np.array_equal(tr[:, :min(16, n)], g[name + "_trace16"]), name
                    assert np.array_equal(emu.cgm(n, q, psi, 0, g[name + "_x"], group=group, layout=layout, am=am), g[name + "_X"]), name
                    assert np.array_equal(emu.cgm(n, q, psi, 1, g[name + "_X"], group=group, layout=layout, am=am), g[name + "_x"] % np.uint64(q)), name
                if tag != "P4":
                    assert np.array_equal(emu.cgm(n, q, psi, 3, g["lcg12_mul_a"], group=group, layout=layout, am=am), g["lcg1_fwd"])
    assert ran >= (6 if tag == "P4" else 12)


def test_cg_trips_emulation_every_size(emu, oracle):
    """n = 4 ... 8192 (partial first trips of every length, single-trip transforms) x word sizes."""
    from conftest import ntt_prime_below
    for logn in range(2, 14):
        n = 1 << logn
        for limit in (2 ** 23, 2 ** 31 - 1, 2 ** 45, 2 ** 60):
            q = ntt_prime_below(limit, n)
            psi = next(p for p in (pow(x, (q - 1) // (2 * n), q) for x in range(2, 500)) if pow(p, n, q) == q - 1)
            rng = np.random.default_rng(logn)
            a = rng.integers(0, q, n, dtype=np.uint64); b = rng.integers(0, q, n, dtype=np.uint64)
            ref = oracle.poly_mult(a[None], b[None], q, psi)[0]
            for group in (1, 2, 4, 8):
                for am in (0, 1, 2, 3):
                    c = emu.cgm(n, q, psi, 2, a, b, group=group, layout=2 if group & 5 else 1, am=am, flags=1)
                    assert c is None or np.array_equal(c, ref), (n, q, group, am)


EDGE64 = [0, 1, 2, 2 ** 32 - 1, 2 ** 32, 2 ** 60 - 1, 2 ** 60, 2 ** 63, 2 ** 64 - 1]


def test_mul_tw64_exact_for_any_word(emu):
    rnd = random.Random(1)
    for q in (PARAMS["P4096_60"][1], 2305843009213693951 - 2 ** 20 + 2 ** 20, 4611686018326724609):   # 60-bit, 2^61-1, ~2^62
        ws = [0, 1, q - 1, q // 2] + [rnd.randrange(q) for _ in range(300)]
        xs = EDGE64 + [q - 1, q, q + 1, 2 * q, 3 * q] + [rnd.randrange(2 ** 64) for _ in range(300)]
        for w in ws:
            for a in xs[:40] if w > 3 else xs:
                a %= 2 ** 64
                assert emu.lib.emu_mul_tw64(a, w, q) == a * w % q
                lazy = emu.lib.emu_mul_tw64_lazy(a, w, q)
                assert lazy < 4 * q and lazy % q == a * w % q


def test_mul_tw32_and_barrett32(emu):
    rnd = random.Random(2)
    for q in (8380417, 7681, 754974721, 2147483647):
        vals = [0, 1, q - 1, q // 2] + [rnd.randrange(q) for _ in range(200)]
        for w in vals[:60]:
            for a in [0, 1, q - 1, q, 2 ** 32 - 1, 2 ** 31] + [rnd.randrange(2 ** 32) for _ in range(60)]:
                assert emu.lib.emu_mul_tw32(a, w, q) == a * w % q
        for a in vals:
            for b in vals[:40]:
                assert emu.lib.emu_barrett32(a, b, q) == a * b % q


def test_pointwise_lazy64_any_words(emu):
    """Split-and-fold product of the lazy 64-bit policy (mulmod_solinas_lazy): exact mod q and < 2q for any word
    times any value below 14q, for every admissible q = 2^k - c; inadmissible (k, c) are reported by h_pw_fast_ok."""
    rnd = random.Random(33)
    qs = [PARAMS["P4096_60"][1], 2 ** 59 - 2 ** 15 + 1, 2 ** 50 - 2 ** 13 + 1 - 0, 2 ** 40 - 87, 2 ** 33 - 9, 2 ** 60 - 93, 2 ** 60 - (2 ** 28 - 57)]
    for q in qs:
        assert emu.lib.emu_pw_fast_ok(q) == 1, q
        top = min(2 ** 64, 16 * q)
        vals = EDGE64 + [q - 1, q, q + 1, 2 * q, 15 * q, top - 1, 2 ** (q.bit_length()) - 1, 2 ** (q.bit_length())]
        vals = [v % 2 ** 64 for v in vals] + [rnd.randrange(2 ** 64) for _ in range(300)]
        bvals = [v for v in vals if v < 14 * q] + [14 * q - 1]                      # second operand: below (LIMIT-2) q
        for a in vals[:40]:
            for b in bvals:
                r = emu.lib.emu_pointwise_lazy64(a, b, q)
                assert r < 2 * q and r % q == a * b % q, (q, a, b)
    for q in (2 ** 60 - 2 ** 30 + 1, 2 ** 36 - 2 ** 17 - 1, 2 ** 61 - 1):        # c too large for the bounds / k > 60
        assert emu.lib.emu_pw_fast_ok(q) == 0, q


def test_split_constant_product_is_exact_and_bounded(emu):
    """mul_sp_acc (modarith.h): u + a*w as six 32x32+64 multiply-adds on the split record {wlo, whi, xlo, xhi}.  For ANY
    64-bit a the result is the integer u + t' with t' == a*w (mod q) and t' below the bound the schedule uses
    (2^(k+1) + (a >> 32) 2^p + 2^(k+1) + 2^32 * 2c), for the reference's modulus and other admissible q = 2^k - c."""
    rnd = random.Random(44)
    for q in (PARAMS["P4096_60"][1], 2 ** 57 - 2 ** 11 + 1 - 0, 2 ** 52 - 2 ** 13 * 5 + 1, 2 ** 47 - 2 ** 9 * 3 + 1):
        k = q.bit_length(); c = 2 ** k - q; p = k - 31
        ws = [0, 1, 2, q - 1, q - 2, q // 2, 2 ** p - 1, 2 ** p, 2 ** (k - 1)] + [rnd.randrange(q) for _ in range(200)]
        avals = EDGE64 + [q - 1, q, 2 ** k - 1, 2 ** k, 2 ** 64 - 2 ** 32, 2 ** 32 * (2 ** 32 - 1) + 1] + [rnd.randrange(2 ** 64) for _ in range(200)]
        for w in ws:
            for a in avals[:30] if w > 2 else avals:
                a %= 2 ** 64
                tmax = 2 ** (k + 1) + (a >> 32) * 2 ** p + 2 ** (k + 1) + 2 ** 32 * 2 * c
                for u in (0, 1, 2 ** 64 - 1 - tmax, rnd.randrange(2 ** 64 - tmax)):
                    r = emu.lib.emu_mul_sp_acc(u, a, w, q)
                    assert r >= u and r - u < tmax and (r - u - a * w) % q == 0, (q, u, a, w)
    # the bound schedule of every fused shape replays exactly for the reference's modulus, and folds stay rare
    for logn in (8, 9, 10, 11, 12, 13):
        assert emu.lib.emu_split_sched_ok(logn, PARAMS["P4096_60"][1]) == 1
        fwd, inv, fout = (emu.lib.emu_split_sched_stat(logn, i) for i in range(3))
        assert fwd <= 2 * logn and inv <= 3 * logn and fout <= 14 * 4096, (logn, fwd, inv, fout)
    assert emu.lib.emu_split_sched_ok(12, 2 ** 60 - 2 ** 30 + 1) == 0          # c too large: not lazy
    assert emu.lib.emu_split_sched_ok(12, 2 ** 41 - 21 * 2 ** 13 + 1) == 0      # 2^33 c is not << 2^k


def test_barrett64_boundaries(emu):
    rnd = random.Random(3)
    for q in (PARAMS["P4096_60"][1], 4611686018326724609, 2 ** 61 - 1, 1099511627689):
        vals = [0, 1, 2, q - 1, q - 2, q // 2, q // 2 + 1] + [rnd.randrange(q) for _ in range(400)]
        for a in vals[:80]:
            for b in vals:
                assert emu.lib.emu_barrett64(a, b, q) == a * b % q


def test_fold_bounds(emu):
    q60, q23 = PARAMS["P4096_60"][1], 8380417
    rnd = random.Random(4)
    for x in EDGE64 + [rnd.randrange(2 ** 64) for _ in range(2000)]:
        r = emu.lib.emu_fold64(x, q60)
        assert r % q60 == x % q60 and r < 2 * q60
    for x in [0, 1, 2 ** 23, 2 ** 32 - 1, 2 ** 31] + [rnd.randrange(2 ** 32) for _ in range(2000)]:
        r = emu.lib.emu_fold32(x, q23)
        assert r % q23 == x % q23 and r < 2 * q23


@pytest.mark.parametrize("n", [512, 2048])
@pytest.mark.parametrize("q", [8380417, 1152921504606830593])
def test_fused_emulation_other_sizes(emu, oracle, n, q):
    """n = 512 / 2048 (8 coefficients per thread, partial last phase): psi found for each (n, q)."""
    from tiny_ntt_amd import numtheory
    psi = numtheory.primitive_2n_root(n, q)
    rng = np.random.default_rng(n)
    a = rng.integers(0, q, (4, n), dtype=np.uint64); b = rng.integers(0, q, (4, n), dtype=np.uint64)
    a[0], b[0] = q - 1, q - 1
    ref = oracle.poly_mult(a, b, q, psi)
    for canonical in (False, True):
        assert np.array_equal(emu.fused(n, q, psi, a, b, canonical=canonical), ref)


@pytest.mark.parametrize("tag", ["P256", "P1024", "P4096", "P4096_60"])
@pytest.mark.parametrize("canonical", [False, True])
def test_fused_standalone_transforms_emulation_matches_golden(emu, golden, tag, canonical):
    g = golden(tag)
    for name in g.cases("ntt"):
        x, X = g[name + "_x"], g[name + "_X"]
        assert np.array_equal(emu.fused_ntt(g.n, g.q, g.psi, 1, x, canonical), X), name                        # cg_ntt
        assert np.array_equal(emu.fused_ntt(g.n, g.q, g.psi, 2, X, canonical), x % np.uint64(g.q)), name       # cg_intt
    assert np.array_equal(emu.fused_ntt(g.n, g.q, g.psi, 0, g["lcg12_mul_a"], canonical), g["lcg1_fwd"])        # forward_ntt_bench


@pytest.mark.parametrize("n", [256, 1024])
def test_fused_emulation_modulus_sweep(emu, oracle, n):
    """CPU stepping of the fused kernel over moduli of 20..62 bits, lazy (q = 2^k - c) and canonical policies
    (the GPU twin of this sweep is test_gpu_parity.py::test_modulus_sweep_every_word_size_both_policies)."""
    from conftest import ntt_prime_below
    from tiny_ntt_amd import numtheory
    lazy = set()
    for k in (20, 26, 31, 32, 33, 36, 41, 47, 52, 57, 60, 61, 62):
        for limit in (2 ** k, int(0.71 * 2 ** k)):
            q = ntt_prime_below(limit, n)
            psi = numtheory.primitive_2n_root(n, q)
            if emu.lib.emu_is_lazy(n, q, psi) == 1:
                lazy.add(q.bit_length())
            rng = np.random.default_rng(k)
            word = 2 ** 32 - 1 if q < 2 ** 31 else 2 ** 64 - 1
            a = rng.integers(0, q, (3, n), dtype=np.uint64); b = rng.integers(0, q, (3, n), dtype=np.uint64)
            a[0], b[0] = q - 1, q - 1
            a[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True); b[1] = word
            assert np.array_equal(emu.fused(n, q, psi, a, b), oracle.poly_mult(a, b, q, psi)), (n, q)
            X = emu.fused_ntt(n, q, psi, 1, a[1])
            assert np.array_equal(X, oracle.cg_ntt(a[1], psi * psi % q, q)), (n, q)
            assert np.array_equal(emu.fused_ntt(n, q, psi, 2, X), a[1] % np.uint64(q)), (n, q)
    # 64-bit lanes: the split-constant product needs 2^33 c << 2^k (h_split_sched_ok), which no 41-bit NTT prime meets
    assert {26, 47, 52, 57, 60} <= lazy and not ({61, 62} & lazy)


@pytest.mark.parametrize("tag", ["P256", "P1024", "P4096", "P4096_60"])
@pytest.mark.parametrize("canonical", [False, True])
def test_fused_cyclic_product_emulation(emu, oracle, tag, canonical):
    """The fused product kernel on the x^n - 1 twiddle tables = python_poly_mult (test_ntt_poly_mult.py:38-43):
    cg_ntt, cg_ntt, pointwise, cg_intt with omega = psi^2 and no twist."""
    n, q, psi = PARAMS[tag]
    omega = psi * psi % q
    rng = np.random.default_rng(7)
    word = 2 ** 32 - 1 if q < 2 ** 31 else 2 ** 64 - 1
    a = rng.integers(0, q, (3, n), dtype=np.uint64); b = rng.integers(0, q, (3, n), dtype=np.uint64)
    a[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True); b[1] = word          # any word is taken mod q
    a[2] = 0; a[2, n - 1] = 1; b[2] = 0; b[2, 1] = 1                                        # x^(n-1) * x = +1 (no sign flip)
    got = emu.fused(n, q, psi, a, b, canonical, cyclic=True)
    for r in range(3):
        A, B = oracle.cg_ntt(a[r], omega, q), oracle.cg_ntt(b[r], omega, q)
        C = np.array([int(x) * int(y) % q for x, y in zip(A, B)], dtype=np.uint64)
        assert np.array_equal(got[r], oracle.cg_intt(C, omega, q)), (tag, r)
    assert got[2][0] == 1 and not got[2][1:].any()


@pytest.mark.parametrize("n", [512, 2048])
def test_fused_standalone_transforms_other_sizes(emu, oracle, n):
    from tiny_ntt_amd import numtheory
    for q in (8380417, 1152921504606830593):
        psi = numtheory.primitive_2n_root(n, q)
        omega = psi * psi % q
        rng = np.random.default_rng(n)
        word = 2 ** 32 - 1 if q < 2 ** 31 else 2 ** 64 - 1
        x = rng.integers(0, word, n, dtype=np.uint64, endpoint=True)        # any word value is taken mod q
        X = oracle.cg_ntt(x, omega, q)
        assert np.array_equal(emu.fused_ntt(n, q, psi, 1, x), X)
        assert np.array_equal(emu.fused_ntt(n, q, psi, 2, X), x % np.uint64(q))


def test_n8192_60bit_emulation_and_oracle_match_reference_golden(emu, oracle):
    """n = 8192 at the reference's 60-bit modulus (vectors: tests/golden/make_golden.py n8192_cases, N overridden on the
    reference modules): the CPU oracle and the CPU stepping of the fused kernels (five register phases, the fourth with
    vector-loaded twiddles) both reproduce the reference's product, forward transform and stage traces."""
    import json
    import os
    from conftest import GOLDEN
    meta = json.load(open(os.path.join(GOLDEN, "golden_P8192_60.json")))
    g = np.load(os.path.join(GOLDEN, "golden_P8192_60.npz"))
    n, q, psi, omega = meta["n"], meta["q"], meta["psi"], meta["omega"]
    X, tr = oracle.cg_ntt(g["a"], omega, q, trace=True)
    assert np.array_equal(X, g["a_ntt"]) and np.array_equal(tr[:, :16], g["a_trace16"])
    assert np.array_equal(oracle.poly_mult(g["a"], g["b"], q, psi), g["c"])
    assert emu.lib.emu_is_lazy(n, q, psi) == 1
    for canonical in (False, True):
        got = emu.fused(n, q, psi, np.stack([g["a"], g["b"]]), np.stack([g["b"], g["a"]]), canonical)
        assert np.array_equal(got[0], g["c"]) and np.array_equal(got[1], g["c"])
        assert np.array_equal(emu.fused_ntt(n, q, psi, 1, g["a"], canonical), g["a_ntt"])
        assert np.array_equal(emu.fused_ntt(n, q, psi, 2, g["a_ntt"], canonical), g["a"])
    xm = np.zeros(n, dtype=np.uint64); xm[n - 1] = 1
    x1 = np.zeros(n, dtype=np.uint64); x1[1] = 1
    assert np.array_equal(emu.fused(n, q, psi, xm[None], x1[None])[0], g["wrap_c"])


def test_kernel_bodies_under_address_and_ub_sanitizers():
    """GPU sanitizers are not available on the pool: the kernels' per-thread code (the product's own headers, stepped on the CPU by
    tests/emu) runs under AddressSanitizer + UndefinedBehaviorSanitizer instead - every LDS image / table index, every shift count
    of fused_core.h, cg_core.h and plan_tables.h for n = 16 ... 4096, both lane widths, every GROUP x layout x arithmetic of the
    constant-geometry trips, fused products and transforms.  The driver also cross-checks every result (tests/emu/sanitize_driver.cpp)."""
    import subprocess
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
    b = subprocess.run(["make", "-C", d, "sanitize"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert b.returncode == 0, b.stdout[-2000:]
    r = subprocess.run([os.path.join(d, "_build", "sanitize_driver")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    m = __import__("re").search(r"(\d+) checks, 0 failures", r.stdout)
    assert m and int(m.group(1)) > 1500, r.stdout
