"""The CPU oracle (oracle/cg_oracle.c, oracle/bench_port.c) pinned against the reference:
golden vectors produced by importing new_reference/cg_ntt.py (tests/golden/make_golden.py)
and the checksums the reference C++ benchmark prints."""
import numpy as np
import pytest

from conftest import PARAMS, REF_CHECKSUMS

TAGS = ["P4", "P256", "P1024", "P4096", "P4096_60"]


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_poly_mult_matches_golden(oracle, golden, tag):
    g = golden(tag)
    for name in g.cases("poly_mult"):
        a, b, c = g[name + "_a"], g[name + "_b"], g[name + "_c"]
        assert np.array_equal(oracle.poly_mult(a, b, g.q, g.psi), c), name
        assert np.array_equal(oracle.poly_mult(a, b, g.q, g.psi, group=8), c), name + " (8-butterfly)"
        assert np.array_equal(oracle.port_mul(g.n, g.q, g.psi, a, b), c), name + " (benchmark port)"


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_ntt_matches_golden_with_traces(oracle, golden, tag):
    g = golden(tag)
    for name in g.cases("ntt"):
        x, X = g[name + "_x"], g[name + "_X"]
        for group in (1, 8):
            out, tr = oracle.cg_ntt(x, g.omega, g.q, group=group, trace=True)
            assert np.array_equal(out, X), name
            w = min(16, g.n)
            assert np.array_equal(tr[:, :w], g[name + "_trace16"]), name + " per-stage trace"
            assert np.array_equal(oracle.cg_intt(X, g.omega, g.q, group=group), x % np.uint64(g.q)), name + " round trip"
        assert np.array_equal(oracle.port_ntt(g.n, g.q, g.psi, x), X), name + " (iterative CT port)"
        assert np.array_equal(oracle.port_ntt(g.n, g.q, g.psi, X, inverse=True), x % np.uint64(g.q))


@pytest.mark.parametrize("tag", ["P1024", "P4096", "P4096_60"])
def test_port_reproduces_reference_benchmark_checksums(oracle, golden, tag):
    n, q, psi = PARAMS[tag]
    a, b = oracle.make_poly(1, n, q), oracle.make_poly(2, n, q)
    g = golden(tag)
    assert np.array_equal(a, g["lcg12_mul_a"]) and np.array_equal(b, g["lcg12_mul_b"])     # make_poly restated correctly
    fwd = oracle.port_forward_bench(n, q, psi, a)
    assert np.array_equal(fwd, g["lcg1_fwd"])
    c = oracle.port_mul(n, q, psi, a, b)
    assert (oracle.checksum(fwd, q), oracle.checksum(c, q)) == REF_CHECKSUMS[tag]
    assert np.array_equal(c, oracle.poly_mult(a, b, q, psi))


def test_oracle_vs_schoolbook_random(oracle):
    n, q, psi = PARAMS["P256"]
    rng = np.random.default_rng(11)
    for _ in range(3):
        a = rng.integers(0, q, n, dtype=np.uint64); b = rng.integers(0, q, n, dtype=np.uint64)
        assert np.array_equal(oracle.poly_mult(a, b, q, psi), oracle.schoolbook(a, b, q))


def test_oracle_reduces_unreduced_inputs(oracle):
    n, q, psi = PARAMS["P256"]
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2 ** 63, n, dtype=np.uint64); b = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    assert np.array_equal(oracle.poly_mult(a, b, q, psi), oracle.poly_mult(a % np.uint64(q), b % np.uint64(q), q, psi))


def test_literature_kat_n4(oracle, golden):
    g = golden("P4")           # test/refs/fast_ntt_negacyclic_convolution.py:161-214
    assert list(oracle.poly_mult(np.array([1, 2, 3, 4], dtype=np.uint64), np.array([5, 6, 7, 8], dtype=np.uint64), g.q, g.psi)) == [7625, 7645, 2, 60]


def test_oracle_follows_the_reference_for_any_omega(oracle):
    """The CPU restatement of cg_ntt / cg_intt against outputs of the reference for omegas that are not primitive roots
    (tests/golden/make_golden.py: general_omega_cases)."""
    import json
    import os
    from conftest import GOLDEN
    meta = json.load(open(os.path.join(GOLDEN, "golden_general_omega.json")))
    arrs = np.load(os.path.join(GOLDEN, "golden_general_omega.npz"))
    assert len(meta) >= 8
    for m in meta:
        x, X = arrs[m["name"] + "_x"], arrs[m["name"] + "_X"]
        assert np.array_equal(oracle.cg_ntt(x, m["omega"], m["q"]), X), m
        if m.get("has_inverse"):
            assert np.array_equal(oracle.cg_intt(X, m["omega"], m["q"]), arrs[m["name"] + "_inv"]), m


def test_oracle_follows_the_reference_for_any_psi_and_modulus(oracle):
    """nwc_poly_mult validates neither psi_2n nor the modulus (cg_ntt.py:78-92): the CPU restatement against outputs of the
    reference for non-root psi, composite and even moduli (tests/golden/make_golden.py: general_psi_cases)."""
    import json
    import os
    from conftest import GOLDEN
    meta = json.load(open(os.path.join(GOLDEN, "golden_general_psi.json")))
    arrs = np.load(os.path.join(GOLDEN, "golden_general_psi.npz"))
    assert sum(m["kind"] == "poly_mult" for m in meta) >= 14 and sum(m["kind"] == "ntt" for m in meta) >= 4
    for m in meta:
        if m["kind"] == "poly_mult":
            a, b, c = (arrs[m["name"] + s] for s in ("_a", "_b", "_c"))
            assert np.array_equal(oracle.poly_mult(a[None], b[None], m["q"], m["psi"])[0], c), m
            assert np.array_equal(oracle.poly_mult(a[None], b[None], m["q"], m["psi"], group=8)[0], c), m
        else:
            x, X = arrs[m["name"] + "_x"], arrs[m["name"] + "_X"]
            assert np.array_equal(oracle.cg_ntt(x, m["omega"], m["q"]), X), m
            assert np.array_equal(oracle.cg_intt(X, m["omega"], m["q"]), arrs[m["name"] + "_inv"]), m
