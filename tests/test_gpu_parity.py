"""GPU parity tests proper (run with -m gpu on the MI355X box).  Everything goes through the
C ABI (libtinyntt.so via tiny_ntt_amd.engine); results must be bit-exact against the committed
golden vectors (made by the reference's cg_ntt.py), against the oracle on seeded inputs, and —
at BASELINE.json's full sizes — satisfy size-independent properties (reference checksums,
linearity, commutativity, identity, wrap-around)."""
import random

import os

import numpy as np
import pytest

from conftest import GOLDEN, PARAMS, REF_CHECKSUMS, ntt_prime_below

pytestmark = pytest.mark.gpu

TAGS = ["P4", "P256", "P1024", "P4096", "P4096_60"]
ALL_VARIANTS = ["fused", "cg", "cg8", "cg8_padded"]
# the whole lane-grouping x LDS-layout sweep of the constant-geometry kernel (BASELINE config 5): identical bits everywhere
CG_SWEEP = ["cg", "cg_swizzled", "cg2", "cg2_padded", "cg2_swizzled", "cg4", "cg4_padded", "cg4_swizzled", "cg8", "cg8_padded", "cg8_swizzled"]


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from tiny_ntt_amd import engine
    lib = engine.load_library()
    assert engine.LIB_PATH.endswith("tiny_ntt_amd/lib/libtinyntt.so") and lib is not None
    return engine


def plan_for(eng, tag, flags=0):
    return eng.get_plan(*PARAMS[tag], 0, flags)


def variants_of(plan):
    return [v for v in ALL_VARIANTS if v != "fused" or plan.has_fused]


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("tag", TAGS)
def test_poly_mult_matches_reference_golden(eng, golden, tag):
    g, plan = golden(tag), plan_for(eng, tag)
    for name in g.cases("poly_mult"):
        a, b, c = g[name + "_a"], g[name + "_b"], g[name + "_c"]
        for v in variants_of(plan):
            got = plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v)
            assert got.dtype == plan.dtype and np.array_equal(got.astype(np.uint64), c), f"{tag} {name} {v}"


@pytest.mark.parametrize("tag", ["P256", "P1024", "P4096", "P4096_60"])
def test_fused_canonical_policy_matches_golden(eng, golden, tag):
    g = golden(tag)
    plan = plan_for(eng, tag, eng.PLAN_FORCE_CANONICAL)
    assert plan.has_fused and not plan.is_lazy
    assert plan_for(eng, tag).is_lazy
    for name in g.cases("poly_mult"):
        got = plan.poly_mult(g[name + "_a"].astype(plan.dtype), g[name + "_b"].astype(plan.dtype), variant="fused")
        assert np.array_equal(got.astype(np.uint64), g[name + "_c"]), name


@pytest.mark.parametrize("tag", TAGS)
def test_transforms_and_stage_traces_match_reference_golden(eng, golden, tag):
    g, plan = golden(tag), plan_for(eng, tag)
    assert plan.omega == g.omega
    w = min(16, g.n)
    for name in g.cases("ntt"):
        x, X = g[name + "_x"], g[name + "_X"]
        for v in ("auto", "fused") if plan.has_fused else ():
            assert np.array_equal(plan.ntt_forward(x.astype(plan.dtype), variant=v).astype(np.uint64), X), f"{name} {v}"
            assert np.array_equal(plan.ntt_inverse(X.astype(plan.dtype), variant=v).astype(np.uint64), x % np.uint64(g.q)), f"{name} {v} inverse"
        for v in CG_SWEEP:
            assert np.array_equal(plan.ntt_forward(x.astype(plan.dtype), variant=v).astype(np.uint64), X), f"{name} {v}"
            assert np.array_equal(plan.ntt_inverse(X.astype(plan.dtype), variant=v).astype(np.uint64), x % np.uint64(g.q)), f"{name} {v} inverse"
            out, trace = plan.ntt_forward_trace(x.astype(plan.dtype), variant=v)
            assert np.array_equal(out.astype(np.uint64), X)
            assert np.array_equal(trace[:, :w].astype(np.uint64), g[name + "_trace16"]), f"{name} {v} per-stage trace (cg_ntt.py:60-62)"
    if tag != "P4":
        for v in ("auto", "cg", "cg8"):
            fwd = plan.twisted_ntt_forward(g["lcg12_mul_a"].astype(plan.dtype), variant=v)
            assert np.array_equal(fwd.astype(np.uint64), g["lcg1_fwd"]), v


# ---------------------------------------------------------------- seeded random vs the oracle
@pytest.mark.parametrize("tag", TAGS)
def test_random_batches_vs_oracle_incl_unreduced_inputs(eng, oracle, tag):
    n, q, psi = PARAMS[tag]
    plan = plan_for(eng, tag)
    rng = np.random.default_rng(1234)
    B = 64
    word = 2 ** (8 * plan.elem_bytes) - 1
    a = rng.integers(0, q, (B, n), dtype=np.uint64); b = rng.integers(0, q, (B, n), dtype=np.uint64)
    a[0], b[0] = q - 1, q - 1
    a[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True); b[1] = word     # any word is taken mod q
    a[2], b[2] = word, word
    a[3] = 0
    ref = oracle.poly_mult(a, b, q, psi)
    for v in variants_of(plan) + CG_SWEEP:
        got = plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v).astype(np.uint64)
        assert np.array_equal(got, ref), f"{tag} {v}: {np.count_nonzero(got != ref)} coefficients differ"
    for v in (["fused"] if plan.has_fused else []) + ["cg"]:
        A = plan.ntt_forward(a.astype(plan.dtype), variant=v).astype(np.uint64)
        for r in (0, 1, 2, 5):
            assert np.array_equal(A[r], oracle.cg_ntt(a[r], plan.omega, q)), (tag, v, r)
        assert np.array_equal(plan.ntt_inverse(A.astype(plan.dtype), variant=v).astype(np.uint64), a % np.uint64(q)), (tag, v)


def test_ragged_and_empty_batches(eng, oracle):
    n, q, psi = PARAMS["P1024"]
    plan = plan_for(eng, "P1024")
    rng = np.random.default_rng(5)
    for B in (1, 2, 3, 63, 65, 257):
        a = rng.integers(0, q, (B, n), dtype=np.uint32); b = rng.integers(0, q, (B, n), dtype=np.uint32)
        assert np.array_equal(plan.poly_mult(a, b).astype(np.uint64), oracle.poly_mult(a.astype(np.uint64), b.astype(np.uint64), q, psi))
    empty = np.empty((0, n), dtype=np.uint32)
    assert plan.poly_mult(empty, empty).shape == (0, n)
    one = plan.poly_mult(a[0], b[0])
    assert one.shape == (n,)
    with pytest.raises(ValueError, match="Expected 1024 coefficients"):
        plan.poly_mult(a[:, :100], b[:, :100])


def test_host_entry_points_pipeline_chunks(eng, oracle):
    """*_host entry points: the H2D -> kernel -> D2H pipeline with more chunks than staging slots, a ragged last
    chunk, pageable and pinned host buffers; results identical to the single-chunk path and to the oracle."""
    import torch
    n, q, psi = PARAMS["P4096_60"]
    plan = eng.Plan(n, q, psi)                       # private plan: the chunk size is per-plan state
    rng = np.random.default_rng(77)
    B = 23
    a = rng.integers(0, q, (B, n), dtype=np.uint64); b = rng.integers(0, q, (B, n), dtype=np.uint64)
    ref = oracle.poly_mult(a, b, q, psi)
    whole = plan.poly_mult(a, b)
    assert np.array_equal(whole, ref)
    pa, pb = torch.from_numpy(a).pin_memory().numpy(), torch.from_numpy(b).pin_memory().numpy()
    for rows in (1, 2, 5, 7, 23, 64):                # 23, 12, 5, 4, 1, 1 chunks
        plan.set_host_chunk_rows(rows)
        assert np.array_equal(plan.poly_mult(a, b), ref), rows
        assert np.array_equal(plan.poly_mult(pa, pb), ref), rows
        A = plan.ntt_forward(a, variant="fused")
        assert np.array_equal(A[B - 1], oracle.cg_ntt(a[B - 1], plan.omega, q)), rows
        assert np.array_equal(plan.ntt_inverse(A, variant="fused"), a), rows
    plan.set_host_chunk_rows(0)
    assert np.array_equal(plan.poly_mult(a, b, variant="cg"), ref)
    plan.close()


def test_generic_moduli_run_canonical(eng, oracle):
    n = 256
    for q in (754974721, 2305843009196916737, 7681 * 0 + 12289):
        if (q - 1) % (2 * n):
            continue
        psi = next(pow(x, (q - 1) // (2 * n), q) for x in range(2, 300) if pow(pow(x, (q - 1) // (2 * n), q), n, q) == q - 1)
        plan = eng.get_plan(n, q, psi)
        rng = np.random.default_rng(q % 1000)
        a = rng.integers(0, q, (5, n), dtype=np.uint64); b = rng.integers(0, q, (5, n), dtype=np.uint64)
        a[0], b[0] = q - 1, q - 1
        ref = oracle.poly_mult(a, b, q, psi)
        for v in variants_of(plan):
            assert np.array_equal(plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v).astype(np.uint64), ref), (q, v)


@pytest.mark.parametrize("n", [256, 4096])
def test_modulus_sweep_every_word_size_both_policies(eng, oracle, n):
    """Moduli from 20 to 62 bits: q = 2^k - c (lazy policy where the bounds allow, checked per plan) and primes far from
    a power of two (canonical policy), 32- and 64-bit lanes, fused and constant-geometry kernels, against the oracle."""
    from tiny_ntt_amd import numtheory
    seen_lazy, seen_canon = set(), set()
    for k in (20, 26, 31, 32, 33, 36, 41, 47, 52, 57, 60, 61, 62):
        for limit in (2 ** k, int(0.71 * 2 ** k)):
            q = ntt_prime_below(limit, n)
            psi = numtheory.primitive_2n_root(n, q)
            plan = eng.Plan(n, q, psi)
            (seen_lazy if plan.is_lazy else seen_canon).add(q.bit_length())
            rng = np.random.default_rng(k)
            word = 2 ** (8 * plan.elem_bytes) - 1
            a = rng.integers(0, q, (6, n), dtype=np.uint64); b = rng.integers(0, q, (6, n), dtype=np.uint64)
            a[0], b[0] = q - 1, q - 1
            a[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True); b[1] = word
            ref = oracle.poly_mult(a, b, q, psi)
            for v in ("fused", "cg"):
                got = plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v).astype(np.uint64)
                assert np.array_equal(got, ref), (n, q, v, plan.is_lazy)
            A = plan.ntt_forward(a.astype(plan.dtype), variant="fused").astype(np.uint64)
            assert np.array_equal(A[1], oracle.cg_ntt(a[1], plan.omega, q)), (n, q)
            assert np.array_equal(plan.ntt_inverse(A.astype(plan.dtype), variant="fused").astype(np.uint64), a % np.uint64(q)), (n, q)
            plan.close()
    # (64-bit lanes are lazy only when 2^33 c << 2^k, h_split_sched_ok: at n = 4096 no 41- or 47-bit NTT prime is that close to 2^k)
    assert {26, 52, 57, 60} <= seen_lazy and {20, 31, 32, 33, 36, 41, 61, 62} <= seen_canon, (seen_lazy, seen_canon)


def test_sizes_without_a_fused_kernel_use_cg(eng, oracle):
    for n in (8, 16, 64, 128):
        q = 8380417
        if (q - 1) % (2 * n):
            continue
        psi = next(pow(x, (q - 1) // (2 * n), q) for x in range(2, 300) if pow(pow(x, (q - 1) // (2 * n), q), n, q) == q - 1)
        plan = eng.get_plan(n, q, psi)
        rng = np.random.default_rng(n)
        a = rng.integers(0, q, (9, n), dtype=np.uint64); b = rng.integers(0, q, (9, n), dtype=np.uint64)
        ref = oracle.poly_mult(a, b, q, psi)
        for v in ["auto"] + variants_of(plan):
            assert np.array_equal(plan.poly_mult(a.astype(np.uint32), b.astype(np.uint32), variant=v).astype(np.uint64), ref), (n, v)


# ---------------------------------------------------------------- the reference's own tests, through the mirror modules
def negacyclic_convolution(a, b, q):
    n = len(a)
    res = [0] * n
    for i, ai in enumerate(a):
        if not ai:
            continue
        for j, bj in enumerate(b):
            k, t = i + j, ai * bj % q
            if k >= n:
                k, t = k - n, (-t) % q
            res[k] = (res[k] + t) % q
    return res


@pytest.fixture()
def mirror(eng):
    import tiny_ntt_amd.cg_ntt as cg
    import tiny_ntt_amd.cg_ntt_8butterfly as cg8
    cg.N, cg.Q = 256, 8380417
    yield cg, cg8
    cg.N, cg.Q = 256, 8380417


PSI_2N = 1239911


def test_mirror_identity_like_reference(mirror):            # new_reference/test_cg_ntt.py:44-52, test_cg_ntt_8butterfly.py:49-57
    cg, cg8 = mirror
    omega = pow(PSI_2N, 2, cg.Q)
    random.seed(0)
    a = [random.randrange(cg.Q) for _ in range(cg.N)]
    lines = []
    t = cg.cg_ntt(a, omega, cg.Q, verbose=True, log_fn=lines.append)
    assert cg.cg_intt(t, omega, cg.Q) == a
    assert lines[0] == "CG NTT start" and sum(l.startswith("  stage_out(first 16)=") for l in lines) == 8
    random.seed(2)
    a = [random.randrange(cg.Q) for _ in range(cg.N)]
    assert cg8.cg_intt_8butterfly(cg8.cg_ntt_8butterfly(a, omega, cg.Q), omega, cg.Q) == a
    random.seed(3)                                          # test_cg_ntt_8butterfly.py:60-68
    a = [random.randrange(cg.Q) for _ in range(cg.N)]
    assert cg8.cg_ntt_8butterfly(a, omega, cg.Q) == cg.cg_ntt(a, omega, cg.Q)


def test_mirror_verbose_log_equals_reference_log(mirror, golden):
    cg, cg8 = mirror
    g = golden("P256")
    x = [int(v) for v in g["seed0_ntt_x"]]
    for mod, fn in ((cg, cg.cg_ntt), (cg8, cg8.cg_ntt_8butterfly)):
        lines = []
        fn(x, g.omega, g.q, True, lines.append)
        outs = [eval(l.split("=", 1)[1]) for l in lines if l.startswith("  stage_out(first 16)=")]
        assert np.array_equal(np.array(outs, dtype=np.uint64), g["seed0_ntt_trace16"])
        assert eval([l for l in lines if l.startswith("  bitrev(first 16)=")][0].split("=", 1)[1]) == [int(v) for v in g["seed0_ntt_bitrev16"]]


def test_mirror_kats_and_random_like_reference(mirror):     # test_cg_ntt.py:55-103, test_cg_ntt_8butterfly.py:71-118
    cg, cg8 = mirror
    def sparse(v):
        return v + [0] * (cg.N - len(v))
    for a, b in ((sparse([1, 2, 3]), sparse([4, 5, 6])), (sparse([1, 2, 3]), sparse([5, 1]))):
        expect = negacyclic_convolution(a, b, cg.Q)
        assert cg.nwc_poly_mult(a, b, PSI_2N) == expect
        assert cg8.nwc_poly_mult_8butterfly(a, b, PSI_2N) == expect
    assert cg.nwc_poly_mult(sparse([1, 2, 3]), sparse([4, 5, 6]), PSI_2N)[:6] == [4, 13, 28, 27, 18, 0]
    assert cg.nwc_poly_mult(sparse([1, 5, 1]), sparse([5, 1]), PSI_2N)[:5] == [5, 26, 10, 1, 0]     # test_ntt_inverse.py:273-275
    random.seed(1)
    a = [random.randrange(cg.Q) for _ in range(cg.N)]
    b = [random.randrange(cg.Q) for _ in range(cg.N)]
    assert cg.nwc_poly_mult(a, b, PSI_2N) == negacyclic_convolution(a, b, cg.Q)
    random.seed(4)
    a = [random.randrange(cg.Q) for _ in range(cg.N)]
    b = [random.randrange(cg.Q) for _ in range(cg.N)]
    assert cg8.nwc_poly_mult_8butterfly(a, b, PSI_2N) == cg.nwc_poly_mult(a, b, PSI_2N)
    assert cg.nwc_poly_mult([-1] + [0] * 255, sparse([2]), PSI_2N)[0] == cg.Q - 2        # Python ints of any sign, reduced with %


def test_mirror_60bit_parameter_override(mirror, golden):
    cg, cg8 = mirror
    g = golden("P4096_60")
    cg.N, cg.Q = g.n, g.q
    a, b = [int(v) for v in g["seed1_mul_a"]], [int(v) for v in g["seed1_mul_b"]]
    c = cg.nwc_poly_mult(a, b, g.psi)
    assert c[:3] == [736315498995752632, 3636685963993676, 530706514351249966]
    assert c == [int(v) for v in g["seed1_mul_c"]]
    assert cg8.nwc_poly_mult_8butterfly(a, b, g.psi) == c


# ---------------------------------------------------------------- BASELINE configs at full size: properties
def _full_size_properties(eng, oracle, tag, batch, check_rows):
    import torch
    n, q, psi = PARAMS[tag]
    plan = plan_for(eng, tag)
    a = plan.fill_lcg(batch, 1, 2)            # row r = make_poly(2r+1)
    b = plan.fill_lcg(batch, 2, 2)            # row r = make_poly(2r+2)
    c = plan.poly_mult(a, b)
    ha, hb, hc = plan.to_host(a[:check_rows]), plan.to_host(b[:check_rows]), plan.to_host(c[:check_rows])
    # device generator == the reference's make_poly; row 0 reproduces the reference benchmark's printed checksum
    assert np.array_equal(ha[0].astype(np.uint64), oracle.make_poly(1, n, q)) and np.array_equal(hb[0].astype(np.uint64), oracle.make_poly(2, n, q))
    assert np.array_equal(ha[5].astype(np.uint64), oracle.make_poly(11, n, q))
    sums = plan.checksum_rows(c)
    assert int(sums[0]) == REF_CHECKSUMS[tag][1]
    fwd = plan.twisted_ntt_forward(a[:1])
    assert int(plan.checksum_rows(fwd)[0]) == REF_CHECKSUMS[tag][0]
    # sample rows incl. the last ones, full compare vs the oracle
    rows = list(range(check_rows))
    ref = oracle.poly_mult(ha.astype(np.uint64), hb.astype(np.uint64), q, psi)
    assert np.array_equal(hc.astype(np.uint64), ref)
    tail = slice(batch - 4, batch)
    assert np.array_equal(plan.to_host(c[tail]).astype(np.uint64),
                          oracle.poly_mult(plan.to_host(a[tail]).astype(np.uint64), plan.to_host(b[tail]).astype(np.uint64), q, psi))
    # checksum kernel == reference checksum function on the sampled rows
    for r in (0, 1, check_rows - 1):
        assert int(sums[r]) == oracle.checksum(hc[r].astype(np.uint64), q)
    # commutativity over the whole batch (checksum of checksums)
    c2 = plan.poly_mult(b, a)
    assert torch.equal(c, c2)
    # identity: a * 1 = a mod q ; wrap-around: (a * x^(n-1)) * x = -a
    ident = torch.zeros_like(a[:256]); ident[:, 0] = 1
    assert torch.equal(plan.poly_mult(a[:256], ident), a[:256])
    xm = torch.zeros_like(a[:256]); xm[:, n - 1] = 1
    x1 = torch.zeros_like(a[:256]); x1[:, 1] = 1
    neg = plan.to_host(plan.poly_mult(plan.poly_mult(a[:256], xm), x1)).astype(np.uint64)
    assert np.array_equal(neg, (np.uint64(q) - plan.to_host(a[:256]).astype(np.uint64)) % np.uint64(q))
    # linearity: (a + a') * b = a*b + a'*b  (mod q), elementwise modular add on the host for a sample
    ap = plan.fill_lcg(256, 1001, 3)
    s = (plan.to_host(a[:256]).astype(np.uint64) + plan.to_host(ap).astype(np.uint64)) % np.uint64(q)
    lhs = plan.poly_mult(s.astype(plan.dtype), plan.to_host(b[:256]))
    rhs = (plan.to_host(c[:256]).astype(np.uint64) + plan.to_host(plan.poly_mult(ap, b[:256])).astype(np.uint64)) % np.uint64(q)
    assert np.array_equal(lhs.astype(np.uint64), rhs)
    # EVERY row of the batch: the fused (persistent, register-tiled) kernel equals the constant-geometry kernel,
    # a different dataflow with canonical arithmetic; the grouped variants on a slab
    assert torch.equal(plan.poly_mult(a, b, variant="cg"), c)
    for v in CG_SWEEP[1:]:
        assert torch.equal(plan.poly_mult(a[:512], b[:512], variant=v), c[:512]), v
    # outputs canonical
    assert int(plan.to_host(c).max()) < q


def test_config2_n1024_24bit_batch4096(eng, oracle):
    _full_size_properties(eng, oracle, "P1024", 4096, 256)


def test_config3_n4096_60bit_batch65536(eng, oracle):
    _full_size_properties(eng, oracle, "P4096_60", 65536, 256)


def test_n4096_24bit_batch8192(eng, oracle):
    _full_size_properties(eng, oracle, "P4096", 8192, 64)


def test_device_buffers_streams_and_aliasing(eng):
    import torch
    plan = plan_for(eng, "P4096_60")
    a = plan.fill_lcg(128, 1, 2); b = plan.fill_lcg(128, 2, 2)
    plan.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c1 = plan.poly_mult(a, b, stream=s)
    s.synchronize()
    c2 = plan.poly_mult(a, b); plan.synchronize()
    assert torch.equal(c1, c2)
    with pytest.raises(eng.TinyNttError, match="alias"):
        plan.poly_mult(a, b, out=a)
    with pytest.raises(eng.TinyNttError):
        plan.poly_mult(a.cpu(), b.cpu())
    with pytest.raises(eng.TinyNttError, match="per-stage traces need a CG variant"):
        plan.ntt_forward_trace(plan.to_host(a[:1]), variant="fused")
    assert torch.equal(plan.ntt_forward(a, variant="fused"), plan.ntt_forward(a, variant="cg"))
    assert torch.equal(plan.ntt_inverse(plan.ntt_forward(a)), a)
    assert plan.kernel_name("fused") == "polymul_fused_kernel" and plan.kernel_name("cg8") == "cg_kernel"


def test_concurrent_launches_of_one_plan_on_several_streams(eng):
    """The dynamically scheduled persistent kernels take their rows from device counters: launches of ONE plan that are
    in flight together (different streams) must use different counter pairs.  Long launches (dynamic) and short ones
    (fixed stride) mixed, products and standalone transforms, results compared with the same calls issued one by one."""
    import torch
    plan = plan_for(eng, "P4096_60")
    sizes = [9000, 4097, 12000, 300, 7000, 2048]
    ins = [(plan.fill_lcg(B, 10 * i + 1, 2), plan.fill_lcg(B, 10 * i + 2, 2)) for i, B in enumerate(sizes)]
    plan.synchronize(); torch.cuda.synchronize()
    ref = [plan.poly_mult(a, b) for a, b in ins]
    ref_t = [plan.ntt_forward(a, variant="fused") for a, _ in ins]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in sizes]
    for rep in range(3):
        outs, outs_t = [], []
        for (a, b), st in zip(ins, streams):
            outs.append(plan.poly_mult(a, b, stream=st))
            outs_t.append(plan.ntt_forward(a, variant="fused", stream=st))
        torch.cuda.synchronize()
        for i in range(len(sizes)):
            assert torch.equal(outs[i], ref[i]) and torch.equal(outs_t[i], ref_t[i]), (rep, i)
    # and many back-to-back launches on one stream (the counter pairs re-arm themselves and wrap around the ring)
    a, b = ins[5]
    for _ in range(1100):                      # more launches than the ring has slots (tn_plan::SCHED_SLOTS = 1024)
        c = plan.poly_mult(a, b)
    torch.cuda.synchronize()
    assert torch.equal(c, ref[5])


def test_config4_total_batch_1M_rows_on_one_gpu_max_size(eng, oracle):
    """BASELINE configs[3]'s whole batch (2^20 pairs, 96 GiB for a, b, c) on a single 288 GB MI355X: the largest
    size in BASELINE.json; exercises > 4 GiB offsets and the persistent grid's tail.  Size-independent checks:
    reference checksum on row 0, sampled rows vs the oracle, slab re-computation, commutativity on a slab."""
    import torch
    n, q, psi = PARAMS["P4096_60"]
    rows = 1 << 20
    free, _total = torch.cuda.mem_get_info()
    if free < 100 * 2 ** 30:
        pytest.skip(f"needs ~96 GiB of free HBM, {free / 2 ** 30:.0f} GiB available")
    plan = plan_for(eng, "P4096_60")
    a = plan.fill_lcg(rows, 1, 2); b = plan.fill_lcg(rows, 2, 2)
    c = plan.poly_mult(a, b)
    sums = plan.checksum_rows(c[:4])
    assert int(sums[0]) == REF_CHECKSUMS["P4096_60"][1]
    idx = [0, 1, 65535, 65536, 524287, 524288, rows - 2, rows - 1]
    ha, hb, hc = plan.to_host(a[idx]), plan.to_host(b[idx]), plan.to_host(c[idx])
    assert np.array_equal(ha[0], oracle.make_poly(1, n, q)) and np.array_equal(ha[-1], oracle.make_poly(2 * (rows - 1) + 1, n, q))
    assert np.array_equal(hc, oracle.poly_mult(ha, hb, q, psi))
    for start in (0, 3 * 65536 + 17, rows - 300):                    # slabs recomputed on their own == the big launch
        sl = slice(start, start + 300)
        assert torch.equal(plan.poly_mult(a[sl].contiguous(), b[sl].contiguous()), c[sl])
    assert torch.equal(plan.poly_mult(b[-4096:].contiguous(), a[-4096:].contiguous()), c[-4096:])
    del a, b, c
    torch.cuda.empty_cache()


def test_two_host_threads_with_their_own_plans(eng, oracle):
    """Threading contract of include/tinyntt.h: distinct plans may be used from distinct host threads."""
    import threading
    results, errors = {}, []

    def work(tag, seed):
        try:
            n, q, psi = PARAMS[tag]
            plan = eng.Plan(n, q, psi)                 # private plan, private stream
            rng = np.random.default_rng(seed)
            a = rng.integers(0, q, (32, n), dtype=np.uint64); b = rng.integers(0, q, (32, n), dtype=np.uint64)
            for _ in range(5):
                got = plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype)).astype(np.uint64)
            results[tag] = (got, a, b)
            plan.close()
        except Exception as e:                         # pragma: no cover
            errors.append((tag, e))

    ts = [threading.Thread(target=work, args=(t, i)) for i, t in enumerate(("P4096_60", "P1024", "P4096", "P256"))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors, errors
    for tag, (got, a, b) in results.items():
        n, q, psi = PARAMS[tag]
        assert np.array_equal(got, oracle.poly_mult(a, b, q, psi)), tag


@pytest.mark.parametrize("n", [512, 2048])
@pytest.mark.parametrize("q", [8380417, 1152921504606830593])
def test_fused_kernels_for_n512_n2048(eng, oracle, n, q):
    from tiny_ntt_amd import numtheory
    psi = numtheory.primitive_2n_root(n, q)
    plan = eng.get_plan(n, q, psi)
    assert plan.has_fused and plan.is_lazy
    rng = np.random.default_rng(n + 1)
    a = rng.integers(0, q, (300, n), dtype=np.uint64); b = rng.integers(0, q, (300, n), dtype=np.uint64)
    a[0], b[0] = q - 1, q - 1
    ref = oracle.poly_mult(a, b, q, psi)
    for v in variants_of(plan):
        assert np.array_equal(plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v).astype(np.uint64), ref), (n, q, v)


def test_n8192_60bit_matches_reference_golden_and_oracle(eng, oracle):
    """n = 8192 at the reference's 60-bit modulus: the largest negacyclic size that modulus admits (2n = 2^14 | q - 1).  Golden
    vectors from the reference with N overridden (tests/golden/make_golden.py n8192_cases); every variant, both policies."""
    import json
    meta = json.load(open(os.path.join(GOLDEN, "golden_P8192_60.json")))
    g = np.load(os.path.join(GOLDEN, "golden_P8192_60.npz"))
    n, q, psi, omega = meta["n"], meta["q"], meta["psi"], meta["omega"]
    plan = eng.get_plan(n, q, psi)
    cplan = eng.get_plan(n, q, psi, 0, eng.PLAN_FORCE_CANONICAL)
    assert plan.has_fused and plan.is_lazy and plan.omega == omega and not cplan.is_lazy
    a, b = g["a"][None].copy(), g["b"][None].copy()
    for p in (plan, cplan):
        for v in ("fused", "cg", "cg8", "cg4_swizzled", "cg2_padded"):
            assert np.array_equal(p.poly_mult(a, b, variant=v)[0], g["c"]), v
        for v in ("fused", "cg", "cg8"):
            assert np.array_equal(p.ntt_forward(g["a"], variant=v), g["a_ntt"]), v
            assert np.array_equal(p.ntt_inverse(g["a_ntt"], variant=v), g["a"]), v
        out, trace = p.ntt_forward_trace(g["a"], variant="cg")
        assert np.array_equal(out, g["a_ntt"]) and np.array_equal(trace[:, :16], g["a_trace16"])
    xm = np.zeros(n, dtype=np.uint64); xm[n - 1] = 1
    x1 = np.zeros(n, dtype=np.uint64); x1[1] = 1
    assert np.array_equal(plan.poly_mult(xm, x1), g["wrap_c"])
    # a random batch through the persistent kernel (several rows per workgroup), incl. unreduced words, vs the CPU oracle
    rng = np.random.default_rng(8192)
    B = 300
    A = rng.integers(0, q, (B, n), dtype=np.uint64); Bm = rng.integers(0, q, (B, n), dtype=np.uint64)
    A[0], Bm[0] = q - 1, q - 1
    A[1] = rng.integers(0, 2 ** 64 - 1, n, dtype=np.uint64, endpoint=True); Bm[1] = 2 ** 64 - 1
    ref = oracle.poly_mult(A[:8], Bm[:8], q, psi)
    got = plan.poly_mult(A, Bm)
    assert np.array_equal(got[:8], ref)
    assert np.array_equal(got, plan.poly_mult(A, Bm, variant="cg"))
    assert np.array_equal(cplan.poly_mult(A[:40], Bm[:40]), got[:40])
    X = plan.ntt_forward(A[:40])
    assert np.array_equal(X, plan.ntt_forward(A[:40], variant="cg")) and np.array_equal(plan.ntt_inverse(X), A[:40] % np.uint64(q))
    assert np.array_equal(plan.cyclic_poly_mult(A[:40], Bm[:40]), plan.cyclic_poly_mult(A[:40], Bm[:40], variant="cg"))


@pytest.mark.gpu
def test_wide_parity_fuzz_all_sizes_and_word_lengths():
    """tests/dev/gpu_fuzz.py: n in {8 ... 8192} x NTT primes of every bit length 14..62 x fused + three random constant-geometry
    variants, special rows (q-1, unreduced words, zeros, a monomial) and random ones, against the CPU oracle (~2,600 checks, seconds)."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "gpu_fuzz.py"), "11", "200"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and ", 0 mismatches" in r.stdout, r.stdout[-2000:]


def test_signed_host_values_and_device_bit_patterns(eng, oracle):
    """Host arrays are VALUES (negative numbers taken mod q like the reference's %, cg_ntt.py:82-83); device tensors are BIT
    PATTERNS (torch.int64 storage read as unsigned words).  The same 64 bits therefore mean different residues on the two
    paths, and each equals what the reference computes for that reading (engine.Plan docstring)."""
    import torch
    plan = plan_for(eng, "P4096_60")
    n, q, psi = PARAMS["P4096_60"]
    rng = np.random.default_rng(11)
    a_signed = rng.integers(-2 ** 63, 2 ** 63 - 1, (2, n), dtype=np.int64)
    b = rng.integers(0, q, (2, n), dtype=np.uint64)
    as_values = np.array([[int(v) % q for v in row] for row in a_signed], dtype=np.uint64)            # Python's %, what the reference does with ints
    as_patterns = a_signed.view(np.uint64)
    ref_values = oracle.poly_mult(as_values, b, q, psi)
    ref_patterns = oracle.poly_mult(as_patterns % np.uint64(q), b, q, psi)
    assert np.array_equal(plan.poly_mult(a_signed, b), ref_values)                                     # host, signed: values
    assert np.array_equal(plan.poly_mult(as_patterns, b), ref_patterns)                                # host, unsigned view: patterns
    dev = plan.poly_mult(torch.from_numpy(a_signed).to("cuda"), plan.to_device(b))                     # device: patterns
    assert np.array_equal(plan.to_host(dev), ref_patterns)
    assert not np.array_equal(ref_values, ref_patterns)


def test_constant_geometry_kernels_take_rows_from_the_counter_like_the_fused_ones(eng):
    """The persistent constant-geometry kernels hand out rows through the same device counters (cg_kernel_impl.h: long launches;
    fixed stride for short ones and under stream capture).  Every row of long and short launches, products and standalone
    transforms, several streams at once and more back-to-back launches than the ring has slots must equal the fused kernel's rows
    (a skipped or doubled row shows as a mismatch: the outputs start zeroed)."""
    import torch
    for tag, sizes in (("P4096_60", [9000, 4097, 300]), ("P256", [70000, 40001, 515])):
        plan = plan_for(eng, tag)
        ins = [(plan.fill_lcg(B, 20 * i + 1, 2), plan.fill_lcg(B, 20 * i + 2, 2)) for i, B in enumerate(sizes)]
        ref = [plan.poly_mult(a, b, variant="fused") for a, b in ins]
        ref_t = [plan.ntt_forward(a, variant="fused") for a, _ in ins]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in sizes]
        for variant in ("cg8_padded", "cg4_swizzled", "cg"):
            for rep in range(2):
                outs = [torch.zeros_like(a) for a, _ in ins]; outs_t = [torch.zeros_like(a) for a, _ in ins]
                torch.cuda.synchronize()
                for (a, b), st, o, ot in zip(ins, streams, outs, outs_t):
                    plan.poly_mult(a, b, out=o, variant=variant, stream=st)
                    plan.ntt_forward(a, variant=variant, out=ot, stream=st)
                torch.cuda.synchronize()
                for i in range(len(sizes)):
                    assert torch.equal(outs[i], ref[i]), (tag, variant, rep, i)
                    assert torch.equal(outs_t[i], ref_t[i]), (tag, variant, rep, i)
        a, b = ins[0]
        c = torch.zeros_like(a)
        for _ in range(1100 if tag == "P256" else 40):     # the ring (1,024 pairs) wraps; every pair re-arms itself
            plan.poly_mult(a, b, out=c, variant="cg8_padded")
        torch.cuda.synchronize()
        assert torch.equal(c, ref[0]), tag
    # rows shorter than 32 KiB are handed out in chunks of several rows per atomic: n = 2048 / 60-bit -> 2 rows, n = 1024 / 24-bit -> 8;
    # an odd batch leaves the last chunk partly past the end
    from tiny_ntt_amd import numtheory
    for n, q, rows in ((2048, 1152921504606830593, 20001), (1024, 8380417, 200003)):
        plan = eng.get_plan(n, q, numtheory.primitive_2n_root(n, q))
        a, b = plan.fill_lcg(rows, 3, 2), plan.fill_lcg(rows, 4, 2)
        ref = plan.poly_mult(a, b, variant="fused")
        for variant in ("cg8_padded", "cg4_padded", "cg"):
            for rep in range(2):
                c = torch.zeros_like(a)
                plan.poly_mult(a, b, out=c, variant=variant)
                torch.cuda.synchronize()
                assert torch.equal(c, ref), (n, variant, rep)
        del a, b, c, ref
    # captured launches run the fixed stride (no counter pair is baked into a graph node)
    plan = plan_for(eng, "P4096_60")
    a, b = plan.fill_lcg(6144, 1, 2), plan.fill_lcg(6144, 2, 2)
    ref = plan.poly_mult(a, b)
    s1 = torch.cuda.Stream(); c = torch.zeros_like(a)
    with torch.cuda.stream(s1):
        plan.poly_mult(a, b, out=c, variant="cg8_padded", stream=s1)
    s1.synchronize()
    g = torch.cuda.CUDAGraph(); c.zero_(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        plan.poly_mult(a, b, out=c, variant="cg8_padded", stream=s1)
    for rep in range(3):
        c.zero_(); torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            g.replay()
        plan.poly_mult(a, b, variant="cg8_padded")          # a live, dynamically scheduled launch beside the replay
        torch.cuda.synchronize()
        assert torch.equal(c, ref), rep


def test_graph_capture_uses_the_fixed_stride_and_replays_beside_live_launches(eng):
    """A launch captured into a hipGraph must not take a slot of the dynamic row scheduler (the slot pointer would be baked into
    the kernel node while the ring keeps advancing, and a later replay could share a counter pair with a live launch: rows
    skipped).  Captured launches run the fixed stride (plan.h: sched_acquire).  Capture a few launches, run more live launches than
    the ring has slots (1,024), then replay the graph while a second stream keeps launching: every result must be right."""
    import torch
    plan = plan_for(eng, "P4096_60")
    rows = 6144                                        # long enough for dynamic row scheduling on the live launches
    a = plan.fill_lcg(rows, 1, 2); b = plan.fill_lcg(rows, 2, 2)
    a2 = plan.fill_lcg(rows, 7, 2); b2 = plan.fill_lcg(rows, 8, 2)
    ref = plan.poly_mult(a, b); ref2 = plan.poly_mult(a2, b2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    c = torch.zeros_like(a); c2 = torch.zeros_like(a)
    with torch.cuda.stream(s1):
        plan.poly_mult(a, b, out=c, stream=s1)         # warm-up outside the capture (lazy kernel attributes)
    s1.synchronize()
    g = torch.cuda.CUDAGraph()
    c.zero_(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        for _ in range(3):
            plan.poly_mult(a, b, out=c, stream=s1)
    for _ in range(1100):                              # the ring wraps: slots used before the capture are handed out again
        plan.poly_mult(a2, b2, out=c2, stream=s2)
    for rep in range(6):
        c.zero_(); torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            g.replay()
        for _ in range(8):                             # live, dynamically scheduled launches beside the replay
            plan.poly_mult(a2, b2, out=c2, stream=s2)
        torch.cuda.synchronize()
        assert torch.equal(c, ref) and torch.equal(c2, ref2), rep
