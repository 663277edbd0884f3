"""GPU tests for the SURVEY.md §8(f) rows: cyclic (RTL-style) product, pointwise product, on-device O(n^2)
checker, plan-table export against the reference hex format, RoCC-protocol session facade."""
import json
import os

import numpy as np
import pytest

from conftest import is_prime, GOLDEN, PARAMS, REF_CHECKSUMS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    from tiny_ntt_amd import engine
    return engine


def cyclic_oracle(oracle, a, b, q, omega):
    A, B = oracle.cg_ntt(a, omega, q), oracle.cg_ntt(b, omega, q)
    C = np.array([int(x) * int(y) % q for x, y in zip(A, B)], dtype=np.uint64)      # python_poly_mult, test_ntt_poly_mult.py:38-43
    return oracle.cg_intt(C, omega, q)


@pytest.mark.parametrize("tag", ["P4", "P256", "P1024", "P4096", "P4096_60"])
def test_cyclic_product_matches_rtl_reference_model(eng, oracle, tag):
    n, q, psi = PARAMS[tag]
    plan = eng.get_plan(n, q, psi)
    rng = np.random.default_rng(99)
    a = rng.integers(0, q, (3, n), dtype=np.uint64); b = rng.integers(0, q, (3, n), dtype=np.uint64)
    a[0] = 0; a[0, :3] = [1, 2, 3]; b[0] = 0; b[0, :2] = [5, 1]                      # chipyard/ntt-test.c KAT
    for v in ("cg", "cg8", "cg8_padded", "cg_swizzled", "cg4_swizzled", "cg2_padded") + (("fused", "auto") if plan.has_fused else ()):
        got = plan.cyclic_poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant=v).astype(np.uint64)
        for r in range(3):
            assert np.array_equal(got[r], cyclic_oracle(oracle, a[r], b[r], q, plan.omega)), (tag, v, r)
    if n >= 8:
        assert list(got[0][:5]) == [5, 11, 17, 3, 0]                                  # no wrap-around: same as the negacyclic KAT
    # cyclic wrap-around differs from negacyclic: x^(n-1) * x = +1
    xm = np.zeros(n, dtype=plan.dtype); xm[n - 1] = 1
    x1 = np.zeros(n, dtype=plan.dtype); x1[1] = 1
    assert plan.cyclic_poly_mult(xm, x1)[0] == 1 and plan.poly_mult(xm, x1)[0] == q - 1
    if not plan.has_fused:
        with pytest.raises(eng.TinyNttError, match="fused kernel not built"):
            plan.cyclic_poly_mult(a.astype(plan.dtype), b.astype(plan.dtype), variant="fused")
    else:                                            # canonical policy + a larger batch through the persistent kernel
        cplan = eng.get_plan(n, q, psi, 0, eng.PLAN_FORCE_CANONICAL)
        aa = rng.integers(0, q, (300, n), dtype=np.uint64).astype(plan.dtype); bb = rng.integers(0, q, (300, n), dtype=np.uint64).astype(plan.dtype)
        ref = plan.cyclic_poly_mult(aa, bb, variant="cg")
        assert np.array_equal(plan.cyclic_poly_mult(aa, bb, variant="fused"), ref)
        assert np.array_equal(cplan.cyclic_poly_mult(aa, bb, variant="fused"), ref)


@pytest.mark.parametrize("tag", ["P256", "P1024", "P4096_60"])
def test_pointwise_and_schoolbook_checker(eng, oracle, tag):
    n, q, psi = PARAMS[tag]
    plan = eng.get_plan(n, q, psi)
    rng = np.random.default_rng(5)
    word = 2 ** (8 * plan.elem_bytes) - 1
    a = rng.integers(0, q, (4, n), dtype=np.uint64); b = rng.integers(0, q, (4, n), dtype=np.uint64)
    a[0], b[0] = q - 1, q - 1
    a[1] = rng.integers(0, word, n, dtype=np.uint64, endpoint=True)                  # unreduced words
    pw = plan.pointwise_mul(a.astype(plan.dtype), b.astype(plan.dtype)).astype(np.uint64)
    expect = np.array([[int(x) * int(y) % q for x, y in zip(ra, rb)] for ra, rb in zip(a, b)], dtype=np.uint64)
    assert np.array_equal(pw, expect)
    sb = plan.schoolbook(a.astype(plan.dtype), b.astype(plan.dtype)).astype(np.uint64)
    for r in range(4):
        assert np.array_equal(sb[r], oracle.schoolbook(a[r], b[r], q)), (tag, r)
    # the O(n^2) kernel and the NTT kernels agree on device: the reference's --check (benchmark_ntt_60bit.cpp:215-223)
    assert np.array_equal(sb, plan.poly_mult(a.astype(plan.dtype), b.astype(plan.dtype)).astype(np.uint64))


def test_device_check_like_reference_dash_dash_check(eng):
    import torch
    plan = eng.get_plan(*PARAMS["P4096_60"])
    a = plan.fill_lcg(8, 1, 2); b = plan.fill_lcg(8, 2, 2)
    assert torch.equal(plan.schoolbook(a, b), plan.poly_mult(a, b))


def test_plan_tables_equal_reference_hex_files(eng):
    import hashlib
    from tiny_ntt_amd import twiddles
    with open(os.path.join(GOLDEN, "reference_hex_digests.json")) as f:
        digests = json.load(f)
    for name, d in digests.items():
        if d["kind"] != "fwd":
            continue
        n, q, psi = PARAMS[d["tag"]]
        plan = eng.get_plan(n, q, psi)
        table = plan.export_table("psi_pow")                                          # what the kernels actually multiply by
        upper = d["first"][1] != d["first"][1].lower()
        assert hashlib.sha256(twiddles.format_hex(table, q, uppercase=upper).encode()).hexdigest() == d["sha256"], name
    n, q, psi = PARAMS["P4096_60"]
    plan = eng.get_plan(n, q, psi)
    inv, ninv = twiddles.inverse_table(n, q, psi), pow(n, q - 2, q)
    assert [int(v) for v in plan.export_table("psi_inv_ninv")] == [x * ninv % q for x in inv]
    om = plan.export_table("omega_pow")
    assert len(om) == n // 2 and int(om[1]) == psi * psi % q
    brv = plan.export_table("psi_brv")
    assert int(brv[1]) == pow(psi, n // 2, q) and int(brv[2]) == pow(psi, n // 4, q)


def test_plan_from_hex_file(eng, tmp_path):
    from tiny_ntt_amd import twiddles
    n, q, psi = PARAMS["P1024"]
    path = tmp_path / "twiddle_forward_1024.hex"
    twiddles.write_hex(str(path), twiddles.forward_table(n, q, psi), q, uppercase=False)
    plan = twiddles.plan_from_hex(str(path), q)
    assert (plan.n, plan.psi) == (n, psi)


@pytest.mark.parametrize("mode", ["cyclic", "negacyclic"])
def test_rocc_session_protocol(eng, oracle, mode):
    from tiny_ntt_amd import rocc
    n, q, psi = PARAMS["P4096"]                     # the accelerator's configuration: chipyard/ntt-test.c:21, 32-bit coefficients
    s = rocc.NttRoccSession(n, q, psi, mode=mode)
    assert s.rocc(rocc.FUNCT_STATUS) == 0
    a, b = [1, 2, 3], [5, 1]                        # chipyard/ntt-test.c:91-108
    c = s.multiply(a, b)
    assert c[:5] == [5, 11, 17, 3, 0] and not any(c[5:])
    st = s.rocc(rocc.FUNCT_STATUS)
    assert st & rocc.STATUS_DONE and not st & rocc.STATUS_BUSY and (st >> 4) & 0xF == rocc.STATE_DONE
    assert st & rocc.STATUS_FWD_DONE and st & rocc.STATUS_INV_DONE
    # debug memories expose the forward transforms of A and B (funct 5/6)
    A = oracle.cg_ntt(np.array(a + [0] * (n - 3), dtype=np.uint64), psi * psi % q, q)
    assert [s.rocc(rocc.FUNCT_DEBUG_READ_A, i) for i in (0, 1, 777, n - 1)] == [int(A[i]) for i in (0, 1, 777, n - 1)]
    # addresses wrap to addrWidth bits; data is TRUNCATED to the 32-bit coefficient width and stored unreduced
    # (load_data := rs2(nttWidth-1, 0), NttRocc.scala:187, nttWidth = 32 :95); unknown funct answers 0
    s.rocc(rocc.FUNCT_LOAD_A, n + 1, q + 7)
    assert s.width == 32 and s._a[1] == q + 7 and s.rocc(99, 1, 2) == 0
    s.rocc(rocc.FUNCT_LOAD_A, 2, (5 << 32) | 9)
    assert s._a[2] == 9
    s.rocc(rocc.FUNCT_LOAD_A, 1, 0); s.rocc(rocc.FUNCT_LOAD_A, 2, 0)
    # an unreduced stored word is taken mod q when used: (q + 3) * x^0 times b == 3 b
    s2 = rocc.NttRoccSession(n, q, psi, mode=mode)
    assert s2.multiply([q + 3], [1, 1])[:3] == [3, 3, 0]
    rng = np.random.default_rng(3)
    ra = rng.integers(0, q, n, dtype=np.uint64); rb = rng.integers(0, q, n, dtype=np.uint64)
    got = np.array(s.multiply(ra, rb), dtype=np.uint64)
    ref = cyclic_oracle(oracle, ra, rb, q, psi * psi % q) if mode == "cyclic" else oracle.poly_mult(ra, rb, q, psi)
    assert np.array_equal(got, ref)


def test_benchmark_cli_twin_reports_like_the_reference_binary():
    """tools/benchmark_ntt_gpu.c: the software_benchmark CLI over the C ABI (plain C99).  Same keys as the reference
    binary (compiled unmodified under oracle/_ref), same checksums for the reference's own input pair (row 0)."""
    import os, subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tiny_ntt_amd", "lib", "benchmark_ntt_gpu")
    assert os.path.exists(exe), "make -C tiny_ntt_amd/csrc builds it"

    def kv(out):
        return dict(l.split("=", 1) for l in out.splitlines() if "=" in l and " " not in l)

    r = subprocess.run([exe, "--reps", "3", "--batch", "300", "--check"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = kv(r.stdout)
    assert int(got["forward_ntt_checksum"]) == REF_CHECKSUMS["P4096_60"][0] and int(got["checksum"]) == REF_CHECKSUMS["P4096_60"][1]
    assert "check=ok rows=4" in r.stdout
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "benchmark_ntt_60bit_scalar")
    if os.path.exists(ref_exe):                       # built from the reference sources in the build container; travels with the snapshot
        ref = kv(subprocess.run([ref_exe, "--reps", "1"], stdout=subprocess.PIPE, text=True, timeout=300).stdout)
        assert set(ref) <= set(got), (sorted(ref), sorted(got))
        assert ref["checksum"] == got["checksum"] and ref["forward_ntt_checksum"] == got["forward_ntt_checksum"]
    # the 24-bit benchmark configuration (CMakeLists.txt:5-7) through the same binary
    n, q, psi = PARAMS["P4096"]
    r = subprocess.run([exe, "--reps", "2", "--batch", "64", "--n", str(n), "--q", str(q), "--psi", str(psi), "--check"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = kv(r.stdout)
    assert int(got["forward_ntt_checksum"]) == REF_CHECKSUMS["P4096"][0] and int(got["checksum"]) == REF_CHECKSUMS["P4096"][1]
    assert subprocess.run([exe, "--bogus"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 2
    # --simple = the other benchmark family (benchmark_simple_60bit.cpp / benchmark_simple.cpp): the O(n^2) direct product timed,
    # the same five lines; checksum against the reference binary compiled under oracle/_ref
    for args, refname in ((["--simple", "--reps", "1", "--batch", "8"], "benchmark_simple_60bit_scalar"),
                          (["--simple", "--reps", "1", "--batch", "8", "--n", "1024", "--q", "8380417", "--psi", "5548360"],
                           "benchmark_simple_1024_scalar")):
        out = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.strip().split("\n")
        assert lines[0] == "benchmark_simple_gpu" and [l.split("=")[0] for l in lines[2:]] == ["total_ns", "avg_ns", "checksum"]
        simple_ref = os.path.join(ROOT, "oracle", "_ref", refname)
        if os.path.exists(simple_ref):
            rl = subprocess.run([simple_ref, "--reps", "1"], stdout=subprocess.PIPE, text=True, timeout=120).stdout.strip().split("\n")
            assert [l.split("=")[0] for l in rl[2:]] == ["total_ns", "avg_ns", "checksum"]
            assert kv(out.stdout)["checksum"] == kv("\n".join(rl))["checksum"]
        else:
            assert kv(out.stdout)["checksum"] == ("2710933653778106521" if refname.endswith("60bit_scalar") else "15308795525113097448")


def test_omega_only_plans_follow_the_reference_for_any_omega(eng):
    """cg_ntt / cg_intt of the reference evaluate their butterflies for ANY omega_n (cg_ntt.py:29-75): outputs generated by
    importing the reference (tests/golden/make_golden.py: general_omega_cases) for non-roots, roots of lower order, 0, 1,
    and an omega without a square root psi; mirrored here through the omega-only plans (tn_plan_create_omega)."""
    import json
    import tiny_ntt_amd.cg_ntt as cg
    import tiny_ntt_amd.cg_ntt_8butterfly as cg8
    from conftest import GOLDEN as GOLDEN_DIR
    meta = json.load(open(os.path.join(GOLDEN_DIR, "golden_general_omega.json")))
    arrs = np.load(os.path.join(GOLDEN_DIR, "golden_general_omega.npz"))
    oldN, oldQ = cg.N, cg.Q
    try:
        for m in meta:
            cg.N, cg.Q = m["n"], m["q"]
            x = [int(v) for v in arrs[m["name"] + "_x"]]
            X = [int(v) for v in arrs[m["name"] + "_X"]]
            assert cg.cg_ntt(x, m["omega"], m["q"]) == X, m
            assert cg8.cg_ntt_8butterfly(x, m["omega"], m["q"]) == X, m
            if m.get("has_inverse"):
                inv = [int(v) for v in arrs[m["name"] + "_inv"]]
                assert cg.cg_intt(X, m["omega"], m["q"]) == inv, m
                assert cg8.cg_intt_8butterfly(X, m["omega"], m["q"]) == inv, m
    finally:
        cg.N, cg.Q = oldN, oldQ
    # what an omega-only plan refuses
    plan = eng.get_omega_plan(256, 8380417, 3)
    assert plan.omega_only and not plan.has_fused and plan.omega == 3
    z = np.zeros((1, 256), dtype=plan.dtype)
    for call in (lambda: plan.poly_mult(z, z), lambda: plan.cyclic_poly_mult(z, z), lambda: plan.twisted_ntt_forward(z),
                 lambda: plan.ntt_forward(z, variant="fused"), lambda: plan.export_table("psi_pow")):
        with pytest.raises(eng.TinyNttError):
            call()
    assert int(plan.export_table("omega_pow")[2]) == 9


def test_general_plans_follow_the_reference_for_any_psi_and_modulus(eng, oracle):
    """nwc_poly_mult(a, b, psi_2n) of the reference validates neither psi_2n nor the modulus (cg_ntt.py:78-92): outputs generated
    by importing it (tests/golden/make_golden.py: general_psi_cases) for non-root psi (0, 1, -1, low-order roots), composite and
    even moduli, through the mirror modules (which fall back to tn_plan_create_general when tn_plan_create rejects the
    parameters) and through the batched engine with every lane grouping; cg_ntt / cg_intt with composite moduli."""
    import json
    import tiny_ntt_amd.cg_ntt as cg
    import tiny_ntt_amd.cg_ntt_8butterfly as cg8
    from conftest import GOLDEN as GOLDEN_DIR
    meta = json.load(open(os.path.join(GOLDEN_DIR, "golden_general_psi.json")))
    arrs = np.load(os.path.join(GOLDEN_DIR, "golden_general_psi.npz"))
    oldN, oldQ = cg.N, cg.Q
    try:
        for m in meta:
            cg.N, cg.Q = m["n"], m["q"]
            if m["kind"] == "poly_mult":
                a, b, c = ([int(v) for v in arrs[m["name"] + s]] for s in ("_a", "_b", "_c"))
                assert cg.nwc_poly_mult(a, b, m["psi"]) == c, m
                assert cg8.nwc_poly_mult_8butterfly(a, b, m["psi"]) == c, m
                plan = eng.get_poly_plan(m["n"], m["q"], m["psi"])
                if pow(m["psi"], m["n"], m["q"]) != m["q"] - 1 or not is_prime(m["q"]):
                    assert plan.general and not plan.has_fused
                    # batched, unreduced words, every lane grouping: the oracle (pinned on these very fixtures) as checker
                    rng = np.random.default_rng(5)
                    word = 2 ** 32 - 1 if plan.elem_bytes == 4 else 2 ** 64 - 1
                    A = rng.integers(0, word, (5, m["n"]), dtype=np.uint64, endpoint=True); B = rng.integers(0, m["q"], (5, m["n"]), dtype=np.uint64)
                    ref = oracle.poly_mult(A, B, m["q"], m["psi"])
                    for v in ("cg", "cg2", "cg4", "cg8", "cg8_padded", "cg_swizzled"):
                        assert np.array_equal(plan.poly_mult(A.astype(plan.dtype), B.astype(plan.dtype), variant=v), ref), (m, v)
                    with pytest.raises(eng.TinyNttError):
                        plan.poly_mult(A.astype(plan.dtype), B.astype(plan.dtype), variant="fused")
            else:
                x, X, inv = ([int(v) for v in arrs[m["name"] + s]] for s in ("_x", "_X", "_inv"))
                assert cg.cg_ntt(x, m["omega"], m["q"]) == X, m
                assert cg8.cg_ntt_8butterfly(x, m["omega"], m["q"]) == X, m
                assert cg.cg_intt(X, m["omega"], m["q"]) == inv, m
    finally:
        cg.N, cg.Q = oldN, oldQ
    # a general plan on VALID parameters gives what the validated plan gives
    n, q, psi = PARAMS["P256"]
    g = eng.get_general_plan(n, q, psi)
    rng = np.random.default_rng(6)
    A = rng.integers(0, q, (3, n), dtype=np.uint64).astype(g.dtype); B = rng.integers(0, q, (3, n), dtype=np.uint64).astype(g.dtype)
    assert np.array_equal(g.poly_mult(A, B), eng.get_plan(n, q, psi).poly_mult(A, B))
    with pytest.raises(eng.TinyNttError):
        g.export_table("psi_brv")


def test_multi_device_entry_points_with_two_entries_on_one_gpu(eng, oracle):
    """tn_multi_*: one host call sharded over several device entries, each with its own plan and stream, no collective
    (SURVEY.md §8e).  A one-GPU box lists device 0 twice (and three times with an uneven batch): every row must equal the oracle,
    and the device-resident form must enqueue on both entries' streams."""
    import torch
    n, q, psi = PARAMS["P4096_60"]
    rng = np.random.default_rng(21)
    for entries, batch in ((2, 64), (3, 101), (2, 1)):
        mp = eng.MultiPlan(n, q, psi, devices=[0] * entries)
        assert mp.size == entries and mp.devices == [0] * entries
        spans = [mp.shard(batch, i) for i in range(entries)]
        assert sum(r for _, r in spans) == batch and spans[0][0] == 0
        a = rng.integers(0, q, (batch, n), dtype=np.uint64); b = rng.integers(0, q, (batch, n), dtype=np.uint64)
        ref = oracle.poly_mult(a, b, q, psi)
        assert np.array_equal(mp.poly_mult(a, b), ref)
        assert np.array_equal(mp.poly_mult(a, b, variant="cg8"), ref)
        # device-resident form: per-entry pointers, enqueue on each entry's own stream, then tn_multi_synchronize
        lib = mp._lib
        ta = [torch.from_numpy(a[f:f + r].view(np.int64)).to("cuda:0") for f, r in spans]
        tb = [torch.from_numpy(b[f:f + r].view(np.int64)).to("cuda:0") for f, r in spans]
        tc = [torch.empty_like(t) for t in ta]
        torch.cuda.synchronize()
        import ctypes
        vp = ctypes.c_void_p
        pa = (vp * entries)(*[t.data_ptr() if t.numel() else None for t in ta])
        pb = (vp * entries)(*[t.data_ptr() if t.numel() else None for t in tb])
        pc = (vp * entries)(*[t.data_ptr() if t.numel() else None for t in tc])
        rows = (ctypes.c_size_t * entries)(*[r for _, r in spans])
        assert lib.tn_multi_poly_mult_dev(mp._h, pa, pb, pc, rows, 0) == eng.TN_OK, lib.tn_multi_last_error()
        assert lib.tn_multi_synchronize(mp._h) == eng.TN_OK
        got = np.concatenate([t.cpu().numpy().view(np.uint64) for t in tc if t.numel()])
        assert np.array_equal(got, ref)
        mp.close()
    with pytest.raises(eng.TinyNttError, match="psi"):
        eng.MultiPlan(n, q, psi + 1, devices=[0, 0])
    with pytest.raises(eng.TinyNttError):
        eng.MultiPlan(n, q, psi, devices=[0, 99])


def test_two_host_threads_share_one_plan(eng, oracle):
    """*_host entry points use the plan's staging buffers, streams and events: they take a per-plan lock, so two host
    threads calling into ONE plan (with different batch sizes, which regrows the staging scratch) get correct results."""
    import threading
    n, q, psi = PARAMS["P1024"]
    plan = eng.get_plan(n, q, psi)
    rng = np.random.default_rng(77)
    jobs = []
    for rows in (3, 700, 40, 1500, 9, 300):
        a = rng.integers(0, q, (rows, n), dtype=np.uint64).astype(plan.dtype); b = rng.integers(0, q, (rows, n), dtype=np.uint64).astype(plan.dtype)
        jobs.append((a, b))
    results, errors = {}, []

    def work(tid):
        try:
            for rep in range(3):
                for j, (a, b) in enumerate(jobs):
                    if (j + tid) % 2 == 0:
                        results[(tid, rep, j)] = plan.poly_mult(a, b)
                    else:
                        results[(tid, rep, j)] = plan.ntt_inverse(plan.ntt_forward(a))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    ref = [oracle.poly_mult(a.astype(np.uint64), b.astype(np.uint64), q, psi) for a, b in jobs]
    for (tid, rep, j), got in results.items():
        if (j + tid) % 2 == 0:
            assert np.array_equal(got.astype(np.uint64), ref[j]), (tid, rep, j)
        else:
            assert np.array_equal(got, jobs[j][0]), (tid, rep, j)


def test_overlapping_output_is_rejected_and_many_streams_stay_correct(eng):
    import torch
    n, q, psi = PARAMS["P4096_60"]
    plan = eng.get_plan(n, q, psi)
    buf = plan.fill_lcg(12, 1, 1)
    a, b = buf[0:4], buf[4:8]
    with pytest.raises(eng.TinyNttError, match="overlap"):
        plan.poly_mult(a, b, out=buf[2:6])                 # c overlaps the tail of a and the head of b
    with pytest.raises(eng.TinyNttError, match="overlap"):
        plan.ntt_forward(buf[0:4], out=buf[3:7])
    # launches on several streams at once share the scheduler ring: every result must still be right
    rows = 20000                                           # large enough for dynamic row scheduling
    A = plan.fill_lcg(rows, 1, 2); B = plan.fill_lcg(rows, 2, 2)
    ref = plan.poly_mult(A, B)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = [torch.empty_like(A) for _ in streams]
    for rep in range(3):
        for s, o in zip(streams, outs):
            plan.poly_mult(A, B, out=o, stream=s)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)


def test_omega_only_plan_at_the_largest_32bit_size(eng, oracle):
    """n = 8192 with the 23-bit modulus has no negacyclic psi (2n does not divide q - 1) but a cyclic omega (n | q - 1): the
    omega-only plan runs the constant-geometry kernels with four lane-steps per thread and stage (CgShape::ITERS)."""
    n, q = 8192, 8380417
    g = next(x for x in range(2, 50) if pow(x, (q - 1) // 2, q) == q - 1)          # a quadratic non-residue
    omega = pow(g, (q - 1) // n, q)
    assert pow(omega, n // 2, q) == q - 1
    plan = eng.get_omega_plan(n, q, omega)
    rng = np.random.default_rng(13)
    x = rng.integers(0, q, (5, n), dtype=np.uint64).astype(plan.dtype)
    for v in ("cg", "cg8", "cg4_swizzled", "cg2_padded"):
        X = plan.ntt_forward(x, variant=v)
        for r in range(5):
            assert np.array_equal(X[r].astype(np.uint64), oracle.cg_ntt(x[r].astype(np.uint64), omega, q)), (v, r)
        assert np.array_equal(plan.ntt_inverse(X, variant=v), x), v
    out, trace = plan.ntt_forward_trace(x[0], variant="cg")
    ref, rtrace = oracle.cg_ntt(x[0].astype(np.uint64), omega, q, trace=True)
    assert np.array_equal(out.astype(np.uint64), ref) and np.array_equal(trace.astype(np.uint64), rtrace)
