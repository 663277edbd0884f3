"""Host logic that needs no GPU: the C-ABI library loads and exports every symbol the header
declares, parameter validation / error mapping, number theory helpers, the interface mirror's
length errors, batch sharding arithmetic."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import PARAMS, ROOT, have_gpu
from tiny_ntt_amd import engine, numtheory
import tiny_ntt_amd.cg_ntt as cg
import tiny_ntt_amd.cg_ntt_8butterfly as cg8


def header_functions():
    src = open(os.path.join(ROOT, "include", "tinyntt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tn_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = engine.load_library()
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/tinyntt.h but not exported"
    assert set(names) == set(engine.EXPORTED_SYMBOLS)
    assert lib.tn_version() == 100


def test_build_id_names_the_sources_the_library_was_built_from():
    """tn_build_id() = sha256 of every source of the library (csrc/Makefile BUILD_ID): measurements kept in profiles/ name the
    build they belong to, so a library older than its sources (or an id older than the kernels) must not go unnoticed."""
    import hashlib
    d = os.path.join(ROOT, "tiny_ntt_amd", "csrc")
    files = ["kernels.hip", "cg_part.hip", "capi.cpp", "multi.cpp", "modarith.h", "fused_core.h", "cg_core.h", "cg_kernel_impl.h", "dev_addr.h", "plan.h",
             "plan_tables.h", "../../include/tinyntt.h"]        # the order of csrc/Makefile: BUILD_ID
    h = hashlib.sha256(b"".join(open(os.path.join(d, f), "rb").read() for f in files)).hexdigest()[:16]
    assert engine.build_id() == h, "tiny_ntt_amd/lib/libtinyntt.so is stale: run make -C tiny_ntt_amd/csrc"


def test_library_links_hip_runtime_not_the_oracle():
    import subprocess
    out = subprocess.run(["readelf", "-d", engine.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "libamdhip64" in out
    assert "oracle" not in out and "emu" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", engine.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "tn_oracle" not in syms and "tn_port" not in syms


def test_plan_validation_errors_precede_device_use():
    n, q, psi = PARAMS["P4096_60"]
    with pytest.raises(ValueError, match="power-of-two"):
        engine.Plan(100, q, psi)
    with pytest.raises(ValueError):
        engine.Plan(2, q, psi)
    with pytest.raises(ValueError, match="exceeds"):
        engine.Plan(16384, q, psi)
    with pytest.raises(engine.TinyNttError, match="psi") as e:
        engine.Plan(n, q, psi + 1)
    assert e.value.status == engine.TN_EBADPARAM
    with pytest.raises(engine.TinyNttError, match="prime"):
        engine.Plan(n, q + 2, psi)
    with pytest.raises(engine.TinyNttError, match="odd"):
        engine.Plan(n, 2 ** 60, psi)
    with pytest.raises(engine.TinyNttError):
        engine.Plan(n, 2 ** 62 + 135, psi)


@pytest.mark.skipif(have_gpu(), reason="checks the no-device failure mode")
def test_no_device_fails_loudly_no_cpu_fallback():
    with pytest.raises(engine.TinyNttError, match="no HIP device") as e:
        engine.Plan(*PARAMS["P256"])
    assert e.value.status == engine.TN_ENODEVICE
    cg.N, cg.Q = 256, 8380417
    with pytest.raises(engine.TinyNttError):
        cg.nwc_poly_mult([0] * 256, [0] * 256, 1239911)


def test_missing_extension_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="not built"):
        engine.load_library(str(tmp_path / "libtinyntt.so"))


def test_mirror_length_errors_match_reference_messages():
    cg.N, cg.Q = 256, 8380417
    with pytest.raises(ValueError, match="Expected 256 coefficients, got 3"):
        cg.cg_ntt([1, 2, 3], 5)
    with pytest.raises(ValueError, match="Expected 256 coefficients, got 255"):
        cg.cg_intt([0] * 255, 5)
    with pytest.raises(ValueError, match="Expected 256 coefficients"):
        cg.nwc_poly_mult([0] * 256, [0] * 255, 1239911)
    with pytest.raises(ValueError, match="Expected 256 coefficients"):
        cg8.nwc_poly_mult_8butterfly([0] * 10, [0] * 256, 1239911)
    with pytest.raises(ValueError, match="Expected 8 butterfly lanes"):
        cg8.butterfly_batch([1] * 7, [1] * 7, [1] * 7)
    assert cg8.N == 256 and cg8.Q == 8380417
    cg.N = 1024
    assert cg8.N == 1024
    cg.N = 256


def test_scalar_helpers():
    q = 8380417
    assert cg.modinv(3, q) * 3 % q == 1 and cg.modinv(5) * 5 % cg.Q == 1
    assert cg.bit_reverse(1, 8) == 128 and cg.bit_reverse(0b1101, 4) == 0b1011
    assert cg.bit_reverse_list([0, 1, 2, 3, 4, 5, 6, 7]) == [0, 4, 2, 6, 1, 5, 3, 7]
    assert cg8.butterfly(5, 7, 3, 17) == ((5 + 21) % 17, (5 - 21) % 17)
    a, b = cg8.butterfly_batch(list(range(8)), list(range(8, 16)), [2] * 8, 97)
    assert a == [(i + 2 * (i + 8)) % 97 for i in range(8)] and b == [(i - 2 * (i + 8)) % 97 for i in range(8)]


def test_numtheory():
    for tag, (n, q, psi) in PARAMS.items():
        omega = psi * psi % q
        r = numtheory.psi_from_omega(omega, n, q)
        assert r * r % q == omega and pow(r, n, q) == q - 1
        found = numtheory.primitive_2n_root(n, q)
        assert pow(found, n, q) == q - 1
    for p in (7681, 8380417, 1152921504606830593, 97):
        for a in (2, 3, 5, 10, 1234567):
            r = numtheory.sqrt_mod(a, p)
            assert r is None or r * r % p == a % p
    with pytest.raises(ValueError, match="primitive"):
        numtheory.psi_from_omega(2, 256, 8380417)
    with pytest.raises(ValueError):
        numtheory.primitive_2n_root(4096, 7681)


def test_find_psi_returns_what_the_reference_script_returns():
    """scripts/find_psi.py:9-43: the smallest psi in [2, max_search) with psi^n == -1, else None.  Expected values were
    produced by importing that script (tests/golden/make_golden.py: find_psi_outputs)."""
    import json
    from conftest import GOLDEN as GOLDEN_DIR
    cases = json.load(open(os.path.join(GOLDEN_DIR, "reference_find_psi.json")))
    assert len(cases) >= 8 and any(c["psi"] is None for c in cases)
    for c in cases:
        assert numtheory.find_psi(c["n"], c["q"], c["max_search"]) == c["psi"], c
    lines = []
    assert numtheory.find_psi(4096, 8380417, log_fn=lines.append) == 687 and "687" in lines[0]


def test_shard_rows_partition():
    from tiny_ntt_amd import dist
    for batch in (0, 1, 7, 8, 65536, 1048576, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [dist.shard_rows(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == batch
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_c_abi_shard_rows_matches_the_python_split():
    """tn_shard_rows (the split tn_multi_* uses) = dist.shard_rows (the split bench.py uses): contiguous blocks covering the batch
    exactly once, sizes differing by at most one row.  Pure host arithmetic: no device needed."""
    from tiny_ntt_amd import dist
    lib = engine.load_library()
    for batch in (0, 1, 7, 8, 65536, 1048576, 1000003):
        for parts in (1, 2, 3, 4, 8):
            for i in range(parts):
                f, r = ctypes.c_size_t(), ctypes.c_size_t()
                assert lib.tn_shard_rows(batch, parts, i, ctypes.byref(f), ctypes.byref(r)) == engine.TN_OK
                assert (f.value, r.value) == dist.shard_rows(batch, parts, i)
    f, r = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.tn_shard_rows(8, 0, 0, ctypes.byref(f), ctypes.byref(r)) == engine.TN_EINVAL
    assert lib.tn_shard_rows(8, 2, 2, ctypes.byref(f), ctypes.byref(r)) == engine.TN_EINVAL


def test_host_rows_take_integers_mod_q_and_refuse_floats():
    """Plan._host_rows (no device needed): any Python / numpy integer is taken mod q like the reference's %, floats raise."""
    class P:                                            # stand-in with the attributes _host_rows reads
        q, dtype, elem_bytes, n = 8380417, np.uint32, 4, 4
    f = engine.Plan._host_rows
    assert f(P(), [1, 2, 3, 4], "a").tolist() == [[1, 2, 3, 4]]
    assert f(P(), [-1, 2 ** 70, 3, 4], "a").tolist() == [[8380416, 2 ** 70 % 8380417, 3, 4]]
    assert f(P(), np.array([-1, 2, 3, 4]), "a").tolist() == [[8380416, 2, 3, 4]]
    assert f(P(), np.array([2 ** 40, 2, 3, 4], dtype=np.uint64), "a").tolist() == [[2 ** 40 % 8380417, 2, 3, 4]]
    with pytest.raises(TypeError):
        f(P(), [1.5, 2, 3, 4], "a")
    with pytest.raises(ValueError):
        f(P(), [1, 2, 3], "a")
