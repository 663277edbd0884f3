import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# tag: (n, q, psi) — SURVEY.md §8 parameter table
PARAMS = {
    "P4": (4, 7681, 1925),
    "P256": (256, 8380417, 1239911),
    "P1024": (1024, 8380417, 5548360),
    "P4096": (4096, 8380417, 283817),
    "P4096_60": (4096, 1152921504606830593, 431606828070683274),
}
# checksums the reference C++ benchmark prints for make_poly(1) x make_poly(2)
REF_CHECKSUMS = {
    "P4096": (2800297349529693940, 11303505593119465445),
    "P4096_60": (15678418584317678507, 2710933653778106521),
    "P1024": (3555142461877891881, 15308795525113097448),
}


def is_prime(m):
    if m < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if m % p == 0:
            return m == p
    d, s = m - 1, 0
    while d % 2 == 0:
        d //= 2; s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, m)
        if x in (1, m - 1):
            continue
        for _ in range(s - 1):
            x = x * x % m
            if x == m - 1:
                break
        else:
            return False
    return True


def ntt_prime_below(limit, n):
    """largest prime q < limit with q = 1 (mod 2n)"""
    q = (limit - 2) // (2 * n) * (2 * n) + 1
    while not is_prime(q):
        q -= 2 * n
    return q


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(directory, *targets):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), *targets], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout)


P64 = ctypes.POINTER(ctypes.c_uint64)


def p64(arr):
    assert arr.dtype == np.uint64 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(P64)


class Oracle:
    """ctypes view of oracle/_build/liboracle.so (the CPU checker; tests only)."""

    def __init__(self):
        so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(so):
            _make("oracle", "all")
        L = self.lib = ctypes.CDLL(so)
        u64, sz, ci = ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
        L.tn_oracle_powmod.argtypes = [u64, u64, u64]; L.tn_oracle_powmod.restype = u64
        L.tn_oracle_modinv.argtypes = [u64, u64]; L.tn_oracle_modinv.restype = u64
        L.tn_oracle_bit_reverse_list.argtypes = [P64, P64, sz]
        for name in ("tn_oracle_cg_ntt", "tn_oracle_cg_ntt_8butterfly"):
            getattr(L, name).argtypes = [P64, P64, sz, u64, u64, P64]
        for name in ("tn_oracle_cg_intt", "tn_oracle_cg_intt_8butterfly"):
            getattr(L, name).argtypes = [P64, P64, sz, u64, u64]
        for name in ("tn_oracle_nwc_poly_mult", "tn_oracle_nwc_poly_mult_8butterfly"):
            getattr(L, name).argtypes = [P64, P64, P64, sz, u64, u64]
        L.tn_oracle_nwc_poly_mult_batch.argtypes = [P64, P64, P64, sz, sz, u64, u64]
        L.tn_oracle_negacyclic_schoolbook.argtypes = [P64, P64, P64, sz, u64]
        L.tn_port_plan_create.argtypes = [sz, u64, u64]; L.tn_port_plan_create.restype = ctypes.c_void_p
        L.tn_port_plan_destroy.argtypes = [ctypes.c_void_p]
        L.tn_port_make_poly.argtypes = [u64, P64, sz, u64]
        L.tn_port_checksum.argtypes = [P64, sz, u64]; L.tn_port_checksum.restype = u64
        L.tn_port_ntt.argtypes = [ctypes.c_void_p, P64, P64, ci]
        L.tn_port_forward_ntt_bench.argtypes = [ctypes.c_void_p, P64, P64]
        L.tn_port_negacyclic_mul_ntt.argtypes = [ctypes.c_void_p, P64, P64, P64]; L.tn_port_negacyclic_mul_ntt.restype = ci
        L.tn_port_negacyclic_mul_batch.argtypes = [ctypes.c_void_p, P64, P64, P64, sz]; L.tn_port_negacyclic_mul_batch.restype = ci

    # -- cg_ntt.py restatement --
    def cg_ntt(self, a, omega, q, group=1, trace=False):
        a = np.ascontiguousarray(a, dtype=np.uint64); n = a.size
        out = np.empty(n, dtype=np.uint64)
        tr = np.empty((n.bit_length() - 1, n), dtype=np.uint64) if trace else None
        fn = self.lib.tn_oracle_cg_ntt if group == 1 else self.lib.tn_oracle_cg_ntt_8butterfly
        assert fn(p64(a), p64(out), n, omega, q, p64(tr) if trace else None) == 0
        return (out, tr) if trace else out

    def cg_intt(self, A, omega, q, group=1):
        A = np.ascontiguousarray(A, dtype=np.uint64); n = A.size
        out = np.empty(n, dtype=np.uint64)
        fn = self.lib.tn_oracle_cg_intt if group == 1 else self.lib.tn_oracle_cg_intt_8butterfly
        assert fn(p64(A), p64(out), n, omega, q) == 0
        return out

    def poly_mult(self, a, b, q, psi, group=1):
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        if a.ndim == 2:
            c = np.empty_like(a)
            assert self.lib.tn_oracle_nwc_poly_mult_batch(p64(a), p64(b), p64(c), a.shape[0], a.shape[1], q, psi) == 0
            return c
        c = np.empty_like(a)
        fn = self.lib.tn_oracle_nwc_poly_mult if group == 1 else self.lib.tn_oracle_nwc_poly_mult_8butterfly
        assert fn(p64(a), p64(b), p64(c), a.size, q, psi) == 0
        return c

    def schoolbook(self, a, b, q):
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.empty_like(a)
        self.lib.tn_oracle_negacyclic_schoolbook(p64(a), p64(b), p64(out), a.size, q)
        return out

    # -- software_benchmark restatement --
    def make_poly(self, seed, n, q):
        out = np.empty(n, dtype=np.uint64)
        self.lib.tn_port_make_poly(seed, p64(out), n, q)
        return out

    def checksum(self, poly, q):
        poly = np.ascontiguousarray(poly, dtype=np.uint64)
        return int(self.lib.tn_port_checksum(p64(poly), poly.size, q))

    def port_plan(self, n, q, psi):
        h = self.lib.tn_port_plan_create(n, q, psi)
        return h

    def port_mul(self, n, q, psi, a, b):
        h = self.lib.tn_port_plan_create(n, q, psi); assert h
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        c = np.empty_like(a)
        if a.ndim == 2:
            assert self.lib.tn_port_negacyclic_mul_batch(h, p64(a), p64(b), p64(c), a.shape[0]) == 0
        else:
            assert self.lib.tn_port_negacyclic_mul_ntt(h, p64(a), p64(b), p64(c)) == 0
        self.lib.tn_port_plan_destroy(h)
        return c

    def port_forward_bench(self, n, q, psi, a):
        h = self.lib.tn_port_plan_create(n, q, psi); assert h
        a = np.ascontiguousarray(a, dtype=np.uint64); out = np.empty_like(a)
        self.lib.tn_port_forward_ntt_bench(h, p64(a), p64(out))
        self.lib.tn_port_plan_destroy(h)
        return out

    def port_ntt(self, n, q, psi, a, inverse=False):
        h = self.lib.tn_port_plan_create(n, q, psi); assert h
        a = np.ascontiguousarray(a, dtype=np.uint64); out = np.empty_like(a)
        self.lib.tn_port_ntt(h, p64(a), p64(out), 1 if inverse else 0)
        self.lib.tn_port_plan_destroy(h)
        return out


class Emu:
    """ctypes view of tests/emu/_build/libemu.so (CPU stepping of the kernels' per-thread code)."""

    def __init__(self):
        so = os.path.join(ROOT, "tests", "emu", "_build", "libemu.so")
        if not os.path.exists(so):
            _make("tests/emu")
        L = self.lib = ctypes.CDLL(so)
        u32, u64, sz, ci = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
        L.emu_fused_poly_mult.argtypes = [u32, u64, u64, ci, P64, P64, P64, sz]
        L.emu_is_lazy.argtypes = [u32, u64, u64]
        L.emu_cg.argtypes = [u32, u64, u64, ci, P64, P64, P64, P64]
        L.emu_cgm.argtypes = [u32, u64, u64, ci, ci, ci, ci, ci, P64, P64, P64, P64]
        L.emu_fused_ntt.argtypes = [u32, u64, u64, ci, ci, P64, P64]
        L.emu_mul_tw64.argtypes = [u64, u64, u64]; L.emu_mul_tw64.restype = u64
        L.emu_mul_tw64_lazy.argtypes = [u64, u64, u64]; L.emu_mul_tw64_lazy.restype = u64
        L.emu_mul_tw32.argtypes = [u32, u32, u32]; L.emu_mul_tw32.restype = u32
        L.emu_barrett64.argtypes = [u64, u64, u64]; L.emu_barrett64.restype = u64
        L.emu_barrett32.argtypes = [u32, u32, u32]; L.emu_barrett32.restype = u32
        L.emu_fold64.argtypes = [u64, u64]; L.emu_fold64.restype = u64
        L.emu_pw_fast_ok.argtypes = [u64]; L.emu_pw_fast_ok.restype = ctypes.c_int
        L.emu_pointwise_lazy64.argtypes = [u64, u64, u64]; L.emu_pointwise_lazy64.restype = u64
        L.emu_fold32.argtypes = [u32, u32]; L.emu_fold32.restype = u32
        L.emu_mul_sp_acc.argtypes = [u64, u64, u64, u64]; L.emu_mul_sp_acc.restype = u64
        L.emu_split_sched_ok.argtypes = [u32, u64]; L.emu_split_sched_ok.restype = ctypes.c_int
        L.emu_split_sched_stat.argtypes = [ctypes.c_int, ctypes.c_int]; L.emu_split_sched_stat.restype = ctypes.c_long

    def fused(self, n, q, psi, a, b, canonical=False, cyclic=False, promised_canonical_inputs=False):
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        a2, b2 = np.atleast_2d(a), np.atleast_2d(b)
        c = np.empty_like(a2)
        rc = self.lib.emu_fused_poly_mult(n, q, psi, int(canonical) | (2 if cyclic else 0) | (4 if promised_canonical_inputs else 0),
                                          p64(a2), p64(b2), p64(c), a2.shape[0])
        if rc == 7:
            return None
        assert rc == 0, rc
        return c.reshape(a.shape)

    def fused_ntt(self, n, q, psi, mode, x, canonical=False):
        """mode 0: twist + forward, 1: cg_ntt, 2: cg_intt — register-tiled standalone transforms (natural order)."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        out = np.empty_like(x)
        rc = self.lib.emu_fused_ntt(n, q, psi, int(canonical), mode, p64(x), p64(out))
        assert rc == 0, rc
        return out

    def cg(self, n, q, psi, mode, a, b=None, trace=False):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
        out = np.empty_like(a)
        tr = np.empty((n.bit_length() - 1, n), dtype=np.uint64) if trace else None
        rc = self.lib.emu_cg(n, q, psi, mode, p64(a), p64(b) if b is not None else None, p64(out), p64(tr) if trace else None)
        assert rc == 0
        return (out, tr) if trace else out

    def cgm(self, n, q, psi, mode, a, b=None, group=8, layout=0, am=0, flags=0, trace=False):
        """The multi-stage trips of cg_kernels.hip stepped on the CPU.  Returns None when the combination is unsupported
        (log2 n < log2(2 group); split arithmetic on a plan that is not lazy with 64-bit lanes)."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
        out = np.empty_like(a)
        tr = np.empty((n.bit_length() - 1, n), dtype=np.uint64) if trace else None
        rc = self.lib.emu_cgm(n, q, psi, mode, group, layout, am, flags, p64(a), p64(b) if b is not None else None, p64(out),
                              p64(tr) if trace else None)
        if rc == 7:
            return None
        assert rc == 0, rc
        return (out, tr) if trace else out


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def emu():
    return Emu()


class Golden:
    def __init__(self, tag):
        self.tag = tag
        self.n, self.q, self.psi = PARAMS[tag]
        self.arr = np.load(os.path.join(GOLDEN, f"golden_{tag}.npz"))     # allow_pickle=False (default)
        with open(os.path.join(GOLDEN, f"golden_{tag}.json")) as f:
            self.meta = json.load(f)
        assert (self.meta["n"], self.meta["q"], self.meta["psi"]) == (self.n, self.q, self.psi)
        self.omega = self.meta["omega"]

    def cases(self, kind):
        return [c["name"] for c in self.meta["cases"] if c["kind"] == kind]

    def __getitem__(self, key):
        return self.arr[key]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(tag):
        if tag not in cache:
            cache[tag] = Golden(tag)
        return cache[tag]
    return get


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
