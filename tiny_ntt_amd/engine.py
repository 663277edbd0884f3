"""ctypes binding of libtinyntt.so (include/tinyntt.h) — the thin Python shim over the C ABI.

Host-side mirror of the reference's operator for the hot path:
`nwc_poly_mult(a, b, psi_2n) -> c` (new_reference/cg_ntt.py:78-92) and the
transforms around it, batched.  All compute happens in the HIP kernels behind
the C ABI; this module only marshals pointers.  There is no CPU fallback: if the
library is missing or no HIP device is visible, calls raise.

Buffers may be
  * numpy arrays (host)  -> the *_host entry points (H2D, kernel, D2H, sync), or
  * torch tensors on a CUDA/HIP device -> the *_dev entry points (enqueue on the
    plan's stream or on a given torch stream; no copies).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TINYNTT_LIB") or os.path.join(_HERE, "lib", "libtinyntt.so")   # env override: developer A/B builds

# tn_status (include/tinyntt.h)
TN_OK, TN_EBADLEN, TN_EBADPARAM, TN_ENODEVICE, TN_EHIP, TN_ENOMEM, TN_EINVAL, TN_EUNSUPPORTED = range(8)
# tn_variant
VARIANT_AUTO, VARIANT_FUSED, VARIANT_CG, VARIANT_CG8, VARIANT_CG8_PADDED = range(5)
VARIANTS = {"auto": VARIANT_AUTO, "fused": VARIANT_FUSED, "cg": VARIANT_CG, "cg8": VARIANT_CG8, "cg8_padded": VARIANT_CG8_PADDED,
            # lane-grouping x LDS-layout sweep of the constant-geometry kernel (BASELINE config 5)
            "cg_swizzled": 5, "cg8_swizzled": 6, "cg2": 7, "cg2_padded": 8, "cg2_swizzled": 9, "cg4": 10, "cg4_padded": 11, "cg4_swizzled": 12}
CG_VARIANTS = tuple(k for k in VARIANTS if k.startswith("cg"))
PLAN_FORCE_CANONICAL = 1
PLAN_CANONICAL_INPUTS = 2     # the caller promises inputs in [0, q): include/tinyntt.h TN_PLAN_CANONICAL_INPUTS

# Every symbol include/tinyntt.h declares (tests check the built library exports them all).
EXPORTED_SYMBOLS = (
    "tn_plan_create", "tn_plan_create_omega", "tn_plan_create_general", "tn_plan_is_general", "tn_plan_destroy", "tn_plan_n", "tn_plan_q", "tn_plan_psi", "tn_plan_omega",
    "tn_plan_elem_bytes", "tn_plan_device", "tn_plan_has_fused", "tn_plan_is_lazy",
    "tn_poly_mult_dev", "tn_poly_mult_host", "tn_plan_set_host_chunk_rows", "tn_cyclic_poly_mult_dev", "tn_pointwise_mul_dev", "tn_schoolbook_dev",
    "tn_plan_export_table", "tn_ntt_forward_dev", "tn_ntt_inverse_dev",
    "tn_ntt_forward_host", "tn_ntt_inverse_host", "tn_ntt_forward_trace_host", "tn_twisted_ntt_forward_dev",
    "tn_twisted_ntt_forward_host", "tn_schoolbook_host",
    "tn_fill_lcg_dev", "tn_checksum_rows_dev", "tn_plan_synchronize", "tn_time_poly_mult_dev",
    "tn_kernel_name", "tn_last_error", "tn_status_string", "tn_version", "tn_build_id",
    "tn_shard_rows", "tn_multi_create", "tn_multi_destroy", "tn_multi_size", "tn_multi_plan", "tn_multi_device",
    "tn_multi_poly_mult_host", "tn_multi_poly_mult_dev", "tn_multi_synchronize", "tn_multi_last_error",
)


class TinyNttError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libtinyntt: {message} (status {status})")
        self.status = status


_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Load libtinyntt.so (built in-tree by `make -C tiny_ntt_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built. Run `make -C tiny_ntt_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both
    # own the GPU, so when torch is installed let it load first and bind libtinyntt to that copy.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(p)
    vp, u32, u64, sz, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
    lib.tn_plan_create.argtypes = [ctypes.POINTER(vp), u32, u64, u64, ci, u32]
    lib.tn_plan_create_omega.argtypes = [ctypes.POINTER(vp), u32, u64, u64, ci, u32]
    lib.tn_plan_create_general.argtypes = [ctypes.POINTER(vp), u32, u64, u64, ci, u32]
    lib.tn_plan_destroy.argtypes = [vp]
    for name, res in (("tn_plan_n", u32), ("tn_plan_q", u64), ("tn_plan_psi", u64), ("tn_plan_omega", u64),
                      ("tn_plan_elem_bytes", u32), ("tn_plan_device", ci), ("tn_plan_has_fused", ci), ("tn_plan_is_lazy", ci),
                      ("tn_plan_is_general", ci)):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = res
    lib.tn_poly_mult_dev.argtypes = [vp, vp, vp, vp, sz, ci, vp]
    lib.tn_poly_mult_host.argtypes = [vp, vp, vp, vp, sz, ci]
    lib.tn_plan_set_host_chunk_rows.argtypes = [vp, sz]
    lib.tn_cyclic_poly_mult_dev.argtypes = [vp, vp, vp, vp, sz, ci, vp]
    lib.tn_pointwise_mul_dev.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.tn_schoolbook_dev.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.tn_plan_export_table.argtypes = [vp, ci, vp]
    for name in ("tn_ntt_forward_dev", "tn_ntt_inverse_dev", "tn_twisted_ntt_forward_dev"):
        getattr(lib, name).argtypes = [vp, vp, vp, sz, ci, vp]
    lib.tn_schoolbook_host.argtypes = [vp, vp, vp, vp, sz]
    for name in ("tn_ntt_forward_host", "tn_ntt_inverse_host", "tn_twisted_ntt_forward_host"):
        getattr(lib, name).argtypes = [vp, vp, vp, sz, ci]
    lib.tn_ntt_forward_trace_host.argtypes = [vp, vp, vp, vp, ci]
    lib.tn_fill_lcg_dev.argtypes = [vp, vp, sz, u64, u64, vp]
    lib.tn_checksum_rows_dev.argtypes = [vp, vp, vp, sz, vp]
    lib.tn_plan_synchronize.argtypes = [vp]
    lib.tn_time_poly_mult_dev.argtypes = [vp, vp, vp, vp, sz, ci, ci, ctypes.POINTER(ctypes.c_float)]
    lib.tn_kernel_name.argtypes = [vp, ci]
    lib.tn_kernel_name.restype = ctypes.c_char_p
    lib.tn_last_error.restype = ctypes.c_char_p
    lib.tn_status_string.argtypes = [ci]
    lib.tn_status_string.restype = ctypes.c_char_p
    lib.tn_version.restype = ci
    lib.tn_build_id.restype = ctypes.c_char_p
    lib.tn_shard_rows.argtypes = [sz, ci, ci, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.tn_multi_create.argtypes = [ctypes.POINTER(vp), u32, u64, u64, ctypes.POINTER(ci), ci, u32]
    lib.tn_multi_destroy.argtypes = [vp]
    lib.tn_multi_size.argtypes = [vp]
    lib.tn_multi_plan.argtypes = [vp, ci]; lib.tn_multi_plan.restype = vp
    lib.tn_multi_device.argtypes = [vp, ci]
    lib.tn_multi_poly_mult_host.argtypes = [vp, vp, vp, vp, sz, ci]
    lib.tn_multi_poly_mult_dev.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz), ci]
    lib.tn_multi_synchronize.argtypes = [vp]
    lib.tn_multi_last_error.restype = ctypes.c_char_p
    if path is None:
        _lib = lib
    return lib


def build_id() -> str:
    """sha256 prefix of the sources the loaded libtinyntt.so was compiled from (tn_build_id)."""
    return load_library().tn_build_id().decode()


def _check(lib, status: int):
    if status == TN_OK:
        return
    msg = lib.tn_last_error().decode() or lib.tn_status_string(status).decode()
    if status == TN_EBADLEN:
        raise ValueError(msg)          # the reference raises ValueError on a wrong length (cg_ntt.py:36-37,:79-80)
    raise TinyNttError(status, msg)


def _variant(v) -> int:
    if isinstance(v, str):
        return VARIANTS[v]
    return int(v)


def _is_torch(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class Plan:
    """Immutable (n, q, psi) plan on one HIP device: twiddle tables + stream.

    Replaces the reference's module constants N, Q (cg_ntt.py:5-6), the psi_2n
    argument (:78) and the constexpr tables of the C++ benchmark
    (benchmark_ntt_60bit.cpp:43-64).

    How coefficients are read (the reference takes Python ints and reduces them with %, cg_ntt.py:82-83):
      * HOST arrays / lists are VALUES: Python ints of any size and sign, and numpy signed integers, are taken mod q with
        Python's non-negative % (-1 -> q - 1); unsigned words wider than the plan's lanes are reduced, not truncated;
        floats raise TypeError.  So a numpy int64 array that was meant as a bag of 64-bit PATTERNS (words >= 2^63 showing as
        negatives, e.g. tensor.numpy() of a torch int64 tensor) must be passed as .view(np.uint64).
      * DEVICE tensors are BIT PATTERNS: torch has no unsigned 64-bit dtype, so torch.int64 / int32 storage is read as
        unsigned words (torch_dtype); the kernels take every word mod q.
    tests/test_host_logic.py::test_host_rows_take_integers_mod_q_and_refuse_floats and
    tests/test_gpu_parity.py::test_signed_host_values_and_device_bit_patterns pin both rules against the reference's %.
    """

    def __init__(self, n: int, q: int, psi: int, device: int = 0, flags: int = 0, omega: int = None, general: bool = False):
        """omega given (psi ignored): an OMEGA-ONLY plan (tn_plan_create_omega) — cg_ntt / cg_intt for any omega_n.
        general: tn_plan_create_general — nothing validated, nwc_poly_mult for ANY psi / modulus as cg_ntt.py:78-92 computes it."""
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        if not (0 <= int(n) < 2 ** 32):
            raise ValueError(f"Expected a power-of-two length >= 4, got {n}")
        if not (0 < int(q) < 2 ** 64):
            raise TinyNttError(TN_EBADPARAM, "q must be an odd prime below 2^62")
        self.omega_only = omega is not None
        self.general = bool(general) and not self.omega_only
        if self.general:
            _check(self._lib, self._lib.tn_plan_create_general(ctypes.byref(self._h), int(n), int(q), int(psi) % int(q), int(device), int(flags)))
        elif self.omega_only:
            _check(self._lib, self._lib.tn_plan_create_omega(ctypes.byref(self._h), int(n), int(q), int(omega) % int(q), int(device), int(flags)))
            psi = 0
        else:
            _check(self._lib, self._lib.tn_plan_create(ctypes.byref(self._h), int(n), int(q), int(psi) % int(q), int(device), int(flags)))
        self.n, self.q, self.psi = int(n), int(q), int(psi) % int(q)
        self.omega = int(self._lib.tn_plan_omega(self._h))
        self.elem_bytes = int(self._lib.tn_plan_elem_bytes(self._h))
        self.dtype = np.uint32 if self.elem_bytes == 4 else np.uint64
        self.device = int(device)
        self.has_fused = bool(self._lib.tn_plan_has_fused(self._h))
        self.is_lazy = bool(self._lib.tn_plan_is_lazy(self._h))
        self.logn = self.n.bit_length() - 1

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tn_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -------------------------------------------------------------
    @property
    def torch_dtype(self):
        import torch
        return torch.int32 if self.elem_bytes == 4 else torch.int64   # DEVICE tensors: bit patterns, read as unsigned words (class docstring)

    def _host_rows(self, x, name):
        arr = np.asarray(x)
        if arr.dtype == object:                     # Python ints of any size / sign: taken mod q like the reference's % (cg_ntt.py:82-83)
            arr = np.array([int(v) % self.q for v in arr.ravel()], dtype=self.dtype).reshape(arr.shape)
        elif arr.dtype.kind == "i":                 # signed words: Python's % is non-negative
            arr = np.mod(arr.astype(np.int64), self.q).astype(self.dtype) if arr.size and arr.min() < 0 else arr
        elif arr.dtype.kind not in "u":
            raise TypeError(f"{name}: coefficients must be integers, got dtype {arr.dtype}")
        if arr.dtype.itemsize > self.elem_bytes and arr.size and int(arr.max()) >= 2 ** (8 * self.elem_bytes):
            arr = np.mod(arr, self.q)               # wider words than the plan's lanes: reduce instead of truncating
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        if arr.ndim == 1:
            arr = arr[None, :]
        if arr.ndim != 2 or arr.shape[1] != self.n:
            raise ValueError(f"Expected {self.n} coefficients, got {arr.shape[-1] if arr.ndim else 0}")
        return arr

    def _dev_rows(self, t, name):
        if not t.is_cuda:
            raise TinyNttError(TN_EINVAL, f"{name}: torch tensor must live on a HIP device")
        if t.element_size() != self.elem_bytes or not t.is_contiguous():
            raise TinyNttError(TN_EINVAL, f"{name}: need a contiguous tensor of {self.elem_bytes}-byte integers")
        if t.dim() == 1:
            rows, cols = 1, t.shape[0]
        elif t.dim() == 2:
            rows, cols = t.shape
        else:
            raise ValueError(f"Expected {self.n} coefficients")
        if cols != self.n:
            raise ValueError(f"Expected {self.n} coefficients, got {cols}")
        return rows

    def _stream_ptr(self, stream):
        """HIP stream handle for the *_dev entry points.

        None  -> torch's current stream on the plan's device, so results obey normal torch stream
                 semantics (torch's default stream has handle 0 = hipStreamLegacy, passed as
                 TN_STREAM_LEGACY because NULL means "the plan's own stream" in the C ABI);
        "plan" -> the plan's own non-blocking stream (call synchronize() before reading results);
        a torch.cuda.Stream or raw integer handle -> that stream.
        """
        if isinstance(stream, str) and stream == "plan":
            return None
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device)
        h = int(getattr(stream, "cuda_stream", stream))
        return ctypes.c_void_p(h if h else 1)

    def set_host_chunk_rows(self, rows: int):
        """Rows per chunk of the H2D -> kernel -> D2H pipeline behind the host-buffer entry points (0 = automatic)."""
        _check(self._lib, self._lib.tn_plan_set_host_chunk_rows(self._h, int(rows)))

    # ---- the operator ----------------------------------------------------------
    def poly_mult(self, a, b, variant="auto", out=None, stream=None):
        """c[r] = a[r] * b[r] in Z_q[x]/(x^n+1).  nwc_poly_mult (cg_ntt.py:78-92), batched."""
        v = _variant(variant)
        if _is_torch(a):
            import torch
            rows = self._dev_rows(a, "a")
            if self._dev_rows(b, "b") != rows:
                raise ValueError(f"Expected {self.n} coefficients")
            c = out if out is not None else torch.empty_like(a)
            self._dev_rows(c, "out")
            _check(self._lib, self._lib.tn_poly_mult_dev(self._h, a.data_ptr(), b.data_ptr(), c.data_ptr(), rows, v, self._stream_ptr(stream)))
            return c
        squeeze = np.ndim(a) == 1
        ha, hb = self._host_rows(a, "a"), self._host_rows(b, "b")
        if ha.shape != hb.shape:
            raise ValueError(f"Expected {self.n} coefficients")
        if out is not None:                  # caller-provided (e.g. pinned) result buffer
            if not (isinstance(out, np.ndarray) and out.dtype == self.dtype and out.flags.c_contiguous and out.size == ha.size):
                raise ValueError("out must be a C-contiguous array of the plan's dtype with the shape of a")
            hc = out.reshape(ha.shape)
        else:
            hc = np.empty_like(ha)
        _check(self._lib, self._lib.tn_poly_mult_host(self._h, ha.ctypes.data, hb.ctypes.data, hc.ctypes.data, ha.shape[0], v))
        return hc[0] if squeeze else hc

    def _binary_dev(self, fn, a, b, out, stream, *extra):
        """Shared marshalling of the device-only binary operators; numpy inputs are staged through torch."""
        import torch
        host = not _is_torch(a)
        squeeze = host and np.ndim(a) == 1
        if host:
            a, b = self.to_device(a), self.to_device(b)
        rows = self._dev_rows(a, "a")
        if self._dev_rows(b, "b") != rows:
            raise ValueError(f"Expected {self.n} coefficients")
        c = out if out is not None else torch.empty_like(a)
        _check(self._lib, fn(self._h, a.data_ptr(), b.data_ptr(), c.data_ptr(), rows, *extra, self._stream_ptr(stream)))
        if host:
            res = self.to_host(c)
            return res[0] if squeeze else res
        return c

    def cyclic_poly_mult(self, a, b, variant="auto", out=None, stream=None):
        """Untwisted product forward->pointwise->inverse: python_poly_mult (test_ntt_poly_mult.py:38-43)."""
        return self._binary_dev(self._lib.tn_cyclic_poly_mult_dev, a, b, out, stream, _variant(variant))

    def pointwise_mul(self, a, b, out=None, stream=None):
        """c[i] = a[i]*b[i] mod q (cg_ntt.py:88; benchmark_ntt_60bit.cpp:142-146)."""
        return self._binary_dev(self._lib.tn_pointwise_mul_dev, a, b, out, stream)

    def schoolbook(self, a, b, out=None, stream=None):
        """O(n^2) direct negacyclic product on device (benchmark_ntt_60bit.cpp:167-180): independent checker."""
        return self._binary_dev(self._lib.tn_schoolbook_dev, a, b, out, stream)

    TABLES = {"psi_pow": 0, "psi_inv_ninv": 1, "omega_pow": 2, "omega_inv_pow": 3, "psi_brv": 4, "psi_inv_brv": 5, "psi_inv_pow": 6}

    def export_table(self, which) -> np.ndarray:
        """One of the plan's device tables as uint64 values (see tn_plan_export_table)."""
        w = self.TABLES[which] if isinstance(which, str) else int(which)
        out = np.empty(self.n // 2 if w in (2, 3) else self.n, dtype=np.uint64)
        _check(self._lib, self._lib.tn_plan_export_table(self._h, w, out.ctypes.data))
        return out

    def _ntt(self, fn_dev, fn_host, x, variant, out, stream):
        v = _variant(variant)
        if _is_torch(x):
            import torch
            rows = self._dev_rows(x, "in")
            y = out if out is not None else torch.empty_like(x)
            _check(self._lib, fn_dev(self._h, x.data_ptr(), y.data_ptr(), rows, v, self._stream_ptr(stream)))
            return y
        squeeze = np.ndim(x) == 1
        hx = self._host_rows(x, "in")
        hy = np.empty_like(hx)
        _check(self._lib, fn_host(self._h, hx.ctypes.data, hy.ctypes.data, hx.shape[0], v))
        return hy[0] if squeeze else hy

    def ntt_forward(self, x, variant="auto", out=None, stream=None):
        """cg_ntt(x, omega=psi^2) (cg_ntt.py:29-65): untwisted, natural order in and out.
        variant "auto"/"fused": register-tiled kernel; "cg"/"cg8"/"cg8_padded": the reference's stage sweep."""
        return self._ntt(self._lib.tn_ntt_forward_dev, self._lib.tn_ntt_forward_host, x, variant, out, stream)

    def ntt_inverse(self, x, variant="auto", out=None, stream=None):
        """cg_intt(x, omega=psi^2) (cg_ntt.py:68-75)."""
        return self._ntt(self._lib.tn_ntt_inverse_dev, self._lib.tn_ntt_inverse_host, x, variant, out, stream)

    def twisted_ntt_forward(self, x, variant="auto", out=None, stream=None):
        """twist + forward: forward_ntt_bench (benchmark_ntt_60bit.cpp:161-165).  Device tensors only."""
        if not _is_torch(x):
            import torch
            t = torch.from_numpy(self._host_rows(x, "in").view(np.int32 if self.elem_bytes == 4 else np.int64)).to(f"cuda:{self.device}")
            y = self._ntt(self._lib.tn_twisted_ntt_forward_dev, None, t, variant, None, None)
            self.synchronize()
            res = y.cpu().numpy().view(self.dtype)
            return res[0] if np.ndim(x) == 1 else res
        return self._ntt(self._lib.tn_twisted_ntt_forward_dev, None, x, variant, out, stream)

    def ntt_forward_trace(self, x, variant="cg"):
        """Forward transform of one polynomial plus every stage's output ([log2 n][n]),
        the lists cg_ntt(..., verbose=True) prints (cg_ntt.py:60-62)."""
        hx = self._host_rows(x, "in")
        if hx.shape[0] != 1:
            raise ValueError("ntt_forward_trace takes one polynomial")
        out = np.empty_like(hx)
        trace = np.empty((self.logn, self.n), dtype=self.dtype)
        _check(self._lib, self._lib.tn_ntt_forward_trace_host(self._h, hx.ctypes.data, out.ctypes.data, trace.ctypes.data, _variant(variant)))
        return out[0], trace

    # ---- synthetic data + digests (reference benchmark conventions) -------------
    def fill_lcg(self, batch: int, seed0: int = 1, seed_stride: int = 2, out=None, stream=None):
        """Device tensor whose row r is make_poly(seed0 + r*seed_stride) (benchmark_ntt_60bit.cpp:79-87)."""
        import torch
        t = out if out is not None else torch.empty((batch, self.n), dtype=self.torch_dtype, device=f"cuda:{self.device}")
        _check(self._lib, self._lib.tn_fill_lcg_dev(self._h, t.data_ptr(), batch, seed0 % 2 ** 64, seed_stride % 2 ** 64, self._stream_ptr(stream)))
        return t

    def checksum_rows(self, t, stream=None):
        """Per-row checksum (benchmark_ntt_60bit.cpp:182-188) as a numpy uint64 array."""
        import torch
        rows = self._dev_rows(t, "src")
        out = torch.empty((rows,), dtype=torch.int64, device=t.device)
        _check(self._lib, self._lib.tn_checksum_rows_dev(self._h, t.data_ptr(), out.data_ptr(), rows, self._stream_ptr(stream)))
        self.synchronize()
        return out.cpu().numpy().view(np.uint64)      # .cpu() waits on the current stream

    def time_poly_mult(self, a, b, c, iters: int, variant="auto") -> float:
        """Mean ms per launch over `iters` launches, HIP events on the plan's stream."""
        import torch
        rows = self._dev_rows(a, "a")
        torch.cuda.synchronize(self.device)            # inputs may have been produced on torch's stream
        ms = ctypes.c_float()
        _check(self._lib, self._lib.tn_time_poly_mult_dev(self._h, a.data_ptr(), b.data_ptr(), c.data_ptr(), rows, _variant(variant), iters, ctypes.byref(ms)))
        return float(ms.value)

    def kernel_name(self, variant="auto") -> str:
        return self._lib.tn_kernel_name(self._h, _variant(variant)).decode()

    def synchronize(self):
        _check(self._lib, self._lib.tn_plan_synchronize(self._h))

    def to_host(self, t) -> np.ndarray:
        """Device tensor -> numpy array of the plan's unsigned dtype."""
        self.synchronize()
        return t.cpu().numpy().view(self.dtype)       # .cpu() waits on the current stream

    def to_device(self, arr):
        import torch
        h = self._host_rows(arr, "arr")
        return torch.from_numpy(h.view(np.int32 if self.elem_bytes == 4 else np.int64)).to(f"cuda:{self.device}")


class MultiPlan:
    """tn_multi_*: one host call sharded over several devices in contiguous row blocks, each entry with its own plan and stream
    (no collective).  devices=None: every visible device; a device may be listed several times."""

    def __init__(self, n: int, q: int, psi: int, devices=None, flags: int = 0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*devices) if devices is not None else None
        st = self._lib.tn_multi_create(ctypes.byref(self._h), int(n), int(q), int(psi) % int(q), arr, len(devices) if devices is not None else 0, int(flags))
        if st != TN_OK:
            msg = self._lib.tn_multi_last_error().decode()
            if st == TN_EBADLEN:
                raise ValueError(msg)
            raise TinyNttError(st, msg)
        self.n, self.q, self.psi = int(n), int(q), int(psi) % int(q)
        self.size = int(self._lib.tn_multi_size(self._h))
        self.devices = [int(self._lib.tn_multi_device(self._h, i)) for i in range(self.size)]
        self.elem_bytes = int(self._lib.tn_plan_elem_bytes(self._lib.tn_multi_plan(self._h, 0)))
        self.dtype = np.uint32 if self.elem_bytes == 4 else np.uint64

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tn_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, batch: int, index: int):
        """(first_row, rows) of entry `index` (tn_shard_rows)."""
        f, r = ctypes.c_size_t(), ctypes.c_size_t()
        _check(self._lib, self._lib.tn_shard_rows(int(batch), self.size, int(index), ctypes.byref(f), ctypes.byref(r)))
        return int(f.value), int(r.value)

    def poly_mult(self, a, b, variant="auto"):
        """Host arrays [batch, n] -> [batch, n]: tn_multi_poly_mult_host."""
        ha = np.ascontiguousarray(a, dtype=self.dtype); hb = np.ascontiguousarray(b, dtype=self.dtype)
        if ha.ndim != 2 or ha.shape != hb.shape or ha.shape[1] != self.n:
            raise ValueError(f"Expected {self.n} coefficients")
        hc = np.empty_like(ha)
        st = self._lib.tn_multi_poly_mult_host(self._h, ha.ctypes.data, hb.ctypes.data, hc.ctypes.data, ha.shape[0], _variant(variant))
        if st != TN_OK:
            raise TinyNttError(st, self._lib.tn_multi_last_error().decode())
        return hc


_plan_cache = {}


def get_plan(n: int, q: int, psi: int, device: int = 0, flags: int = 0) -> Plan:
    key = (int(n), int(q), int(psi) % int(q), int(device), int(flags))
    p = _plan_cache.get(key)
    if p is None:
        p = _plan_cache[key] = Plan(*key)
    return p


def get_omega_plan(n: int, q: int, omega: int, device: int = 0) -> Plan:
    """Cached omega-only plan: the transforms cg_ntt(a, omega_n, modulus) / cg_intt for any omega_n (cg_ntt.py:29-75)."""
    key = ("omega", int(n), int(q), int(omega) % int(q), int(device))
    p = _plan_cache.get(key)
    if p is None:
        p = _plan_cache[key] = Plan(int(n), int(q), 0, int(device), 0, omega=int(omega))
    return p


def get_general_plan(n: int, q: int, psi: int, device: int = 0) -> Plan:
    """Cached general plan (tn_plan_create_general): nwc_poly_mult for any psi_2n / modulus, as cg_ntt.py:78-92 computes it."""
    key = ("general", int(n), int(q), int(psi) % int(q), int(device))
    p = _plan_cache.get(key)
    if p is None:
        p = _plan_cache[key] = Plan(int(n), int(q), int(psi), int(device), 0, general=True)
    return p


def get_poly_plan(n: int, q: int, psi: int, device: int = 0) -> Plan:
    """The plan the reference-shaped entry points use: the validated one (throughput kernels) when (q, psi) admit it,
    else the general one — the reference validates neither (cg_ntt.py:78-92)."""
    try:
        return get_plan(n, q, psi, device)
    except TinyNttError as e:
        if e.status != TN_EBADPARAM:
            raise
    return get_general_plan(n, q, psi, device)


def clear_plan_cache():
    for p in _plan_cache.values():
        p.close()
    _plan_cache.clear()
