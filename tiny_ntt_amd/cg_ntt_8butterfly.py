"""Mirror of new_reference/cg_ntt_8butterfly.py: the same transforms with butterflies issued
8 per lane-step (TN_VARIANT_CG8 kernels).  N and Q are read from the cg_ntt mirror at call
time, so setting tiny_ntt_amd.cg_ntt.N / .Q configures both modules.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

from . import cg_ntt as _base
from .cg_ntt import bit_reverse_list, modinv  # noqa: F401  (re-exported like the reference's import, :5)

VARIANT = "cg8"


def __getattr__(name):          # N, Q follow tiny_ntt_amd.cg_ntt (the reference binds them at import, :5)
    if name in ("N", "Q"):
        return getattr(_base, name)
    raise AttributeError(name)


def butterfly(a: int, b: int, omega: int, modulus: int = None) -> Tuple[int, int]:
    """cg_ntt_8butterfly.py:8-10 — scalar helper of the interface (host ints; not the data path)."""
    modulus = _base.Q if modulus is None else modulus
    t = (omega * b) % modulus
    return (a + t) % modulus, (a - t) % modulus


def butterfly_batch(a_vals: Sequence[int], b_vals: Sequence[int], omega_vals: Sequence[int],
                    modulus: int = None) -> Tuple[List[int], List[int]]:
    """cg_ntt_8butterfly.py:13-27: exactly 8 lanes, else ValueError."""
    if not (len(a_vals) == len(b_vals) == len(omega_vals) == 8):
        raise ValueError("Expected 8 butterfly lanes")
    pairs = [butterfly(a_vals[i], b_vals[i], omega_vals[i], modulus) for i in range(8)]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def cg_ntt_8butterfly(a_prime: List[int], omega_n: int, modulus: int = None, verbose: bool = False,
                      log_fn=print) -> List[int]:
    """cg_ntt_8butterfly.py:41-97."""
    return _base.cg_ntt(a_prime, omega_n, modulus, verbose, log_fn, _variant=VARIANT)


def cg_intt_8butterfly(A: List[int], omega_n: int, modulus: int = None) -> List[int]:
    """cg_ntt_8butterfly.py:100-104."""
    return _base.cg_intt(A, omega_n, modulus, _variant=VARIANT)


def nwc_poly_mult_8butterfly(a: List[int], b: List[int], psi_2n: int) -> List[int]:
    """cg_ntt_8butterfly.py:107-121."""
    return _base.nwc_poly_mult(a, b, psi_2n, _variant=VARIANT)
