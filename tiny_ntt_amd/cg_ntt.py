"""Drop-in mirror of the reference golden model's interface, computed on the MI355X.

Same names, argument meaning and error behaviour as orhosko/tiny-ntt's
new_reference/cg_ntt.py: module attributes N and Q (read at call time, as there),
modinv, bit_reverse, bit_reverse_list, cg_ntt, cg_intt, nwc_poly_mult.  Lists of
Python ints in and out.  Every transform runs in the HIP kernels behind
libtinyntt.so (engine.Plan); nothing here computes a butterfly on the CPU.

    import tiny_ntt_amd.cg_ntt as cg_ntt
    cg_ntt.N, cg_ntt.Q = 4096, 1152921504606830593
    c = cg_ntt.nwc_poly_mult(a, b, 431606828070683274)
"""
from __future__ import annotations

from typing import List

import numpy as np

from . import engine, numtheory

N = 256            # new_reference/cg_ntt.py:5
Q = 8380417        # new_reference/cg_ntt.py:6
DEVICE = 0
VARIANT = "cg"     # kernel schedule used for the transforms ("cg" = the reference's constant-geometry dataflow)
POLY_VARIANT = "auto"   # schedule for nwc_poly_mult ("auto" = fused throughput kernel when available)


def modinv(value: int, modulus: int = None) -> int:
    """new_reference/cg_ntt.py:9-10."""
    return numtheory.modinv(value, Q if modulus is None else modulus)


bit_reverse = numtheory.bit_reverse   # new_reference/cg_ntt.py:13-18


def bit_reverse_list(values: List[int]) -> List[int]:
    """new_reference/cg_ntt.py:21-26 (host-side index permutation, used for the verbose log)."""
    bits = (len(values) - 1).bit_length()
    out = [0] * len(values)
    for idx, val in enumerate(values):
        out[bit_reverse(idx, bits)] = val
    return out


def _as_words(values, modulus, dtype):
    # the reference accepts any Python ints and reduces with % (cg_ntt.py:55-59, :82-83)
    return np.array([int(v) % modulus for v in values], dtype=dtype)


def _plan_for_omega(n, omega_n, modulus, variant):
    """The reference evaluates cg_ntt for ANY omega_n (cg_ntt.py:29-65): the constant-geometry variants run on an
    omega-only plan (no psi needed).  The register-tiled variants ("fused" / "auto") are keyed on psi, so they need
    omega_n to be a primitive n-th root with a square root mod the modulus."""
    if variant in engine.CG_VARIANTS:
        return engine.get_omega_plan(n, modulus, omega_n, DEVICE)
    psi = numtheory.psi_from_omega(omega_n, n, modulus)
    return engine.get_plan(n, modulus, psi, DEVICE)


def cg_ntt(a_prime: List[int], omega_n: int, modulus: int = None, verbose: bool = False, log_fn=print,
           _variant=None) -> List[int]:
    """new_reference/cg_ntt.py:29-65: cyclic NTT, natural order in and out."""
    modulus = Q if modulus is None else modulus
    if len(a_prime) != N:
        raise ValueError(f"Expected {N} coefficients, got {len(a_prime)}")
    variant = _variant or VARIANT
    plan = _plan_for_omega(N, omega_n, modulus, variant)
    x = _as_words(a_prime, modulus, plan.dtype)
    if not verbose:
        return [int(v) for v in plan.ntt_forward(x, variant=variant)]
    out, trace = plan.ntt_forward_trace(x, variant=variant)
    label = "CG NTT start" if variant == "cg" else "CG NTT 8-butterfly start"
    log_fn(label)                                                     # :44 / cg_ntt_8butterfly.py:56
    log_fn(f"  omega_n={omega_n} modulus={modulus}")                  # :45
    log_fn(f"  input(first 16)={list(a_prime[:16])}")                 # :46
    log_fn(f"  bitrev(first 16)={bit_reverse_list(list(a_prime))[:16]}")   # :47
    for stage in range(1, plan.logn + 1):
        k = N >> stage
        log_fn(f"  stage={stage} k={k} omega_s={pow(omega_n, k, modulus)}")        # :61
        log_fn(f"  stage_out(first 16)={[int(v) for v in trace[stage - 1][:16]]}")  # :62
    return [int(v) for v in out]


def cg_intt(A: List[int], omega_n: int, modulus: int = None, _variant=None) -> List[int]:
    """new_reference/cg_ntt.py:68-75."""
    modulus = Q if modulus is None else modulus
    if len(A) != N:
        raise ValueError(f"Expected {N} coefficients, got {len(A)}")
    variant = _variant or VARIANT
    plan = _plan_for_omega(N, omega_n, modulus, variant)
    return [int(v) for v in plan.ntt_inverse(_as_words(A, modulus, plan.dtype), variant=variant)]


def nwc_poly_mult(a: List[int], b: List[int], psi_2n: int, _variant=None) -> List[int]:
    """new_reference/cg_ntt.py:78-92: c = a*b in Z_Q[x]/(x^N + 1)."""
    if len(a) != N or len(b) != N:
        raise ValueError(f"Expected {N} coefficients")
    plan = engine.get_poly_plan(N, Q, psi_2n, DEVICE)     # any psi_2n, any modulus: the reference validates neither
    variant = _variant or POLY_VARIANT
    if plan.general and variant in ("auto", "fused"):
        variant = VARIANT if VARIANT in engine.CG_VARIANTS else "cg"
    c = plan.poly_mult(_as_words(a, Q, plan.dtype), _as_words(b, Q, plan.dtype), variant=variant)
    return [int(v) for v in c]


def nwc_poly_mult_batch(a, b, psi_2n: int, n: int = None, q: int = None, variant="auto", device: int = None):
    """Batched form: a, b are [batch, n] numpy arrays (host) or torch device tensors."""
    n = N if n is None else n
    q = Q if q is None else q
    plan = engine.get_plan(n, q, psi_2n, DEVICE if device is None else device)
    return plan.poly_mult(a, b, variant=variant)
