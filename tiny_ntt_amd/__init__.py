"""tiny_ntt_amd — MI355X-native batched negacyclic polynomial multiplication.

Drop-in for the hot path of orhosko/tiny-ntt: `nwc_poly_mult(a, b, psi_2n) -> c`
(new_reference/cg_ntt.py:78) and the transforms around it, as hand-written HIP
kernels for gfx950 behind a C ABI (include/tinyntt.h).  See DESIGN.md.
"""
from . import dist, engine, numtheory, twiddles    # noqa: F401
from .engine import Plan, TinyNttError, get_plan   # noqa: F401

__all__ = ["dist", "engine", "numtheory", "Plan", "TinyNttError", "get_plan"]
