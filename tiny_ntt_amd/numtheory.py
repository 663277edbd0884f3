"""Host-side number theory for plan parameters (plan-time only; never on the data path)."""
from __future__ import annotations


def modinv(value: int, modulus: int) -> int:
    """Fermat inverse, q prime — same contract as the reference's modinv (new_reference/cg_ntt.py:9-10)."""
    return pow(value, modulus - 2, modulus)


def bit_reverse(value: int, bits: int) -> int:
    """new_reference/cg_ntt.py:13-18."""
    out = 0
    for _ in range(bits):
        out = (out << 1) | (value & 1)
        value >>= 1
    return out


def sqrt_mod(a: int, p: int):
    """A square root of a modulo the odd prime p (Tonelli-Shanks), or None."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    s, d = 0, p - 1
    while d % 2 == 0:
        d //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, d, p), pow(a, d, p), pow(a, (d + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


def psi_from_omega(omega: int, n: int, q: int) -> int:
    """A primitive 2n-th root psi with psi^2 == omega (needed because plans are keyed on psi).

    cg_ntt(a, omega) (cg_ntt.py:29) takes only omega; for a primitive n-th root omega every
    square root psi satisfies psi^n == omega^(n/2) == -1.
    """
    omega %= q
    if pow(omega, n, q) != 1 or (n > 1 and pow(omega, n // 2, q) != q - 1):
        raise ValueError(f"omega_n={omega} is not a primitive {n}-th root of unity mod {q}")
    r = sqrt_mod(omega, q)
    if r is None:
        raise ValueError(f"omega_n={omega} has no square root mod {q} (2n must divide q-1)")
    return min(r, q - r)


def find_psi(n: int, q: int, max_search: int = 10000, log_fn=None):
    """The reference's parameter finder (scripts/find_psi.py:9-43): the SMALLEST psi in [2, max_search) with
    psi^(2n) == 1 and psi^n == -1 (mod q), or None when that range holds none (e.g. the 60-bit set).  The reference
    prints its progress; pass log_fn=print for the same two summary lines."""
    for psi in range(2, max_search):
        if pow(psi, 2 * n, q) == 1 and pow(psi, n, q) == q - 1:
            if log_fn:
                log_fn(f"\u2713 Found \u03c8 = {psi}")
                log_fn(f"    \u03c9 = \u03c8\u00b2 = {psi * psi % q}")
            return psi
    if log_fn:
        log_fn(f"\u2717 No \u03c8 found in range [2, {max_search})")
    return None


def primitive_2n_root(n: int, q: int) -> int:
    """A primitive 2n-th root of unity mod the prime q for ANY admissible (n, q) (no search bound): the smallest
    g^((q-1)/2n) over g < 2000 that has order exactly 2n.  Not a reference function (its finder is find_psi above,
    which gives up beyond 10^4); used where tests need a psi for arbitrary moduli."""
    if (q - 1) % (2 * n):
        raise ValueError(f"2n={2 * n} does not divide q-1")
    e = (q - 1) // (2 * n)
    best = None
    for g in range(2, 2000):
        c = pow(g, e, q)
        if pow(c, n, q) == q - 1:
            best = c if best is None else min(best, c)
    if best is None:
        raise ValueError("no primitive 2n-th root found")
    return best


def barrett_constants(q: int):
    """(k, mu): k = bitlen(q), mu = floor(2^(2k) / q) — scripts/precompute_constants.py:30-55, the
    constants of rtl/barrett_reduction.v:6-7 (k=23, mu=8396807 for q=8380417)."""
    k = q.bit_length()
    return k, (1 << (2 * k)) // q


def montgomery_constants(q: int):
    """(k, R, R^-1 mod q, q' = -q^-1 mod R) with R = 2^k — scripts/precompute_constants.py:58-111."""
    k = q.bit_length()
    R = 1 << k
    r_inv = pow(R, -1, q)
    q_prime = (-pow(q, -1, R)) % R
    return k, R, r_inv, q_prime


def barrett_reduce(product: int, q: int) -> int:
    """The reference recipe on Python ints (precompute_constants.py:38-46, self-test :145-172), with the
    second conditional subtraction kept."""
    k, mu = barrett_constants(q)
    q1 = product >> (k - 1)
    q2 = (q1 * mu) >> (k + 1)
    r = product - q2 * q
    while r >= q:
        r -= q
    return r
