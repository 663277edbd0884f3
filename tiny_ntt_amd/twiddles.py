"""Twiddle tables in the reference's on-disk format.

The reference keeps psi^k / psi^-k, k = 0..N-1, as `$readmemh` text: one value per line,
uppercase hex, zero-padded to ceil(bits/4) digits (rtl/twiddle_forward_4096_60bit.hex etc.,
written by scripts/generate_twiddles.py:59-77 and scripts/generate_inverse_twiddles.py).
This module writes and reads that format and builds plans from such files.
Plan-time host code only; nothing here is on the data path.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

from . import numtheory


def power_table(root: int, count: int, q: int) -> List[int]:
    """[root^0, root^1, ..., root^(count-1)] mod q (make_power_table, benchmark_ntt_60bit.cpp:43-51)."""
    out, v = [], 1 % q
    for _ in range(count):
        out.append(v)
        v = v * root % q
    return out


def forward_table(n: int, q: int, psi: int) -> List[int]:
    """psi^k, k < n — the contents of rtl/twiddle_forward*.hex."""
    return power_table(psi % q, n, q)


def inverse_table(n: int, q: int, psi: int) -> List[int]:
    """psi^-k, k < n — the contents of rtl/twiddle_inverse*.hex."""
    return power_table(numtheory.modinv(psi % q, q), n, q)


def hex_digits(q: int) -> int:
    return (q.bit_length() + 3) // 4


def format_hex(values: Iterable[int], q: int, uppercase: bool = True) -> str:
    """The reference's files come in both cases (the 60-bit and N=256 tables upper, the 24-bit N=1024/4096 lower)."""
    w = hex_digits(q)
    spec = f"0{w}X" if uppercase else f"0{w}x"
    return "".join(f"{int(v):{spec}}\n" for v in values)


def write_hex(path: str, values: Iterable[int], q: int, uppercase: bool = True) -> None:
    with open(path, "w") as f:
        f.write(format_hex(values, q, uppercase))


def parse_hex(text: str) -> List[int]:
    out = []
    for line in text.splitlines():
        line = line.split("//")[0].strip()
        if line:
            out.append(int(line, 16))
    return out


def read_hex(path: str) -> List[int]:
    with open(path) as f:
        return parse_hex(f.read())


def psi_from_table(table: Sequence[int], q: int) -> int:
    """The root a forward table was generated from (entry 1), after checking every entry is its power
    and that it is a primitive 2n-th root (psi^n == -1)."""
    n = len(table)
    if n < 4 or n & (n - 1):
        raise ValueError(f"Expected a power-of-two number of twiddles, got {n}")
    psi = table[1] % q
    if list(table) != power_table(psi, n, q):
        raise ValueError("table is not the list of powers of its second entry")
    if pow(psi, n, q) != q - 1:
        raise ValueError("table root is not a primitive 2n-th root of unity (psi^n != -1)")
    return psi


def plan_from_hex(forward_hex_path: str, q: int, device: int = 0):
    """Build an engine.Plan whose (n, psi) come from a reference-format forward twiddle file."""
    from . import engine
    table = read_hex(forward_hex_path)
    return engine.get_plan(len(table), q, psi_from_table(table, q), device)
