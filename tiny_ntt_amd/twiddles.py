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


# ---- command line: the reference's scripts/ as one tool --------------------------------------
#   python -m tiny_ntt_amd.twiddles forward  [--n N --q Q --psi PSI --width BITS --output-dir DIR]   scripts/generate_twiddles.py:136-200
#   python -m tiny_ntt_amd.twiddles inverse  [...same...]                                             scripts/generate_inverse_twiddles.py:158-
#   python -m tiny_ntt_amd.twiddles find-psi [N Q]                                                    scripts/find_psi.py:64-85
#   python -m tiny_ntt_amd.twiddles constants [--q Q]                                                 scripts/precompute_constants.py:113-
# Defaults are the reference's (N = 256, Q = 8380417, psi = 1239911, 24-bit words; find-psi: N = 4096).  The hex files carry
# the reference's names (twiddle_forward.hex / twiddle_inverse.hex) and its $readmemh format: ceil(width / 4) uppercase digits.
def _write_table(kind: str, args) -> int:
    n, q, psi = args.n, args.q, args.psi
    if pow(psi, 2 * n, q) != 1 or pow(psi, n, q) != q - 1:               # verify_psi_properties (generate_twiddles.py:44-56)
        print(f"psi={psi} is not a primitive {2 * n}-th root of unity mod {q}")
        return 1
    table = forward_table(n, q, psi) if kind == "forward" else inverse_table(n, q, psi)
    digits = (args.width + 3) // 4
    if q.bit_length() > args.width:
        print(f"--width {args.width} is narrower than the modulus ({q.bit_length()} bits)")
        return 1
    import os
    os.makedirs(args.output_dir, exist_ok=True)
    path = os.path.join(args.output_dir, f"twiddle_{kind}.hex")
    with open(path, "w") as f:
        f.write("".join(f"{v:0{digits}X}\n" for v in table))
    print(f"Parameters: N={n}, Q={q}, psi={psi}")
    print(f"  First 5: {table[:5]}")
    print(f"  Last 5:  {table[-5:]}")
    print(f"Generated hex file: {path}")
    print(f"  Entries: {len(table)}")
    print(f"  Width: {args.width} bits ({digits} hex digits)")
    return 0


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(prog="python -m tiny_ntt_amd.twiddles", description=__doc__.splitlines()[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    for kind in ("forward", "inverse"):
        g = sub.add_parser(kind, help=f"write twiddle_{kind}.hex")
        g.add_argument("--n", type=int, default=256)
        g.add_argument("--q", type=int, default=8380417)
        g.add_argument("--psi", type=int, default=1239911)
        g.add_argument("--width", type=int, default=24)
        g.add_argument("--output-dir", default=".")
    f = sub.add_parser("find-psi", help="smallest psi in [2, 10000) with psi^N = -1 (mod Q)")
    f.add_argument("n", type=int, nargs="?", default=4096)
    f.add_argument("q", type=int, nargs="?", default=8380417)
    c = sub.add_parser("constants", help="Barrett / Montgomery constants of a modulus")
    c.add_argument("--q", type=int, default=8380417)
    args = ap.parse_args(argv)
    if args.cmd in ("forward", "inverse"):
        return _write_table(args.cmd, args)
    if args.cmd == "find-psi":
        psi = numtheory.find_psi(args.n, args.q, log_fn=print)
        if psi is None:
            return 1
        print(f"parameter PSI = {psi};")
        return 0
    k, mu = numtheory.barrett_constants(args.q)
    mk, R, r_inv, q_prime = numtheory.montgomery_constants(args.q)
    print(f"Q = {args.q}")
    print(f"Barrett:    K = {k}  MU = floor(2^{2 * k} / Q) = {mu}")
    print(f"Montgomery: K = {mk}  R = 2^{mk} = {R}  R_INV = {r_inv}  Q_PRIME = {q_prime}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
