"""Plan/constants tooling: twiddle tables in the reference's hex format,
Barrett / Montgomery constants.  CPU only."""
import hashlib
import json
import os
import random

import pytest

from conftest import GOLDEN, PARAMS
from tiny_ntt_amd import numtheory, twiddles


def test_hex_export_reproduces_reference_files_byte_for_byte():
    with open(os.path.join(GOLDEN, "reference_hex_digests.json")) as f:
        digests = json.load(f)
    assert len(digests) == 8
    for name, d in digests.items():
        n, q, psi = PARAMS[d["tag"]]
        table = twiddles.forward_table(n, q, psi) if d["kind"] == "fwd" else twiddles.inverse_table(n, q, psi)
        upper = d["first"][1] == d["first"][1].upper() and d["first"][1] != d["first"][1].lower()
        text = twiddles.format_hex(table, q, uppercase=upper)
        assert text.splitlines()[:3] == d["first"] and text.splitlines()[-1] == d["last"], name
        assert len(text.splitlines()) == d["lines"] == n
        assert hashlib.sha256(text.encode()).hexdigest() == d["sha256"], f"{name}: exported table differs from the reference file"


def test_hex_round_trip_and_plan_parameters(tmp_path):
    n, q, psi = PARAMS["P4096_60"]
    path = tmp_path / "fwd.hex"
    twiddles.write_hex(str(path), twiddles.forward_table(n, q, psi), q)
    back = twiddles.read_hex(str(path))
    assert back == twiddles.forward_table(n, q, psi) and len(back[1:2]) == 1
    assert twiddles.psi_from_table(back, q) == psi
    assert twiddles.hex_digits(q) == 15 and twiddles.hex_digits(8380417) == 6
    inv = twiddles.inverse_table(n, q, psi)
    assert all(a * b % q == 1 for a, b in zip(back[:50], inv[:50]))
    with pytest.raises(ValueError, match="powers"):
        twiddles.psi_from_table(back[:2] + [5] + back[3:], q)
    with pytest.raises(ValueError, match="power-of-two"):
        twiddles.psi_from_table(back[:100], q)
    with pytest.raises(ValueError, match="primitive"):
        twiddles.psi_from_table(twiddles.power_table(psi * psi % q, n, q), q)       # omega is only an n-th root
    assert twiddles.parse_hex("0A // comment\n\nff\n") == [10, 255]


def test_barrett_and_montgomery_constants():
    assert numtheory.barrett_constants(8380417) == (23, 8396807)                 # rtl/barrett_reduction.v:6-7
    assert numtheory.barrett_constants(1152921504606830593) == (60, 1152921504606863359)
    for q in (8380417, 1152921504606830593, 7681, 3329):
        k, R, r_inv, q_prime = numtheory.montgomery_constants(q)
        assert R == 1 << k and R * r_inv % q == 1 and (q * q_prime + 1) % R == 0
        rnd = random.Random(q)
        for _ in range(2000):
            a, b = rnd.randrange(q), rnd.randrange(q)
            assert numtheory.barrett_reduce(a * b, q) == a * b % q
        assert numtheory.barrett_reduce((q - 1) * (q - 1), q) == 1


def test_command_line_tool_writes_the_reference_files(tmp_path, capsys):
    """python -m tiny_ntt_amd.twiddles forward|inverse|find-psi|constants: the reference's scripts/ as one tool; the hex files it
    writes hash to the reference's own rtl/twiddle_*.hex (digests committed as a fixture)."""
    with open(os.path.join(GOLDEN, "reference_hex_digests.json")) as f:
        digests = json.load(f)
    checked = 0
    for name, d in digests.items():
        upper = d["first"][1] == d["first"][1].upper() and d["first"][1] != d["first"][1].lower()
        if not upper:
            continue                              # the tool writes uppercase like the scripts; two of the reference's files are lowercase
        n, q, psi = PARAMS[d["tag"]]
        kind = "forward" if d["kind"] == "fwd" else "inverse"
        width = len(d["first"][0]) * 4
        assert twiddles.main([kind, "--n", str(n), "--q", str(q), "--psi", str(psi), "--width", str(width), "--output-dir", str(tmp_path)]) == 0
        text = open(tmp_path / f"twiddle_{kind}.hex").read()
        assert hashlib.sha256(text.encode()).hexdigest() == d["sha256"], name
        checked += 1
    assert checked >= 4
    assert twiddles.main(["forward", "--psi", "5", "--output-dir", str(tmp_path)]) == 1        # not a 2N-th root: refused like the script's assert
    capsys.readouterr()
    assert twiddles.main(["find-psi", "4096", "8380417"]) == 0
    assert "parameter PSI = 687;" in capsys.readouterr().out                                   # scripts/find_psi.py's answer (fixture reference_find_psi.json)
    assert twiddles.main(["find-psi", "4096", "1152921504606830593"]) == 1                     # none below 10^4, like the script
    capsys.readouterr()
    assert twiddles.main(["constants", "--q", "8380417"]) == 0
    out = capsys.readouterr().out
    assert "K = 23" in out and "MU = floor(2^46 / Q) = 8396807" in out                          # rtl/barrett_reduction.v:6-7
