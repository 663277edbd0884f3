"""bench.py's host-side logic that needs no GPU: how the global LCG-seeded batch is split over ranks (BASELINE.json
configs[3]), the per-rank seeds, the configuration table, and that `--gpus N` without a launcher really
starts N ranks."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_default_split_covers_the_global_batch_exactly_once(world):
    cfg = bench.CONFIGS["cfg3"]
    blocks = [bench.shard_plan(cfg, world, r) for r in range(world)]
    expect_global = 65536 if world == 1 else 1 << 20           # configs[2] at N=1, configs[3] (one 2^20-row batch) at N>1
    assert all(b["global_batch"] == expect_global for b in blocks)
    assert all(b["scaling"] == ("weak" if world == 1 else "strong") for b in blocks)
    # contiguous, disjoint, complete; rows_g = batch / G
    pos = 0
    for b in blocks:
        assert b["first_row"] == pos and b["rows"] == expect_global // world
        pos += b["rows"]
    assert pos == expect_global
    # seeds: local row i of a block is global row first_row + i = make_poly(2r+1) x make_poly(2r+2)
    seen = set()
    for b in blocks:
        sa, sb, stride = bench.seeds_for(b["first_row"])
        assert stride == 2 and sb == sa + 1
        first, last = sa, sa + stride * (b["rows"] - 1)
        assert first == 2 * b["first_row"] + 1 and last == 2 * (b["first_row"] + b["rows"] - 1) + 1
        assert not (seen & {first, last})
        seen |= {first, last}
    assert min(seen) == 1 and max(seen) == 2 * (expect_global - 1) + 1


def test_rows_and_global_batch_overrides():
    cfg = bench.CONFIGS["cfg3"]
    for world in (1, 2, 8):
        for r in range(world):
            w = bench.shard_plan(cfg, world, r, rows_arg=1000)
            assert (w["first_row"], w["rows"], w["global_batch"], w["scaling"]) == (r * 1000, 1000, 1000 * world, "weak")
    # uneven split: block sizes differ by at most one and still tile the batch
    blocks = [bench.shard_plan(cfg, 8, r, global_batch_arg=1003) for r in range(8)]
    assert sum(b["rows"] for b in blocks) == 1003 and max(b["rows"] for b in blocks) - min(b["rows"] for b in blocks) == 1
    assert [b["first_row"] for b in blocks] == [sum(x["rows"] for x in blocks[:i]) for i in range(8)]
    with pytest.raises(SystemExit):
        bench.shard_plan(cfg, 2, 0, rows_arg=10, global_batch_arg=20)


def test_config_table_matches_the_reference_parameter_sets():
    c3, c2 = bench.CONFIGS["cfg3"], bench.CONFIGS["cfg2"]
    assert pow(c3["psi"], c3["n"], c3["q"]) == c3["q"] - 1 and c3["q"] == 2 ** 60 - 2 ** 14 + 1      # rtl/ntt_poly_mult.sv:16-24
    assert pow(c2["psi"], c2["n"], c2["q"]) == c2["q"] - 1 and (c2["n"], c2["q"]) == (1024, 8380417)  # test/Makefile:268
    assert 3 * c3["n"] * c3["elem_bytes"] == 98304 and 3 * c2["n"] * c2["elem_bytes"] == 12288
    from conftest import REF_CHECKSUMS as REFERENCE_CHECKSUMS
    assert c3["checksum_row0"] == REFERENCE_CHECKSUMS["P4096_60"][1] and c2["checksum_row0"] == REFERENCE_CHECKSUMS["P1024"][1]


def test_gpus_flag_without_a_launcher_starts_that_many_ranks():
    """`python bench.py --gpus 2` used to run ONE rank and report n_gpus=1.  Without a GPU every rank stops at
    the device check, so here the observable is: two ranks ran (both complain) and the exit status is non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = "-1"          # also on a GPU box: no device for the children
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a HIP device") == 2, r.stderr
    assert r.stdout.strip() == ""
