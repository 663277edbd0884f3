"""bench.py end to end on the GPU box (small workloads): the contract line, the 24-bit configuration, and the N>1 flow
rehearsed with two ranks sharing the one GPU over gloo (the real 8-GPU RCCL run belongs to the driver)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_single_gpu_line_small_workload():
    j = run_bench("--steps", "3", "--warmup", "1", "--rows", "2048", "--no-cpu-baseline", "--latency")
    assert j["n_gpus"] == 1 and j["dtype"] == "u64" and j["unit"] == "poly-mults/s" and j["higher_is_better"] is True
    assert j["control_plane"] == "none" and j["scaling"] == "weak"
    p = j["parity"]
    assert p["bit_exact"] is True and p["first_row_checksum_device"] == p["reference_row0_checksum"] == 2710933653778106521
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["algorithmic_bytes_per_launch"] == 2048 * 98304 and 0 < r["frac"] < 1
    # (kernel_ms is printed with four decimals: at 0.08 ms per launch that alone is 0.06 % of the quotient)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.002 * r["achieved"]
    assert j["config"]["lib_build_id"] not in ("", "unknown")
    assert j["per_rank"]["rows"] == [2048] and set(j["latency"]["device_resident_us"]) == {"1", "16", "256"} and j["latency"]["host_buffer_us"]["1"] > 0
    assert j["strong_scaling_n1_point"] is None          # only with --strong-point


def test_cfg2_line_with_reference_cpu_baseline():
    j = run_bench("--config", "cfg2", "--steps", "5", "--warmup", "2")
    assert j["dtype"] == "u32" and j["config"]["n"] == 1024 and j["config"]["global_batch"] == 4096
    assert j["roofline"]["algorithmic_bytes_per_launch"] == 4096 * 12288
    assert j["parity"]["bit_exact"] is True and j["parity"]["first_row_checksum_device"] == 15308795525113097448
    assert j["parity"]["rows_compared_with_cpu_oracle"] == 64
    b = j["cpu_baseline"]
    assert b is not None and b["value"] > 0 and b["cores"] >= 1 and b["cpu_model"] not in ("", "unknown")
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "benchmark_ntt_1024_avx512")):
        assert b["kind"] == "reference" and "benchmark_ntt_1024" in b["sample"]


def test_two_ranks_without_launcher_strong_split_over_gloo():
    j = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--global-batch", "3000", "--no-cpu-baseline",
                  env={"BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["control_plane"] == "gloo" and j["control_plane_ranks"] == 2
    assert j["config"]["global_batch"] == 3000 and j["config"]["rows_per_gpu"] == 1500
    assert j["parity"]["bit_exact"] is True
    # a straggler would show in the per-rank lists; the N = 1 blocks stay off an N > 1 line
    assert j["per_rank"]["rows"] == [1500, 1500] and len(j["per_rank"]["kernel_ms"]) == 2 and min(j["per_rank"]["kernel_ms"]) > 0
    assert j["latency"] is None and j["strong_scaling_n1_point"] is None and j["spinup_launches"] >= 40          # at least 40 launches and 0.15 s of them (bench.py: SPINUP, SPINUP_SECONDS)
