#!/usr/bin/env python3
"""bench.py — headline benchmark: batched negacyclic poly-mults/s on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg2] [--rows R | --global-batch B]

A step = one pass of the hot path (tn_poly_mult_dev, fused kernel) over this rank's block of synthetic
polynomial pairs, inputs resident in HBM.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json `configs`):
  cfg3 (default)  n=4096, q=2^60-2^14+1, u64 — the configuration the metric is quoted on.
      N=1: 65,536 pairs (configs[2]).  N>1: ONE global batch of 2^20 pairs split in contiguous row blocks,
      rows_g = batch/N (configs[3]) -> "scaling": "strong".  `--rows R` instead gives every
      rank its own R-row block of the global LCG-seeded batch ("weak"); `--global-batch B` picks B.
  cfg2            n=1024, q=8380417 (24-bit), u32, batch 4,096 (configs[1]); cpu_baseline from the reference's
      benchmark_ntt.cpp built for that parameter set (checksum G3 gate).
No collective on the data path: global row r is make_poly(2r+1) x make_poly(2r+2) on whichever rank owns it.

N>1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK/LOCAL_RANK/
WORLD_SIZE/MASTER_* from the environment); called WITHOUT that launcher, `--gpus N` starts the N ranks itself
(fresh child processes, before anything touches the GPU) and exits with their status.

Extra objects on the line:
  roofline      — algorithmic bytes (3*n*w B per product: read a, read b, write c) per launch /
                  mean launch duration measured with HIP events on the stream the kernel runs on.
  cpu_baseline  — the reference's own benchmark binary (oracle/_ref, kind "reference") or this repo's C restatement
                  (kind "port"), timed on this box's host cores for a bounded sample; CPU model and core count stated.
  control_plane — "rccl" | "gloo" | "none": what carried the barrier / max-reduce, and how many ranks it counted.
The result is gated on bit-exactness first (every rank): the first global row must reproduce the checksum the reference
C++ benchmark prints and sampled rows must equal the on-device O(n^2) direct product; in the cpu_baseline leg (N=1)
64 sampled rows are also compared with the CPU oracle.  Any mismatch aborts the run.
Order: one checked pass, a fixed device spin-up (40 untimed launches: the shader clock needs ~0.1 s to settle after
idle), the W warm-up steps, then exactly K timed steps between barriers — so the result does not depend on K or W.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

# the host driver of this pool supports only dmabuf IPC: without this RCCL / cross-process device memory fails with
# hipIpcGetMemHandle: invalid argument (already exported on the build and GPU boxes; set here for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SPINUP = 40                             # untimed launches before the W warm-up steps (setup: the shader clock settles ~0.1 s after idle); on the line
SPINUP_SECONDS = 0.15                   # ... and at least this long: 40 launches of a short workload (cfg2: 23 us each) end long before the clock has
                                        # settled (measured: 23.5-24.5 us per step after 1 ms of launches, 20.4 us after 45 ms, profiles/r3_cfg2_short_launch_ab.txt)

# parameter sets: SURVEY.md §8 table; checksums: what the reference benchmark prints for make_poly(1) x make_poly(2)
CONFIGS = {
    "cfg3": dict(n=4096, q=1152921504606830593, psi=431606828070683274, elem_bytes=8, dtype="u64",          # rtl/ntt_poly_mult.sv:16-24
                 rows=65536, global_batch=1 << 20, checksum_row0=2710933653778106521,
                 ref_bins=[f"benchmark_ntt_60bit_{k}" for k in ("avx512", "avx2", "scalar")], port=True,
                 label="n=4096, q=2^60-2^14+1 (60-bit)", baseline_cfg="BASELINE configs[2]",
                 metric="negacyclic poly-mults/sec (n=4096, 60-bit q), bit-exact vs cg_ntt.py"),
    "cfg2": dict(n=1024, q=8380417, psi=5548360, elem_bytes=4, dtype="u32",                                  # test/Makefile:268,276
                 rows=4096, global_batch=4096, checksum_row0=15308795525113097448,
                 ref_bins=[f"benchmark_ntt_1024_{k}" for k in ("avx512", "scalar")], port=False,
                 label="n=1024, q=8380417 (24-bit)", baseline_cfg="BASELINE configs[1]",
                 metric="negacyclic poly-mults/sec (n=1024, 24-bit q), bit-exact vs cg_ntt.py"),
}


def shard_plan(cfg, world, rank, rows_arg=None, global_batch_arg=None):
    """Which rows of the global LCG-seeded batch this rank owns -> dict(first_row, rows, global_batch, scaling).

    Global row r is always make_poly(2r+1) x make_poly(2r+2), so the union of the rank blocks is
    the global batch exactly once whatever the split.  N=1 default: cfg['rows'] rows.  N>1 default: cfg['global_batch']
    rows split rows_g = batch/N (strong scaling, BASELINE configs[3]).  --rows R: R rows per rank (weak scaling)."""
    from tiny_ntt_amd.dist import shard_rows
    if rows_arg is not None and global_batch_arg is not None:
        raise SystemExit("--rows and --global-batch are mutually exclusive")
    if rows_arg is not None:
        return dict(first_row=rank * rows_arg, rows=rows_arg, global_batch=rows_arg * world, scaling="weak")
    if global_batch_arg is None:
        global_batch_arg = cfg["rows"] if world == 1 else cfg["global_batch"]
    first, count = shard_rows(global_batch_arg, world, rank)
    return dict(first_row=first, rows=count, global_batch=global_batch_arg, scaling="weak" if world == 1 else "strong")


def seeds_for(first_row):
    """(seed0 of a, seed0 of b, stride): row i of the block is global row first_row + i."""
    return 2 * first_row + 1, 2 * first_row + 2, 2


def verify(plan, a, b, c, first_global_row, cfg):
    """Bit-exactness gate on every rank (never timed), without the CPU oracle: (1) global row 0 must reproduce the
    checksum the reference C++ benchmark prints for make_poly(1) x make_poly(2); (2) sampled rows must equal the
    on-device O(n^2) direct negacyclic product (tn_schoolbook_dev: a different algorithm and kernel,
    benchmark_ntt_60bit.cpp:167).  Returns (rows compared, checksum of this rank's first row as computed on device)."""
    import torch
    sums = plan.checksum_rows(c[:8], stream="plan")
    if first_global_row == 0 and int(sums[0]) != cfg["checksum_row0"]:
        raise SystemExit(f"PARITY FAILURE: row 0 checksum {int(sums[0])} != reference {cfg['checksum_row0']}")
    k = min(8, a.shape[0])
    idx = sorted(set(list(range(k)) + list(range(a.shape[0] - k, a.shape[0]))))
    sa, sb, sc = a[idx].contiguous(), b[idx].contiguous(), c[idx].contiguous()   # gathers run on torch's stream ...
    torch.cuda.synchronize(a.device)                                               # ... the checker on the plan's (non-blocking) one
    direct = plan.schoolbook(sa, sb, stream="plan")
    plan.synchronize()
    if not torch.equal(direct, sc):
        raise SystemExit("PARITY FAILURE: sampled rows differ from the direct O(n^2) product")
    return len(idx), int(sums[0])


def oracle_check(plan, a, b, c, cfg):
    """cpu_baseline leg only: 64 sampled rows against the CPU oracle (the checker; never the thing measured)."""
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    P = ctypes.POINTER(ctypes.c_uint64)
    lib.tn_oracle_nwc_poly_mult_batch.argtypes = [P, P, P, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]
    k = min(32, a.shape[0])
    idx = sorted(set(list(range(k)) + list(range(a.shape[0] - k, a.shape[0]))))
    ha = np.ascontiguousarray(plan.to_host(a[idx]).astype(np.uint64))
    hb = np.ascontiguousarray(plan.to_host(b[idx]).astype(np.uint64))
    hc = plan.to_host(c[idx]).astype(np.uint64)
    ref = np.empty_like(ha)
    rc = lib.tn_oracle_nwc_poly_mult_batch(ha.ctypes.data_as(P), hb.ctypes.data_as(P), ref.ctypes.data_as(P), len(idx), cfg["n"], cfg["q"], cfg["psi"])
    if rc != 0 or not np.array_equal(hc, ref):
        raise SystemExit("PARITY FAILURE: sampled rows differ from the CPU oracle")
    return len(idx)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _kv(text):
    return dict(line.split("=", 1) for line in text.splitlines() if "=" in line and " " not in line)


def cpu_baseline(cfg, budget_s=16.0):
    """The reference's benchmark on the host cores: one thread exactly like the reference (same pair every rep),
    then one process per host core — every core the box reports, and the 16 of a one-GPU lease's CPU share — and, with this
    repo's C restatement of the same algorithm, one process per core over DISJOINT rows of the batch the GPU gets."""
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    reported = ncpu
    try:                                             # a lease's CPU share is a cgroup quota, not an affinity mask: running more
        with open("/sys/fs/cgroup/cpu.max") as f:    # processes than that only time-slices them (measured: 256 processes on a 16-CPU
            quota, period = f.read().split()[:2]     # share took 87 s for a lower aggregate than 16 processes)
        if quota != "max":
            ncpu = max(1, min(ncpu, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    ref_dir, port_dir = os.path.join(ROOT, "oracle", "_ref"), os.path.join(ROOT, "oracle", "_build")
    candidates = [(os.path.join(ref_dir, b), "reference", b.rsplit("_", 1)[1]) for b in cfg["ref_bins"]]
    if cfg["port"]:
        candidates += [(os.path.join(port_dir, f"bench_port{s}"), "port", s.strip("_") or "scalar") for s in ("_avx512", "_avx2", "")]

    def run(exe, reps):
        r = subprocess.run([exe, "--reps", str(reps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(f"{exe} rc={r.returncode}")
        kv = _kv(r.stdout)
        if int(kv["checksum"]) != cfg["checksum_row0"]:
            raise RuntimeError("baseline checksum mismatch")
        return float(kv["avg_ns"])

    def parallel(argv_of, procs_n):
        t0 = time.time()
        procs = [subprocess.Popen(argv_of(i), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(procs_n)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        wall = time.time() - t0
        if any(p.returncode for p in procs):
            raise RuntimeError("parallel baseline run failed")
        return sum(1e9 / float(_kv(o)["avg_ns"]) for o in outs), wall, outs

    for exe, kind, simd in candidates:
        if not os.path.exists(exe):
            continue
        try:
            probe = run(exe, 50)                                   # calibrate reps
            share = budget_s / 4.0
            reps1 = max(100, int(share * 1e9 / (probe * 1.35)))
            single_ns = run(exe, reps1)
            repsN = max(100, int(share * 1e9 / (probe * 1.5)))
            counts = sorted({min(16, ncpu), min(ncpu, 128)})      # a one-GPU lease's CPU share, and every core this job may use
            runs = []
            for pn in counts:
                rate, wall, _ = parallel(lambda i: [exe, "--reps", str(repsN)], pn)
                runs.append({"processes": pn, "value": round(rate, 1), "wall_s": round(wall, 1)})
            best = max(runs, key=lambda r: r["value"])
            others = {}                                  # the same benchmark's other builds, one thread, ~1 s each
            for k in ("scalar", "avx2", "avx512"):
                e2 = os.path.join(os.path.dirname(exe), os.path.basename(exe).rsplit("_", 1)[0] + "_" + k) if kind == "reference" else None
                if e2 and e2 != exe and os.path.exists(e2):
                    try:
                        others[k] = round(1e9 / run(e2, max(100, int(1e9 / (probe * 1.5)))), 1)
                    except Exception as e:
                        sys.stderr.write(f"[bench] {e2} skipped: {e}\n")
            disjoint = None
            port = next((os.path.join(port_dir, f"bench_port{s}") for s in ("_avx512", "_avx2", "") if os.path.exists(os.path.join(port_dir, f"bench_port{s}"))), None)
            if cfg["port"] and port:
                try:                                     # disjoint rows of the GPU's batch: 64 rows (6 MiB) per process, rows 64 i ...
                    rows_p, pn = 64, best["processes"]
                    passes = max(1, int(share * 1e9 / (probe * 1.6 * rows_p)))
                    rate, wall, outs = parallel(lambda i: [port, "--rows", str(rows_p), "--first-row", str(rows_p * i), "--reps", str(passes)], pn)
                    disjoint = {"value": round(rate, 1), "processes": pn, "rows_per_process": rows_p, "passes": passes, "wall_s": round(wall, 1),
                                "binary": os.path.basename(port), "kind": "port"}
                except Exception as e:
                    sys.stderr.write(f"[bench] disjoint-rows baseline skipped: {e}\n")
            return {"value": best["value"], "unit": "poly-mults/s", "cores": best["processes"], "host_cpus_reported": reported, "cpu_share_of_this_job": ncpu,
                    "cpu_model": cpu_model(), "kind": kind, "simd": simd,
                    "single_thread_value": round(1e9 / single_ns, 1), "single_thread_avg_ns": round(single_ns),
                    "single_thread_other_builds": others, "all_core_runs": runs, "disjoint_rows_all_core": disjoint,
                    "sample": f"{os.path.basename(exe)}: same pair make_poly(1)xmake_poly(2) every rep (reference main loop); "
                              f"1 thread x {reps1} reps, then {' and '.join(str(r['processes']) for r in runs)} processes x {repsN} reps each; "
                              f"value = the faster of those runs; disjoint_rows_all_core = this repo's C restatement of the same algorithm over "
                              f"disjoint rows of the GPU's batch (SURVEY §8d item 3)"}
        except Exception as e:                                     # e.g. SIGILL on a host without AVX-512
            sys.stderr.write(f"[bench] cpu baseline candidate {exe} skipped: {e}\n")
    return None


def latency_block(plan, a, b, c, variant):
    """The reference's only PUBLISHED metric is single-polynomial latency (reports/final-report.tex:1364-1392: CPU 433-709 us per
    product; :1339-1342: RTL 153-383 us).  Mean time of one launch on small batches, HIP events on the plan's stream over
    back-to-back launches (device-resident), and one whole host-buffer call (H2D + kernel + D2H + sync) for one pair."""
    import numpy as np
    out = {"device_resident_us": {}, "note": "mean over back-to-back launches (tn_time_poly_mult_dev); reference: 686-709 us CPU (60-bit), "
                                             "433-436 us CPU (24-bit), 153-383 us RTL estimate, one polynomial pair"}
    for k in (1, 16, 256):
        if k <= a.shape[0]:
            plan.time_poly_mult(a[:k], b[:k], c[:k], 20, variant)
            out["device_resident_us"][str(k)] = round(plan.time_poly_mult(a[:k], b[:k], c[:k], 200, variant) * 1e3, 2)
    ha, hb = plan.to_host(a[:1]).copy(), plan.to_host(b[:1]).copy()
    hc = np.empty_like(ha)
    for _ in range(5):
        plan.poly_mult(ha, hb, variant=variant, out=hc)
    t0 = time.perf_counter()
    for _ in range(50):
        plan.poly_mult(ha, hb, variant=variant, out=hc)
    out["host_buffer_us"] = {"1": round((time.perf_counter() - t0) / 50 * 1e6, 1)}
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (nothing in this
    process has touched the GPU or imported torch), let rank 0 print the line, return the worst exit status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(abs(p.wait()) for p in procs)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)     # the first launches after idle run below the steady clock
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3")
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU, weak scaling (default: see the module docstring)")
    ap.add_argument("--global-batch", type=int, default=None, help="rows of ONE global batch split over the ranks, strong scaling")
    ap.add_argument("--variant", default="fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # both off by default: they launch the SAME kernel on other batch sizes, and the default command's rocprofv3 --kernel-trace --stats
    # summary must show that kernel's average launch duration for the bench workload alone (profiles/r3_f_kernel_stats.csv)
    ap.add_argument("--latency", action="store_true", help="N=1: add the small-batch latency block (1 / 16 / 256 pairs device-resident, 1 pair host-buffer)")
    ap.add_argument("--strong-point", action="store_true",
                    help="N=1, cfg3: add the N = 1 point of the strong-scaling curve (the whole 2^20-pair batch of BASELINE configs[3] on this GPU)")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="N>1: if the RCCL control plane cannot come up, run the barrier / max-reduce over gloo instead of failing")
    ap.add_argument("--scatter-gather", type=int, default=0, metavar="ROWS",
                    help="N>1 only, off by default: also time the OPTIONAL scatter of a, b from rank 0 and gather of c "
                         "(tiny_ntt_amd.dist, point-to-point over RCCL/xGMI) on ROWS rows per rank; reported separately, never part of value")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from tiny_ntt_amd import dist as tdist, engine

    cfg = CONFIGS[args.config]
    rank, local_rank, world = tdist.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # one process per GPU; BENCH_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the N>1 flow on a 1-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    control_plane, counted = "none", 1
    if world > 1:
        if backend == "nccl":
            # the only collectives of this job are the timing barrier and a max-reduce of two floats
            try:
                tdist.init_process_group(backend)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize(dev)
                counted = int(probe.item())
                if counted != world:
                    raise RuntimeError(f"RCCL all-reduce counted {counted} ranks, expected {world}")
                control_plane = "rccl"
            except Exception as e:
                sys.stderr.write(f"[bench] rank {rank}: RCCL control plane failed: {type(e).__name__}: {str(e)[:400]}\n")
                if not args.allow_gloo:
                    sys.stderr.write("[bench] not falling back silently: pass --allow-gloo (or BENCH_BACKEND=gloo) to run the "
                                     "barrier / max-reduce over gloo\n")
                    raise SystemExit(3)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
        if backend != "nccl":
            tdist.init_process_group(backend)
            probe = torch.ones(1)
            dist.all_reduce(probe)
            counted = int(probe.item())
            control_plane = backend
    red_dev = dev if control_plane == "rccl" else torch.device("cpu")

    plan = engine.Plan(cfg["n"], cfg["q"], cfg["psi"], device=dev_index)
    if plan.elem_bytes != cfg["elem_bytes"]:
        raise SystemExit("unexpected lane width for this configuration")
    sp = shard_plan(cfg, world, rank, args.rows, args.global_batch)
    rows, first_row = sp["rows"], sp["first_row"]
    bytes_per_product = 3 * cfg["n"] * cfg["elem_bytes"]
    modmuls = 3 * (cfg["n"] // 2) * (cfg["n"].bit_length() - 1) + 5 * cfg["n"]
    S = "plan"                                               # every launch, sync and HIP event of this run uses the plan's own stream
    sa, sb, stride = seeds_for(first_row)
    a = plan.fill_lcg(rows, sa, stride, stream=S)            # global row r: make_poly(2r+1), make_poly(2r+2)
    b = plan.fill_lcg(rows, sb, stride, stream=S)
    c = torch.empty_like(a)
    plan.synchronize()

    # correctness first (one pass, checked), so that nothing idles the device between warm-up and the timed steps
    plan.poly_mult(a, b, variant=args.variant, out=c, stream=S)
    plan.synchronize()
    checked, first_sum = verify(plan, a, b, c, first_row, cfg)
    # device spin-up (part of setup, like plan creation and data generation): after idle the first ~0.1 s of launches run
    # below the steady shader clock, whatever W the caller asks for
    spin_launches, t_spin = 0, time.perf_counter()
    while spin_launches < SPINUP or time.perf_counter() - t_spin < SPINUP_SECONDS:
        for _ in range(32):
            plan.poly_mult(a, b, variant=args.variant, out=c, stream=S)
        spin_launches += 32
        plan.synchronize()
    for _ in range(max(args.warmup, 0)):                      # the W untimed warm-up steps of the contract
        plan.poly_mult(a, b, variant=args.variant, out=c, stream=S)
    plan.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        plan.synchronize()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.poly_mult(a, b, variant=args.variant, out=c, stream=S)     # enqueue on the plan's stream, inputs resident in HBM
    plan.synchronize()
    torch.cuda.synchronize(dev)
    elapsed_local = elapsed = time.perf_counter() - t0
    barrier()
    elapsed = tdist.max_over_ranks(elapsed, red_dev)

    # dominant kernel: mean launch duration with HIP events on the plan's stream (the one the timed loop used)
    my_kernel_ms = plan.time_poly_mult(a, b, c, max(args.steps, 5), args.variant)
    kernel_ms = tdist.max_over_ranks(my_kernel_ms, red_dev)
    per_rank = tdist.gather_floats([float(rows), my_kernel_ms, elapsed_local * 1e3 / args.steps], red_dev)   # a straggler shows here
    max_rows = int(tdist.max_over_ranks(rows, red_dev))        # the rank that sets the time owns the largest block
    achieved = max_rows * bytes_per_product / (kernel_ms * 1e-3) / 1e9

    # the timed launches must have left the checked result in place
    _, last_sum = verify(plan, a, b, c, first_row, cfg)
    parity_ok = last_sum == first_sum

    sg = None
    if args.scatter_gather > 0 and world > 1:
        # optional data-movement leg: the compute path above never needs it
        r_sg = min(args.scatter_gather, rows)
        on_dev = control_plane == "rccl"
        stage = (lambda t: t) if on_dev else (lambda t: t.cpu())
        back = (lambda t: t) if on_dev else (lambda t: t.to(dev))
        full_a = stage(plan.fill_lcg(r_sg * world, 1, 2)) if rank == 0 else None
        full_b = stage(plan.fill_lcg(r_sg * world, 2, 2)) if rank == 0 else None
        sdev = dev if on_dev else torch.device("cpu")
        barrier()
        t1 = time.perf_counter()
        my_a = tdist.scatter_rows(full_a, r_sg * world, cfg["n"], a.dtype, sdev)
        my_b = tdist.scatter_rows(full_b, r_sg * world, cfg["n"], a.dtype, sdev)
        torch.cuda.synchronize(dev); barrier()
        t_sc = tdist.max_over_ranks(time.perf_counter() - t1, red_dev)
        my_c = plan.poly_mult(back(my_a), back(my_b), variant=args.variant)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        full_c = tdist.gather_rows(stage(my_c), r_sg * world, cfg["n"])
        torch.cuda.synchronize(dev); barrier()
        t_ga = tdist.max_over_ranks(time.perf_counter() - t2, red_dev)
        ok = True
        if rank == 0:
            ref_c = plan.poly_mult(back(full_a), back(full_b), variant=args.variant)
            torch.cuda.synchronize(dev)
            ok = bool(torch.equal(back(full_c), ref_c))
        row_bytes = cfg["n"] * cfg["elem_bytes"]
        sg = {"rows_per_rank": r_sg, "scatter_ms": round(t_sc * 1e3, 3), "gather_ms": round(t_ga * 1e3, 3),
              "scatter_GBps": round(2 * r_sg * (world - 1) * row_bytes / t_sc / 1e9, 1),
              "gather_GBps": round(r_sg * (world - 1) * row_bytes / t_ga / 1e9, 1), "gathered_product_bit_exact": ok,
              "transport": "RCCL point-to-point" if on_dev else "gloo (host staging)"}

    # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/gpu_pmc.sh): only if taken on THIS library build
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("lib_build_id") != engine.build_id():
                traffic_note = f"profiles/traffic_latest.json belongs to build {tj.get('lib_build_id')}, this library is {engine.build_id()}: not reported"
            elif tj.get("rows") == max_rows and tj.get("kernel") == plan.kernel_name(args.variant) and tj.get("config", "cfg3") == args.config:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    total = sp["global_batch"] * args.steps
    value = total / elapsed
    if rank == 0:
        base, oracle_rows, latency, curve = None, 0, None, None
        if not (args.no_cpu_baseline or world > 1):          # the CPU leg, part 1: the oracle as checker on sampled rows
            oracle_rows = oracle_check(plan, a, b, c, cfg)
        if world == 1 and args.latency:
            latency = latency_block(plan, a, b, c, args.variant)
        if world == 1 and args.config == "cfg3" and args.strong_point:
            # the N = 1 point of the strong-scaling curve (BASELINE configs[3]: ONE batch of 2^20 pairs over 1/2/4/8 GPUs): the whole
            # batch on this GPU (96 GiB of a, b, c), same kernel, a few launches; N > 1 lines divide THIS batch
            try:
                del a, b, c
                torch.cuda.empty_cache()
                gb = cfg["global_batch"]
                A = plan.fill_lcg(gb, 1, 2, stream=S); B = plan.fill_lcg(gb, 2, 2, stream=S); C = torch.empty_like(A)
                plan.time_poly_mult(A, B, C, 2, args.variant)
                ms = plan.time_poly_mult(A, B, C, 5, args.variant)
                ok = int(plan.checksum_rows(C[:1], stream="plan")[0]) == cfg["checksum_row0"]
                curve = {"global_batch": gb, "n_gpus": 1, "value": round(gb / (ms * 1e-3), 1), "ms_per_step": round(ms, 3),
                         "row0_matches_reference": ok, "command": f"python bench.py --gpus 1 --global-batch {gb}  (or: python bench.py --strong-point)"}
                del A, B, C
                torch.cuda.empty_cache()
            except Exception as e:                                 # e.g. a smaller GPU: the point is optional
                curve = {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}
        if not (args.no_cpu_baseline or world > 1):          # the CPU leg, part 2: the reference binary as baseline
            base = cpu_baseline(cfg)
        line = {
            "metric": cfg["metric"],
            "value": round(value, 1),
            "unit": "poly-mults/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": sp["scaling"],
            "vs_baseline": None,
            "dtype": cfg["dtype"],
            "data": "synthetic",
            "config": {"workload": f"{cfg['label']}, global batch {sp['global_batch']} = {rows} rows on this GPU"
                                   f"{'' if world == 1 else ' (contiguous row blocks, rows_g = batch/N)'}, LCG-seeded rows resident in HBM "
                                   f"({cfg['baseline_cfg'] if world == 1 else 'BASELINE configs[3]' if args.config == 'cfg3' and sp['scaling'] == 'strong' else cfg['baseline_cfg'] + ' per GPU'})",
                       "name": args.config, "n": cfg["n"], "q": cfg["q"], "rows_per_gpu": max_rows, "global_batch": sp["global_batch"],
                       "variant": args.variant, "kernel": plan.kernel_name(args.variant), "lazy_reduction": plan.is_lazy,
                       "lib_build_id": engine.build_id(),
                       "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "ntts_per_s": round(3 * value, 1),
            # SURVEY.md §8(d) secondary ceiling: modular multiplications of one product = 3 (n/2) log2 n butterflies + 5 n
            # (two twists, pointwise, n^-1, untwist), reported beside the HBM roofline (the path is integer-ALU work)
            "integer_work": {"modmuls_per_product": modmuls, "modmuls_per_s": round(value * modmuls, 1)},
            "control_plane": control_plane,
            "control_plane_ranks": counted,
            "parity": {"first_row_checksum_device": first_sum, "reference_row0_checksum": cfg["checksum_row0"],
                       "row0_matches_reference": (first_sum == cfg["checksum_row0"]) if first_row == 0 else None,
                       "rows_compared_with_direct_product_on_device": checked,
                       "rows_compared_with_cpu_oracle": oracle_rows,
                       "bit_exact": bool(parity_ok and (first_row != 0 or first_sum == cfg["checksum_row0"]))},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": plan.kernel_name(args.variant), "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_bytes_per_launch": max_rows * bytes_per_product},
            "cpu_baseline": base,
            "latency": latency,
            "strong_scaling_n1_point": curve,
            "per_rank": {"rows": [int(r[0]) for r in per_rank], "kernel_ms": [round(r[1], 5) for r in per_rank],
                         "ms_per_step": [round(r[2], 5) for r in per_rank]},
            "spinup_launches": spin_launches,
        }
        if traffic_note:
            line["roofline"]["traffic_note"] = traffic_note
        if sg is not None:
            line["scatter_gather"] = sg
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
