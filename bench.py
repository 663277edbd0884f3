#!/usr/bin/env python3
"""bench.py — headline benchmark: batched negacyclic poly-mults/s at n=4096, 60-bit q on N MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A step = one pass of the hot path (tn_poly_mult_dev, fused kernel) over one batch of 65,536
synthetic polynomial pairs per GPU, inputs resident in HBM (BASELINE.json configs[2]; weak
scaling: every rank gets its own 65,536-row block of the same global LCG-seeded batch, no
collective on the data path).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     — algorithmic bytes (3*n*8 B per product: read a, read b, write c)
                 per launch / mean launch duration measured with HIP events on the plan's stream.
  cpu_baseline — the reference's own benchmark binary (oracle/_ref, built from the reference
                 sources in the build container; kind "reference") or this repo's C restatement
                 of it (kind "port"), timed on this box's host cores for a bounded sample.
The result is gated on bit-exactness first: row 0 must reproduce the checksum the reference C++
benchmark prints and sampled rows must equal the on-device O(n^2) direct product (every rank); in the
cpu_baseline leg (N=1) 64 sampled rows are also compared with the CPU oracle.  Any mismatch aborts the run.
Order: one checked pass, a fixed device spin-up (40 untimed launches: the shader clock needs ~0.1 s to settle after
idle), the W warm-up steps, then exactly K timed steps between barriers — so the result does not depend on K or W.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_COEFF = 4096
Q = 1152921504606830593                 # 2^60 - 2^14 + 1   (rtl/ntt_poly_mult.sv:18)
PSI = 431606828070683274                # rtl/ntt_poly_mult.sv:19
ROWS_PER_GPU = 65536                    # BASELINE.json configs[2]
REF_CHECKSUM_ROW0 = 2710933653778106521 # printed by benchmark_ntt_60bit_* for make_poly(1) x make_poly(2)
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PRODUCT = 3 * N_COEFF * 8


def verify(plan, a, b, c, first_global_row):
    """Bit-exactness gate on every rank (never timed), without the CPU oracle: (1) row 0 must reproduce the checksum
    the reference C++ benchmark prints for make_poly(1) x make_poly(2); (2) sampled rows must equal the on-device
    O(n^2) direct negacyclic product (tn_schoolbook_dev: a different algorithm and kernel, benchmark_ntt_60bit.cpp:167)."""
    import torch
    sums = plan.checksum_rows(c[:8])
    if first_global_row == 0 and int(sums[0]) != REF_CHECKSUM_ROW0:
        raise SystemExit(f"PARITY FAILURE: row 0 checksum {int(sums[0])} != reference {REF_CHECKSUM_ROW0}")
    idx = list(range(8)) + list(range(a.shape[0] - 8, a.shape[0]))
    direct = plan.schoolbook(a[idx].contiguous(), b[idx].contiguous())
    plan.synchronize()
    if not torch.equal(direct, c[idx]):
        raise SystemExit("PARITY FAILURE: sampled rows differ from the direct O(n^2) product")
    return len(idx)


def oracle_check(plan, a, b, c):
    """cpu_baseline leg only: 64 sampled rows against the CPU oracle (the checker; never the thing measured)."""
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    P = ctypes.POINTER(ctypes.c_uint64)
    lib.tn_oracle_nwc_poly_mult_batch.argtypes = [P, P, P, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]
    idx = list(range(32)) + list(range(a.shape[0] - 32, a.shape[0]))
    ha = np.ascontiguousarray(plan.to_host(a[idx]).astype(np.uint64))
    hb = np.ascontiguousarray(plan.to_host(b[idx]).astype(np.uint64))
    hc = plan.to_host(c[idx]).astype(np.uint64)
    ref = np.empty_like(ha)
    rc = lib.tn_oracle_nwc_poly_mult_batch(ha.ctypes.data_as(P), hb.ctypes.data_as(P), ref.ctypes.data_as(P), len(idx), N_COEFF, Q, PSI)
    if rc != 0 or not np.array_equal(hc, ref):
        raise SystemExit("PARITY FAILURE: sampled rows differ from the CPU oracle")
    return len(idx)


def cpu_baseline(budget_s=12.0):
    """Reference benchmark on the host cores: single thread, then one process per core (bounded sample)."""
    cores = min(os.cpu_count() or 1, 16)
    ref_dir, port_dir = os.path.join(ROOT, "oracle", "_ref"), os.path.join(ROOT, "oracle", "_build")
    candidates = [(os.path.join(ref_dir, f"benchmark_ntt_60bit_{k}"), "reference", k) for k in ("avx512", "avx2", "scalar")]
    candidates += [(os.path.join(port_dir, f"bench_port{s}"), "port", s.strip("_") or "scalar") for s in ("_avx512", "_avx2", "")]

    def run(exe, reps):
        r = subprocess.run([exe, "--reps", str(reps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(f"{exe} rc={r.returncode}")
        kv = dict(line.split("=", 1) for line in r.stdout.splitlines() if "=" in line and " " not in line)
        if int(kv["checksum"]) != REF_CHECKSUM_ROW0:
            raise RuntimeError("baseline checksum mismatch")
        return float(kv["avg_ns"])

    for exe, kind, simd in candidates:
        if not os.path.exists(exe):
            continue
        try:
            probe = run(exe, 50)                                   # ~0.1 s: calibrate reps
            reps1 = max(100, int(budget_s * 0.4 * 1e9 / (probe * 1.35)))
            single_ns = run(exe, reps1)
            repsN = max(100, int(budget_s * 0.6 * 1e9 / (probe * 1.35)))
            t0 = time.time()
            procs = [subprocess.Popen([exe, "--reps", str(repsN)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
            outs = [p.communicate(timeout=600)[0] for p in procs]
            wall = time.time() - t0
            if any(p.returncode for p in procs):
                raise RuntimeError("parallel baseline run failed")
            per = [float(dict(l.split("=", 1) for l in o.splitlines() if "=" in l and " " not in l)["avg_ns"]) for o in outs]
            allcore = sum(1e9 / ns for ns in per)
            others = {}                                  # the same benchmark's other builds, one thread, ~1 s each
            for k in ("scalar", "avx2", "avx512"):
                e2 = os.path.join(os.path.dirname(exe), os.path.basename(exe).rsplit("_", 1)[0] + "_" + k) if kind == "reference" else None
                if e2 and e2 != exe and os.path.exists(e2):
                    try:
                        others[k] = round(1e9 / run(e2, max(100, int(1e9 / (probe * 1.5)))), 1)
                    except Exception as e:
                        sys.stderr.write(f"[bench] {e2} skipped: {e}\n")
            return {"value": round(allcore, 1), "unit": "poly-mults/s", "cores": cores, "kind": kind,
                    "simd": simd, "single_thread_value": round(1e9 / single_ns, 1), "single_thread_avg_ns": round(single_ns),
                    "single_thread_other_builds": others,
                    "sample": f"{os.path.basename(exe)}: same pair make_poly(1)xmake_poly(2) every rep (reference main loop); "
                              f"1 thread x {reps1} reps, then {cores} processes x {repsN} reps ({wall:.1f} s wall)"}
        except Exception as e:                                     # e.g. SIGILL on a host without AVX-512
            sys.stderr.write(f"[bench] cpu baseline candidate {exe} skipped: {e}\n")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)     # the first launches after idle run below the steady clock
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows per GPU (default: the BASELINE config)")
    ap.add_argument("--variant", default="fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scatter-gather", type=int, default=0, metavar="ROWS",
                    help="N>1 only, off by default: also time the OPTIONAL scatter of a, b from rank 0 and gather of c "
                         "(tiny_ntt_amd.dist, point-to-point over RCCL/xGMI) on ROWS rows per rank; reported separately, never part of value")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tiny_ntt_amd import dist as tdist, engine

    rank, local_rank, world = tdist.env_rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # one process per GPU; BENCH_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the N>1 flow on a 1-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            # the only collectives of this job are the timing barrier and a max-reduce of two floats: if RCCL cannot
            # come up on this node, run that control plane over gloo instead of losing the measurement
            try:
                tdist.init_process_group(backend)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize(dev)
                assert int(probe.item()) == world
            except Exception as e:
                sys.stderr.write(f"[bench] RCCL control plane unavailable ({type(e).__name__}: {str(e)[:200]}); using gloo for barrier/max-reduce\n")
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
                tdist.init_process_group(backend)
        else:
            tdist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    plan = engine.Plan(N_COEFF, Q, PSI, device=dev_index)
    rows = args.rows
    first_row = rank * rows                                    # this rank's block of the global batch
    a = plan.fill_lcg(rows, 2 * first_row + 1, 2)              # global row r: make_poly(2r+1), make_poly(2r+2)
    b = plan.fill_lcg(rows, 2 * first_row + 2, 2)
    c = torch.empty_like(a)
    plan.synchronize()

    # correctness first (one pass, checked), so that nothing idles the device between warm-up and the timed steps
    plan.poly_mult(a, b, variant=args.variant, out=c)
    plan.synchronize()
    checked = verify(plan, a, b, c, first_row)
    # device spin-up (part of setup, like plan creation and data generation): after idle the first ~0.1 s of launches run
    # below the steady shader clock (profiles/: 3.6 ms against 2.7 ms), whatever W the caller asks for
    for _ in range(40):
        plan.poly_mult(a, b, variant=args.variant, out=c)
    for _ in range(max(args.warmup, 0)):                      # the W untimed warm-up steps of the contract
        plan.poly_mult(a, b, variant=args.variant, out=c)
    plan.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        plan.synchronize()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.poly_mult(a, b, variant=args.variant, out=c)       # enqueue on the plan's stream, inputs resident in HBM
    plan.synchronize()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = tdist.max_over_ranks(elapsed, red_dev)

    # dominant kernel: mean launch duration with HIP events on the stream it runs on
    kernel_ms = plan.time_poly_mult(a, b, c, max(args.steps, 5), args.variant)
    kernel_ms = tdist.max_over_ranks(kernel_ms, red_dev)
    achieved = rows * BYTES_PER_PRODUCT / (kernel_ms * 1e-3) / 1e9

    sg = None
    if args.scatter_gather > 0 and world > 1:
        # optional data-movement leg: the compute path above never needs it
        r_sg = min(args.scatter_gather, rows)
        stage = (lambda t: t) if backend == "nccl" else (lambda t: t.cpu())
        back = (lambda t: t) if backend == "nccl" else (lambda t: t.to(dev))
        full_a = stage(plan.fill_lcg(r_sg * world, 1, 2)) if rank == 0 else None
        full_b = stage(plan.fill_lcg(r_sg * world, 2, 2)) if rank == 0 else None
        sdev = dev if backend == "nccl" else torch.device("cpu")
        barrier()
        t1 = time.perf_counter()
        my_a = tdist.scatter_rows(full_a, r_sg * world, N_COEFF, a.dtype, sdev)
        my_b = tdist.scatter_rows(full_b, r_sg * world, N_COEFF, a.dtype, sdev)
        torch.cuda.synchronize(dev); barrier()
        t_sc = tdist.max_over_ranks(time.perf_counter() - t1, red_dev)
        my_c = plan.poly_mult(back(my_a), back(my_b), variant=args.variant)
        plan.synchronize()
        t2 = time.perf_counter()
        full_c = tdist.gather_rows(stage(my_c), r_sg * world, N_COEFF)
        torch.cuda.synchronize(dev); barrier()
        t_ga = tdist.max_over_ranks(time.perf_counter() - t2, red_dev)
        ok = True
        if rank == 0:
            ref_c = plan.poly_mult(back(full_a), back(full_b), variant=args.variant)
            plan.synchronize()
            ok = bool(torch.equal(back(full_c), ref_c))
        row_bytes = N_COEFF * 8
        sg = {"rows_per_rank": r_sg, "scatter_ms": round(t_sc * 1e3, 3), "gather_ms": round(t_ga * 1e3, 3),
              "scatter_GBps": round(2 * r_sg * (world - 1) * row_bytes / t_sc / 1e9, 1),
              "gather_GBps": round(r_sg * (world - 1) * row_bytes / t_ga / 1e9, 1), "gathered_product_bit_exact": ok,
              "transport": "RCCL point-to-point" if backend == "nccl" else "gloo (host staging)"}

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")   # HBM bytes per launch from separate rocprofv3 --pmc passes
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("rows") == rows and tj.get("kernel") == plan.kernel_name(args.variant):
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    total = rows * world * args.steps
    value = total / elapsed
    if rank == 0:
        base, oracle_rows = None, 0
        if not (args.no_cpu_baseline or world > 1):          # the CPU leg: oracle as checker, reference binary as baseline
            oracle_rows = oracle_check(plan, a, b, c)
            base = cpu_baseline()
        line = {
            "metric": "negacyclic poly-mults/sec (n=4096, 60-bit q), bit-exact vs cg_ntt.py",
            "value": round(value, 1),
            "unit": "poly-mults/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"n=4096, q=2^60-2^14+1 (60-bit), batch={rows} per GPU, LCG-seeded rows resident in HBM (BASELINE configs[2])",
                       "n": N_COEFF, "q": Q, "rows_per_gpu": rows, "global_batch": rows * world,
                       "variant": args.variant, "kernel": plan.kernel_name(args.variant), "lazy_reduction": plan.is_lazy,
                       "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "ntts_per_s": round(3 * value, 1),
            "parity": {"row0_checksum": REF_CHECKSUM_ROW0, "rows_compared_with_direct_product_on_device": checked,
                       "rows_compared_with_cpu_oracle": oracle_rows, "bit_exact": True},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": plan.kernel_name(args.variant), "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": rows * BYTES_PER_PRODUCT},
            "cpu_baseline": base,
        }
        if sg is not None:
            line["scatter_gather"] = sg
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
