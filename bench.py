#!/usr/bin/env python3
"""bench.py -- AMG-PCG solve of the 3-D 7-pt Laplacian on MI355X (BASELINE.json metric).

One "step" = one solve phase of the reference's solve loop
(examples/src/C_laplacian/laplacian.c:445-463: ResetInitialGuess + LinearSolverApply),
i.e. BoomerAMG-preconditioned PCG from x0 = 0 to ||r||/||b|| < 1e-6 with the matrix,
right-hand side and hierarchy already resident in HBM.  DOF/s = N / solve-phase time,
exactly the reference's "solve" timer (src/internal/solver.c:668-683); AMG setup is the
reference's separate "prec" timer and is reported beside it (setup_ms), not hidden.

N = 1: BASELINE config 2 (256^3 on one MI355X).  N > 1: the same global problem row
partitioned over N ranks (strong scaling, RCCL halo exchange + fused dot all-reduce).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def spmv_bytes(nrows, ncols, nnz):
    """CSR fp64 + int32, matrix once, x once, y once."""
    return 12.0 * nnz + 4.0 * (nrows + 1) + 8.0 * ncols + 8.0 * nrows


def cpu_baseline(sample_n):
    """Time the CPU oracle (kind 'port') on a bounded sample of the same workload."""
    from oracle import oracle_ffi as o
    # the GPU box shares its host: use the cores this process may run on, at most 16
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", min(avail, 16)))
    os.environ["OMP_NUM_THREADS"] = str(threads)  # read when libgomp initialises (first oracle call)
    A, b = o.lap7(sample_n, sample_n, sample_n)
    t0 = time.perf_counter()
    amg = o.Amg(A, o.amg_params(True))
    t1 = time.perf_counter()
    r = o.pcg(A, b, amg)
    t2 = time.perf_counter()
    n = sample_n ** 3
    return {"value": n / (t2 - t1), "unit": "DOF/s", "cores": threads, "kind": "port",
            "sample": f"lap7 {sample_n}^3 AMG-PCG solve phase (oracle/amg_oracle.c, OpenMP SpMV/Jacobi/dots), "
                      f"{r['iters']} iters in {t2 - t1:.2f} s; serial oracle setup {t1 - t0:.1f} s not counted",
            "iters": r["iters"]}


def run_single(args):
    import hypredrive_amd as h
    n = args.n
    A = h.lap7(n, n, n, want_rhs=False)
    N, _, nnz = A.dims
    h.sync()
    t0 = time.perf_counter()
    amg = h.Amg(A)
    h.sync()
    setup_ms = (time.perf_counter() - t0) * 1e3
    kp = h.KrylovParams.default(False)
    if args.warmup > 0:
        h.solve_device(A, amg, kp, nsolves=args.warmup, profile_k1=False)
    h.sync()
    t0 = time.perf_counter()
    res = h.solve_device(A, amg, kp, nsolves=args.steps, profile_k1=True)
    h.sync()
    t1 = time.perf_counter()
    ms_per_step = (t1 - t0) * 1e3 / args.steps
    iters = res["iters"]
    # algorithmic bytes of one solve: iters PCG iterations + (iters + 1) V-cycles
    bytes_solve = iters * h.pcg_iteration_bytes(A) + (iters + 1) * amg.vcycle_bytes
    k1_bytes = spmv_bytes(N, N, nnz)
    k1_ms = res["k1_avg_ms"]
    achieved = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_spmv_level0_bytes_per_launch")
        except Exception:
            traffic = None
    g, o = amg.complexities
    out = {
        "metric": "DOF/s, AMG-PCG solve phase, 3D 7-pt Laplacian",
        "value": N / (ms_per_step * 1e-3),
        "unit": "DOF/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"lap7 {n}^3 fp64 AMG-PCG (PMIS, ext+i Pmax 4, l1-Jacobi V(1,1), GE coarse), "
                               f"BASELINE config 2", "rows": N, "nnz": nnz, "parallelism": "1 GPU",
                   "rtol": 1e-6, "timed": "solve phase only (reference 'solve' timer); setup_ms is the 'prec' timer"},
        "iters": iters, "true_rel_res": res["true_rel"], "setup_ms": setup_ms,
        "solve_ms_each": [float(x) for x in res["solve_ms"]],
        "num_levels": amg.num_levels, "operator_complexity": o, "grid_complexity": g,
        "solve_phase_hbm_gbs": bytes_solve / (ms_per_step * 1e-3) / 1e9,
        "solve_phase_hbm_frac": bytes_solve / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "dof_iters_per_s": N * iters / (ms_per_step * 1e-3),
        "roofline": {"kernel": "k_spmv_stream<PLAIN,DOT> (level-0 PCG SpMV, LDS-staged, fused <s,p>)", "bound": "hbm",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "bytes_per_launch": k1_bytes, "avg_ms": k1_ms},
    }
    if not args.no_kernel_table:
        kt = {}
        for kind, name in ((0, "spmv"), (1, "l1_jacobi"), (2, "residual"), (3, "vcycle")):
            ms, by = h.time_kernel(kind, A, amg if kind == 3 else None, 20)
            kt[name] = {"ms": ms, "GB/s": by / ms / 1e6, "frac": by / ms / 1e6 / HBM_PEAK_GBS}
        out["kernels"] = kt
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", dest="n", type=int, default=256, help="grid points per dimension (global)")
    ap.add_argument("--cpu-sample", type=int, default=128, help="grid size of the CPU-baseline sample (about 20 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        from hypredrive_amd import dist_bench
        out = dist_bench.run(args)
        if out is None:
            return
    else:
        out = run_single(args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
