#!/usr/bin/env python3
"""bench.py -- AMG-PCG solve of the 3-D 7-pt Laplacian on MI355X (BASELINE.json metric).

One "step" = one pass of the reference's solve loop (examples/src/C_laplacian/laplacian.c:445-463:
HYPREDRV_LinearSystemResetInitialGuess + HYPREDRV_LinearSolverApply), i.e. BoomerAMG-preconditioned
PCG from x0 = 0 to ||r||/||b|| < 1e-6 with the matrix, right-hand side and hierarchy already resident
in HBM.  Every N runs the SAME code path: the HYPREDRV_* C API (include/HYPREDRV.h), one process per
GPU.  AMG setup is the reference's separate "prec" timer, reported beside the metric (setup_ms).

N = 1: BASELINE config 2 (256^3 on one MI355X).  N > 1: one 256^3 block per GPU, row partitioned
(weak scaling; 8 GPUs = BASELINE config 3, 512^3), RCCL halo exchange + dot all-reduce.  --strong
keeps the global problem at --grid^3 instead.

Launch: under torchrun (WORLD_SIZE set) this process is one rank.  `python bench.py --gpus N` with
N > 1 and no WORLD_SIZE starts the N ranks itself as CHILD processes (reference:
scripts/node_scaling.sh:1275-1292 `mpirun -np N`), before anything here touches the GPU.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# RCCL / device-memory sharing across processes needs dmabuf IPC on this driver: must be in the environment before the first HIP call
# of the process, whoever launched it (the driver's torchrun, bench.py's own spawn, a child bench)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# The headline is measured first and published at once (stderr + gpurun_out/bench_headline.json; printed to stdout by the SIGTERM handler
# if the job is ended during the extras); the extras that follow share ONE time budget and are skipped, by name, once it is spent.
STATE = {"headline": None, "printed": False, "deadline": None}


def publish_headline(out):
    STATE["headline"] = out
    STATE["deadline"] = time.time() + float(os.environ.get("HDA_BENCH_EXTRAS_BUDGET", "600"))
    line = json.dumps(out)
    print("[bench] headline (extras follow): " + line, file=sys.stderr, flush=True)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_headline.json"), "w") as f:
            f.write(line + "\n")
    except OSError:
        pass


def budget_left():
    return 1e9 if STATE["deadline"] is None else STATE["deadline"] - time.time()


def print_line(out):
    if not STATE["printed"]:
        STATE["printed"] = True
        print(json.dumps(out), flush=True)


def _on_term(signum, frame):  # ended from outside during the extras: the measured headline must not be lost
    if STATE["headline"] is not None and not STATE["printed"]:
        STATE["headline"]["extras_interrupted"] = f"signal {signum} during the extras"
        print_line(STATE["headline"])
    os._exit(143)

YAML = "solver: pcg\npreconditioner:\n  preset: poisson\n"
# BASELINE config 5 (--workload aniso): GMRES(30) + BoomerAMG with the ILU(0) complex smoother on level 0, Jacobi-iterative
# triangular solves (reference examples/ex8.yml variant 4 block; src/internal/amg.c:899-921, ilu.c:21-23)
YAML_ANISO = ("solver:\n  gmres:\n    relative_tol: 1.0e-6\npreconditioner:\n  amg:\n    smoother:\n      type: ilu\n      num_levels: 1\n"
              "      ilu:\n        type: bj-iluk\n        tri_solve: 0\n")


def spmv_bytes(nrows, ncols, nnz):
    """CSR fp64 + int32, matrix once, x once, y once."""
    return 12.0 * nnz + 4.0 * (nrows + 1) + 8.0 * ncols + 8.0 * nrows


def gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0


def find_hypre():
    """BASELINE.md's second CPU number needs a hypre installation on the box: look where one would be (no network,
    nothing is installed by this script).  Returns the include directory or None."""
    import glob
    roots = [os.environ.get(k) for k in ("HYPRE_ROOT", "HYPRE_DIR", "HYPRE_HOME")]
    roots += ["/usr", "/usr/local", "/opt/hypre", "/opt/rocm", "/opt/conda", os.path.expanduser("~/.local")]
    roots += glob.glob("/opt/*hypre*") + glob.glob("/opt/spack/opt/spack/*/*/hypre-*")
    for r in roots:
        if r and os.path.exists(os.path.join(r, "include", "HYPRE.h")):
            return os.path.join(r, "include")
    return None


def cpu_baseline(sample_n, gpu_iters=None):
    """Time the CPU oracle (kind 'port') on the benchmark's own configuration (or a smaller sample with --cpu-sample)."""
    from oracle import oracle_ffi as o
    # the GPU box shares its host: use the cores this process may run on, at most 16
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", min(avail, 16)))
    os.environ["OMP_NUM_THREADS"] = str(threads)  # read when libgomp initialises (first oracle call)
    os.environ.setdefault("OMP_WAIT_POLICY", "active")  # the baseline gets spinning barriers (the test default is passive)
    A, b = o.lap7(sample_n, sample_n, sample_n)
    t0 = time.perf_counter()
    amg = o.Amg(A, o.amg_params(True))
    t1 = time.perf_counter()
    times, r = [], None
    reps = 3 if sample_n >= 200 else 5
    for _ in range(reps):  # the solve phase is short next to the setup: repeat it
        ts = time.perf_counter()
        r = o.pcg(A, b, amg)
        times.append(time.perf_counter() - ts)
    n = sample_n ** 3
    med = sorted(times)[len(times) // 2]
    out = {"value": n / med, "unit": "DOF/s", "cores": threads, "kind": "port",
           "sample": f"lap7 {sample_n}^3 AMG-PCG solve phase (oracle/amg_oracle.c, OpenMP SpMV/Jacobi/dots on {threads} threads), "
                     f"{r['iters']} iters, median of {reps} solves {med:.3f} s (min {min(times):.3f}); oracle setup "
                     f"{t1 - t0:.1f} s reported separately (setup_s), not counted in value",
           "iters": r["iters"], "solve_s": med, "setup_s": t1 - t0, "grid": sample_n,
           # a hypre install would allow the reference's own CPU path as a second baseline; none has been found on these boxes
           "hypre_on_box": find_hypre()}
    if gpu_iters is not None:
        out["iters_match"] = (int(gpu_iters) == int(r["iters"]))
    return out


# ------------------------------------------------------------------------------------------- launch

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (never re-exec: this
    process has not touched the GPU and never will), forward rank 0's JSON line, return the children's status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line, errs = None, []
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.startswith("{") and '"error"' in ln and '"stage"' in ln:
            errs.append(ln)
        else:
            print(ln, file=sys.stderr)
    if errs:
        print(errs[0], flush=True)  # one line: the first rank that said where joining RCCL failed
    if line:
        print(line, flush=True)     # (after an RCCL failure: the host-staged measurement, labelled by its `rccl_error`)
    return r.returncode if (r.returncode or line) else 1


# ------------------------------------------------------------------------------------------- one rank

PHASE = {"name": "start"}  # where a multi-rank run is: what the watchdog of main() reports when the run never finishes


def error_line(stage, rank, msg, nccl_debug=None):
    """A first multi-GPU run that cannot join RCCL says where in ONE JSON line on stdout (stage: uid | init | halo_comm | selftest |
    devices | peer | fallback | timeout:<stage>), with what RCCL itself logged on that rank (NCCL_DEBUG=WARN, hypredrive_amd/dist.py),
    then the process exits non-zero."""
    rec = {"error": str(msg)[:600], "stage": stage, "rank": rank}
    if nccl_debug:
        rec["nccl_debug"] = nccl_debug[-1500:]
    sys.stdout.write(json.dumps(rec) + "\n")  # one write: ranks share the pipe
    sys.stdout.flush()


class World:
    """torch.distributed (gloo) only carries the launcher's side: barriers and the reductions of the report."""

    def __init__(self):
        import threading
        from hypredrive_amd import dist as hdist
        self.hdist = hdist
        rank0 = int(os.environ.get("RANK", "0"))
        limit = float(os.environ.get("HDA_BENCH_INIT_TIMEOUT", "300"))

        def watchdog():  # a collective that never returns (a rank missing, a hung ncclCommInitRank) must not eat the driver's whole slot
            error_line("timeout:" + str(hdist._keep.get("stage", "rendezvous")), rank0, f"joining the ranks did not finish within {limit:.0f} s",
                       hdist.rccl_debug_tail())
            os._exit(3)

        t = threading.Timer(limit, watchdog)
        t.daemon = True
        t.start()
        self.rccl_error = None
        try:
            # strict: N ranks with N GPUs on anything but RCCL is an error, not a measurement (no SILENT staged fallback)
            self.rank, self.size = hdist.init(strict=not os.environ.get("HDA_TRANSPORT"))
        except hdist.TransportError as e:
            # every rank gets here together (dist.init agrees on the failure before raising).  The error line comes first; then, unless
            # HDA_BENCH_NO_FALLBACK is set, the same job is measured over the host-staged transport so that the run still says whether
            # the partitioned solve works across these GPUs -- labelled (`transport`, `rccl_error`), and the process still exits non-zero.
            error_line(e.stage, e.rank, e, getattr(e, "nccl_debug", None))
            if os.environ.get("HDA_BENCH_NO_FALLBACK"):
                raise SystemExit(2)
            self.rccl_error = {"stage": e.stage, "rank": e.rank, "error": str(e)[:400], "nccl_debug": (getattr(e, "nccl_debug", None) or "")[-600:]}
            try:
                from hypredrive_amd import hypredrv as hd
                hd.lib().HYPREDRV_AMD_CommFinalize()
                os.environ.pop("HDA_TRANSPORT", None)  # (a forced transport would be tried again)
                self.rank, self.size = hdist.init("staged")
            except Exception as e2:  # noqa: BLE001
                error_line("fallback", rank0, e2)
                raise SystemExit(2)
        finally:
            t.cancel()
        self.dist = None
        if self.size > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def reduce(self, values, op="sum"):
        if not self.dist:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64)
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN}[op])
        return t.tolist()


def run(args):
    import hypredrive_amd as hh
    from hypredrive_amd import hypredrv as hd
    w = World()
    n = args.n
    P = w.hdist.factor3(w.size)
    weak = not args.strong
    gn = (n * P[0], n * P[1], n * P[2]) if weak else (n, n, n)
    N = gn[0] * gn[1] * gn[2]
    ndev = hh.device_count()
    transport = hh._lib.comm_name()
    if w.size > 1 and ndev >= w.size and transport != "rccl" and not os.environ.get("HDA_TRANSPORT"):
        # one GPU per rank is there: a host-staged transport would be a silent fallback, not a measurement
        error_line("fallback", w.rank, f"{w.size} ranks with {ndev} visible GPUs must run on RCCL, transport is '{transport}'")
        raise SystemExit(2)
    aniso = args.workload == "aniso"
    if aniso:
        # a heterogeneous anisotropic reservoir operator (hypredrive_amd/synthetic.py; SPE10 itself is unreachable offline), the
        # global n^3 system cut into contiguous row blocks (z slabs): every rank hands over its rows as CSR arrays
        from hypredrive_amd.synthetic import spe10_like
        weak, P, gn, N = False, (1, 1, w.size), (n, n, n), n ** 3
        ip, ix, v, b = spe10_like(n)
        lo, hi = w.rank * N // w.size, (w.rank + 1) * N // w.size
        h = hd.Hypredrv(YAML_ANISO)
        h.set_matrix_csr(lo, hi - 1, ip[lo:hi + 1] - ip[lo], ix[ip[lo]:ip[hi]], v[ip[lo]:ip[hi]])
        h.set_rhs_array(lo, hi - 1, b[lo:hi])
        h.finish_system()
        del ip, ix, v, b
    else:
        h = hd.Hypredrv(YAML)
        h.set_laplacian7(gn, P)
    # the reference's protocol is one warm-up run, then the timed ones (scripts/node_scaling.sh): the first setup of a
    # process also pays for device-memory allocation (bimodal, 0.03-0.9 s on these boxes), the second one runs out of the
    # library's caching allocator and is the "prec" timer proper
    PHASE["name"] = "setup"
    setup, setup_alloc = [], []
    for _ in range(2):
        hh.sync()
        w.barrier()
        hh._lib.memory_driver_stats(reset=True)
        t0 = time.perf_counter()
        h.create_and_setup()
        hh.sync()
        w.barrier()
        setup.append((time.perf_counter() - t0) * 1e3)
        setup_alloc.append(hh._lib.memory_driver_stats())  # what of it the driver's hipMalloc took (this rank)
        if len(setup) == 1:
            h.destroy_solver()
    setup = w.reduce(setup, "max")
    A, amg = hh._lib.borrow(h)  # the objects HYPREDRV_LinearSolverSetup built, for the kernel-level measurement entries
    # First contact with an asynchronous transport (RCCL): the same W + K solves are run FIRST with every halo exchange ahead of its
    # product on the library stream (hda_set_overlap(0): one communicator in use at a time, nothing concurrent), and only then as
    # designed -- exchanges on the communication stream under the owned-column part of the product.  The headline is the overlapped
    # form; `serial_exchange` carries the other.  If the overlapped phase raises or does not finish, the line is written from the serial
    # measurement and says so (`overlap_error`): a complete measurement of this job on RCCL either way.
    serial, guard = None, None
    two_phase = w.size > 1 and (transport == "rccl" or os.environ.get("HDA_BENCH_SERIAL_FIRST"))
    if two_phase:
        PHASE["name"] = "serial-exchange rehearsal"
        hh.load().hda_set_overlap(0)
        for _ in range(args.warmup):
            h.apply()
        w.barrier()
        hh.sync()
        t0 = time.perf_counter()
        st, lst = [], None
        for _ in range(args.steps):
            lst = h.apply()
            st.append(lst["solve_s"] * 1e3)
        hh.sync()
        w.barrier()
        dts = w.reduce([time.perf_counter() - t0], "max")[0]
        serial = {"what": "the same W + K solves with every halo exchange ahead of its product on the library stream (hda_set_overlap(0)), run "
                          "before the overlapped ones", "ms_per_step": dts * 1e3 / args.steps, "value": N / (dts / args.steps),
                  "iters": lst["iters"], "converged": lst["converged"], "final_rel": lst["final_rel"],
                  "solve_timer_ms": w.reduce([sorted(st)[len(st) // 2]], "max")[0]}
        hh.load().hda_set_overlap(-1)
        import threading
        limit = float(os.environ.get("HDA_BENCH_OVERLAP_TIMEOUT", str(max(60.0, 50.0 * dts))))

        def give_up(why):
            if w.rank == 0:
                line = {"metric": "DOF/s, AMG-PCG solve phase, 3D 7-pt Laplacian", "value": serial["value"], "unit": "DOF/s", "n_gpus": w.size,
                        "steps": args.steps, "warmup": args.warmup, "ms_per_step": serial["ms_per_step"], "higher_is_better": True,
                        "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                        "config": {"workload": f"lap7 {gn[0]}x{gn[1]}x{gn[2]} fp64 AMG-PCG (PMIS, ext+i Pmax 4, l1-Jacobi V(1,1), GE coarse), "
                                               + (f"{n}^3 per rank" if weak else "fixed size"), "rows": N, "parallelism": f"row blocks {P[0]}x{P[1]}x{P[2]}, one process per GPU",
                                   "exchanges": "ahead of their products on the library stream (HDA_OVERLAP=0): the overlapped form failed, see overlap_error"},
                        "transport": transport, "iters": serial["iters"], "converged": serial["converged"], "final_rel": serial["final_rel"],
                        "solve_timer_ms": serial["solve_timer_ms"], "setup_ms": setup[1], "setup_cold_ms": setup[0],
                        "overlap_error": why, "serial_exchange": serial}
                sys.stdout.write(json.dumps(line) + "\n")
                sys.stdout.flush()
            # every rank: kernels of the overlapped phase may never finish.  The line above is a complete measurement of the job with serial
            # exchanges -- and the designed (overlapped) path FAILED: "a result line, and a failure", as for the staged fallback (exit 2)
            os._exit(2)

        guard = threading.Timer(limit, give_up, args=(f"the overlapped phase did not finish within {limit:.0f} s (phase: see stderr)",))
        guard.daemon = True
        guard.start()
        if os.environ.get("HDA_BENCH_TEST_OVERLAP_HANG"):  # (test hook: the overlapped phase never returns)
            time.sleep(1e6)
    PHASE["name"] = "warmup"
    try:
        for _ in range(args.warmup):
            h.apply()
    except Exception as e:  # noqa: BLE001 - an RCCL error inside the overlapped products: the serial measurement stands
        if serial is None:
            raise
        give_up(f"the overlapped phase raised: {e!r}"[:600])
    # probes: HIP events on the library stream around every launch of four kernels inside the timed solves.
    # dominant = Jacobi sweep on the biggest operator of the hierarchy kept in plain CSR (level 1 for this workload:
    # level 0 is stencil-coded and cheaper); k1 = the level-0 PCG product; P0 / R0 = level-0 prolongation / restriction
    lv_nnz = [amg.level_matrix(l, 0).dims[2] for l in range(amg.num_levels - 1)]
    fb0 = hh.format_bytes(A, amg)
    dom = max(range(len(lv_nnz)), key=lambda l: (0 if (l == 0 and fb0["coded"]) else lv_nnz[l])) if lv_nnz else 0
    Ad = amg.level_matrix(dom, 0)
    hh.probe_spmv(None, 0)
    dom_mode = 2
    if aniso and amg.num_levels > 1:
        # with ILU(0) on level 0 the largest l1-Jacobi sweep is the level-1 one; the level-0 operator is applied in the smoother's
        # residuals and GMRES' products (k1)
        dom = max(range(1, len(lv_nnz)), key=lambda l: lv_nnz[l]) if len(lv_nnz) > 1 else 0
        Ad = amg.level_matrix(dom, 0)
    probes = {"dom": hh._lib.probe_add(Ad, dom_mode), "k1": hh._lib.probe_add(A, 0)}
    if aniso:
        probes["res0"] = hh._lib.probe_add(A, 1)
    if amg.num_levels > 1:
        P0, R0 = amg.level_matrix(0, 1), amg.level_matrix(0, 2)
        probes["P0"] = hh._lib.probe_add(P0, 0)
        probes["R0"] = hh._lib.probe_add(R0, 0)
    hh._lib.comm_stats(reset=True)
    PHASE["name"] = "timed solves"
    w.barrier()
    hh.sync()
    hh.load().hda_marker(1)  # (an empty kernel: the solve phase's boundary in traces and counter passes, tools/pmc_traffic.py)
    t0 = time.perf_counter()
    last, solve_timer, vcyc = None, [], 0
    try:
        for _ in range(args.steps):
            last = h.apply()
            solve_timer.append(last["solve_s"] * 1e3)
            vcyc = hh.load().hda_last_precond_calls()
    except Exception as e:  # noqa: BLE001
        if serial is None:
            raise
        give_up(f"the overlapped phase raised: {e!r}"[:600])
    hh.load().hda_marker(2)
    hh.sync()
    w.barrier()
    dt = w.reduce([time.perf_counter() - t0], "max")[0]
    if guard is not None:
        guard.cancel()
    PHASE["name"] = "report"
    ms_per_step = dt * 1e3 / args.steps
    cs = hh._lib.comm_stats()
    pr = {k: hh._lib.probe_read_id(v) for k, v in probes.items()}
    hh.probe_spmv(None, 0)
    iters = last["iters"]
    # bytes of one solve on this rank: iters PCG iterations + the V-cycles really run (hypre's PCG runs iters + 1, the
    # last one unused; here it is skipped).  "csr" = SURVEY 8(d) figures, "fmt" = what the kernels stream (coded operators)
    (it_csr, it_fmt), (vc_csr, vc_fmt) = h.solve_phase_bytes()
    by = w.reduce([iters * it_csr + vcyc * vc_csr, iters * it_fmt + vcyc * vc_fmt, 1.0], "sum")
    ranks_seen = int(round(by[2]))
    timer_ms = w.reduce([sorted(solve_timer)[len(solve_timer) // 2]], "max")[0]
    out = None
    if w.rank == 0:
        dn, dc, dnnz = Ad.dims
        fbd = hh.format_bytes(Ad)
        dom_ms, dom_count = pr["dom"]
        dom_bytes = spmv_bytes(dn, dc, dnnz) + 16.0 * dn  # + b, dinv of the sweep
        a_n, a_c, a_nnz = A.dims
        k1_ms, k1_count = pr["k1"]
        k1_bytes = spmv_bytes(a_n, a_c, a_nnz)  # the second operand of the fused dot <Ap, p> is x itself: no further stream
        k1_fmt = fb0["spmv"]
        g, o = amg.complexities
        traffic, traffic_src = {}, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
                traffic_src = f"profiles/traffic.json ({traffic.get('round', 'committed rocprofv3 --pmc run')}), not measured in this run"
            except Exception:
                traffic = {}
        per_it = 1.0 / max(iters * args.steps, 1)
        out = {
            "metric": "DOF/s, GMRES+AMG(ILU0 smoother) solve phase, anisotropic heterogeneous diffusion" if aniso
                      else "DOF/s, AMG-PCG solve phase, 3D 7-pt Laplacian",
            "value": N / (ms_per_step * 1e-3), "unit": "DOF/s",
            "n_gpus": w.size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"spe10-like {n}^3 fp64 (7-pt finite volumes, log-normal permeability over 3 decades, k_v/k_h 1e-3; "
                                    "hypredrive_amd/synthetic.py), GMRES(30) + BoomerAMG (PMIS, ext+i Pmax 4, V(1,1): ILU(0) smoother on level 0 "
                                    "with Jacobi-iterative triangular solves, l1-Jacobi below), BASELINE config 5 stand-in") if aniso else
                                   f"lap7 {gn[0]}x{gn[1]}x{gn[2]} fp64 AMG-PCG (PMIS, ext+i Pmax 4, l1-Jacobi V(1,1), GE coarse), "
                                   + ("BASELINE config 2" if (w.size == 1 and n == 256) else
                                      "BASELINE config 3" if (w.size == 8 and weak and n == 256) else f"{n}^3 per rank" if weak else "fixed size"),
                       "rows": N, "rows_per_rank": a_n, "parallelism": f"row blocks {P[0]}x{P[1]}x{P[2]}, one process per GPU",
                       "rtol": 1e-6, "api": "HYPREDRV_LinearSystemResetInitialGuess + HYPREDRV_LinearSolverApply per step (every N)",
                       "timed": "K steps between barrier + device sync; includes the reference's untimed r0 / final-residual evaluations "
                                "(solve_timer_ms is the reference's 'solve' timer alone); setup_ms is the 'prec' timer"},
            "ranks_seen": ranks_seen, "transport": transport, "rccl_error": w.rccl_error, "serial_exchange": serial,
            # levels of the hierarchy cut into row blocks; the rest is the replicated tail every rank cycles redundantly (a level goes
            # there once it has fewer than HDA_REPLICATE_ROWS_PER_RANK = 50 000 rows per rank, at least 100 000 in all)
            "partitioned_levels": hh.load().hda_amd_partitioned_levels(h.h), "levels_total": hh.load().hda_amd_hierarchy_levels(h.h),
            # rank-to-rank operations rank 0 issued per PCG iteration (V-cycle included): all-reduces (<s,p>; <r,r> + <r,z> fused into one
            # two-double reduction; restricted residual of the replicated tail) and neighbour halo exchanges
            "allreduces_per_iter": cs["allreduce"] * per_it, "halo_exchanges_per_iter": cs["exchange"] * per_it,
            "collectives_per_iter": (cs["allreduce"] + cs["exchange"]) * per_it,
            "halo_exchanges_overlapped_per_iter": cs["overlapped"] * per_it,
            "allreduce_doubles_per_iter": cs["allreduce_doubles"] * per_it, "halo_doubles_per_iter": cs["exchange_doubles"] * per_it,
            "iters": iters, "vcycles": vcyc, "converged": last["converged"], "final_rel": last["final_rel"],
            "solve_timer_ms": timer_ms, "setup_ms": setup[1], "setup_cold_ms": setup[0],
            # the first setup of a process pays the driver's allocator (the second runs out of the library's cache): how much of the
            # difference that is on THIS box (bimodal over the pool: tens of ms here, several hundred on other boxes)
            "setup_cold_alloc": setup_alloc[0], "setup_alloc": setup_alloc[1],
            "num_levels": amg.num_levels, "operator_complexity": o, "grid_complexity": g,
            "hbm_in_use_gb": hh.memory_stats()[0] / 1e9, "hbm_peak_gb": hh.memory_stats()[1] / 1e9,  # rank 0's allocator: resident / peak
            # aggregate over the ranks; fractions against n_gpus x 8 TB/s.  solve_phase_hbm_* = the bytes the kernels really stream (the
            # formats in HBM: level 0 is stencil-coded, the Galerkin levels windowed) per solve / time; solve_phase_csr_equiv_* = the SURVEY
            # 8(d) CSR figure of the same solve / time (exceeds what HBM delivers where a format is smaller than CSR: not a bandwidth).
            # N = 1 at 256^3 adds solve_phase_traffic_gb: FETCH_SIZE / WRITE_SIZE summed over every kernel of one solve
            "solve_phase_hbm_gbs": gbs(by[1], ms_per_step), "solve_phase_hbm_frac": gbs(by[1], ms_per_step) / (HBM_PEAK_GBS * w.size),
            "solve_phase_csr_equiv_gbs": gbs(by[0], ms_per_step), "solve_phase_csr_equiv_frac": gbs(by[0], ms_per_step) / (HBM_PEAK_GBS * w.size),
            "solve_phase_bytes_per_solve": {"format": by[1], "csr_equiv": by[0]},
            "dof_iters_per_s": N * iters / (ms_per_step * 1e-3),
            "roofline": {"kernel": (f"k_spmv_win<JACOBI> on rank 0's block of the level-{dom} operator ({dn} rows, {dnnz} nnz, windowed CSR: fp64 values, "
                                    f"2-byte column positions, {fbd['spmv'] / max(dnnz, 1):.1f} B streamed per entry all told): " if fbd["windowed"] else
                                    f"k_spmv_stream<JACOBI> on rank 0's block of the level-{dom} operator ({dn} rows, {dnnz} nnz, plain CSR): ")
                                   + ("largest l1-Jacobi sweep of the cycle" if aniso else "largest share of the solve") + f", {dom_count} launches timed inside it",
                         "bound": "hbm", "achieved": gbs(dom_bytes, dom_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs(dom_bytes, dom_ms) / HBM_PEAK_GBS,
                         "traffic": traffic.get(f"k_spmv_stream_jacobi_level{dom}_bytes_per_launch") if w.size == 1 and n == 256 else None,
                         "traffic_source": traffic_src if w.size == 1 and n == 256 else None,
                         "bytes_per_launch": dom_bytes, "format_bytes_per_launch": fbd["spmv"] + 16.0 * dn, "avg_ms": dom_ms},
            # level-0 PCG product (the north-star SpMV).  The operator is stencil-coded (1 B/entry), so the CSR-equivalent
            # rate exceeds what HBM can deliver; "format" is the honest HBM rate; plain_csr below is the uncoded kernel
            "level0_spmv": {"kernel": ("k_spmv_rowclass<PLAIN,DOT>" if fb0.get("row_coded") else "k_spmv_coded_row<PLAIN,DOT>") if fb0["coded"]
                            else ("k_spmv_win<PLAIN,DOT> (windowed CSR: 2-byte column positions; run form on a structured grid)" if fb0.get("windowed")
                                  else "k_spmv_stream<PLAIN,DOT>"),
                            "coded": fb0["coded"], "avg_ms": k1_ms, "launches": k1_count,
                            "csr_bytes_per_launch": k1_bytes, "csr_equiv_gbs": gbs(k1_bytes, k1_ms),
                            "csr_equiv_frac": gbs(k1_bytes, k1_ms) / HBM_PEAK_GBS,
                            "format_bytes_per_launch": k1_fmt, "format_gbs": gbs(k1_fmt, k1_ms),
                            "format_frac": gbs(k1_fmt, k1_ms) / HBM_PEAK_GBS,
                            "traffic": traffic.get("k_spmv_level0_bytes_per_launch") if w.size == 1 and n == 256 else None},
        }
        if "P0" in pr:
            for key, M, extra in (("level0_prolongation", P0, 8.0), ("level0_restriction", R0, 0.0)):
                mn, mc, mnnz = M.dims
                ms, cnt = pr["P0" if key.endswith("prolongation") else "R0"]
                bts = spmv_bytes(mn, mc, mnnz) + extra * mn  # P: x += P e reads x too
                out[key] = {"rows": mn, "cols": mc, "nnz": mnnz, "avg_ms": ms, "launches": cnt, "csr_bytes_per_launch": bts,
                            "csr_equiv_gbs": gbs(bts, ms), "csr_equiv_frac": gbs(bts, ms) / HBM_PEAK_GBS}
        if aniso:  # the byte model behind these four is the PCG iteration's: not quoted for GMRES
            for k in ("solve_phase_hbm_gbs", "solve_phase_hbm_frac", "solve_phase_csr_equiv_gbs", "solve_phase_csr_equiv_frac", "solve_phase_bytes_per_solve"):
                out[k] = None
        if "res0" in pr:
            ms, cnt = pr["res0"]
            bts = spmv_bytes(a_n, a_c, a_nnz) + 8.0 * a_n
            out["level0_residual"] = {"kernel": "k_spmv_stream<RESID> on the level-0 operator (plain CSR), inside the ILU(0) smoothing steps",
                                      "avg_ms": ms, "launches": cnt, "csr_bytes_per_launch": bts, "csr_equiv_gbs": gbs(bts, ms),
                                      "csr_equiv_frac": gbs(bts, ms) / HBM_PEAK_GBS}
        if not args.child:
            publish_headline(out)
        if w.size == 1 and not aniso:
            single_extras(args, out, hh, A, amg, fb0, iters)
    del A, amg
    h.destroy_solver()
    h.close()
    w.hdist.finalize()
    return out


def single_extras(args, out, hh, A, amg, fb0, iters):
    """N = 1 only, after the headline has been published: the CPU baseline (part of the line's contract: first), the counter passes
    behind roofline.traffic and solve_phase_traffic_gb, the same solve through the kernel-level seam, the per-kernel table, the
    plain-CSR / uncoded child runs, the reference's CPU-build defaults (cpu_defaults) and the aggressive-coarsening option.  All of
    them share the budget of publish_headline(); one that does not fit is recorded as skipped, by name."""
    skipped = []

    def fits(name, need_s):
        if budget_left() >= need_s:
            return True
        skipped.append(f"{name} (needs ~{need_s:.0f} s, {max(budget_left(), 0):.0f} s of HDA_BENCH_EXTRAS_BUDGET left)")
        return False

    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample or args.n, iters if (args.cpu_sample or args.n) == args.n else None)
        if "iters_match" in out["cpu_baseline"]:
            out["iters_match"] = out["cpu_baseline"]["iters_match"]
    if not args.no_traffic and args.n == 256 and fits("traffic", 30):
        t, why = measure_traffic(args, limit_s=min(240, budget_left()))
        rf = out["roofline"]
        if t is not None:
            dom_key = [k for k in t if k.startswith("k_spmv_stream_jacobi_level") and k.endswith("_bytes_per_launch")]
            if dom_key:
                rf["traffic"] = t[dom_key[0]]
                rf["traffic_source"] = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over a one-solve child "
                                        f"({t['seconds']:.0f} s), tools/pmc_traffic.py; FETCH_SIZE unit calibrated on k_cg_dir: x{t['calibration']['fetch_factor']:.3f}")
                rf["traffic_over_format_bytes"] = rf["traffic"] / rf["format_bytes_per_launch"]
            if t.get("k_spmv_level0_bytes_per_launch"):
                out["level0_spmv"]["traffic"] = t["k_spmv_level0_bytes_per_launch"]
            for key in ("level0_prolongation", "level0_restriction"):
                if key in out and t.get(key + "_bytes_per_launch"):
                    out[key]["traffic"] = t[key + "_bytes_per_launch"]
            if t.get("solve_phase_traffic_bytes"):
                # every kernel of ONE solve (the child's --steps 1), counters summed between the two marker kernels; against the format bytes
                # of the same solve: > 1 = re-reads that missed the caches (FETCH_SIZE counts Infinity-Cache hits too: fabric traffic)
                fmt = out["solve_phase_bytes_per_solve"]["format"]
                out["solve_phase_traffic_gb"] = t["solve_phase_traffic_bytes"] / 1e9
                out["traffic_over_format"] = t["solve_phase_traffic_bytes"] / fmt if fmt else None
                out["solve_phase_traffic_gbs"] = gbs(t["solve_phase_traffic_bytes"], out["ms_per_step"])
                out["solve_phase_traffic"] = t.get("solve_phase")
        else:
            rf["traffic_source"] = (rf.get("traffic_source") or "no committed profile") + f"; live counter passes failed: {why}"
    kp = hh.KrylovParams.default(False)
    res = hh.solve_device(A, amg, kp, nsolves=args.steps, profile_k1=False)
    sm = sorted(float(x) for x in res["solve_ms"])
    out["seam"] = {"what": "hda_solve_device on the same operator and hierarchy: Krylov loop alone, reference 'solve' timer boundaries",
                   "ms_per_step": sm[len(sm) // 2], "iters": res["iters"], "true_rel_res": res["true_rel"]}
    if not args.no_kernel_table:
        kt = {}
        for kind, name in ((0, "spmv"), (1, "l1_jacobi"), (2, "residual"), (3, "vcycle")):
            ms, by = hh.time_kernel(kind, A, amg if kind == 3 else None, 20)
            kt[name] = {"ms": ms, "csr_equiv_GB/s": by / ms / 1e6, "csr_equiv_frac": by / ms / 1e6 / HBM_PEAK_GBS}
            if kind == 3:
                kt[name]["format_GB/s"] = fb0["vcycle"] / ms / 1e6
        out["kernels"] = kt
    if not args.no_plain_csr and fb0["coded"]:
        if fits("plain_csr", 40):
            out["plain_csr"] = plain_csr_child(args)
        if fits("uncoded", 40):
            out["uncoded"] = plain_csr_child(args, window="1")
    if not args.no_cpu_defaults and fits("cpu_defaults", 40):
        out["cpu_defaults"] = cpu_defaults_run(args)
    if not args.no_cpu_defaults and args.n % 32 == 0 and args.n >= 64 and fits("cpu_defaults_rank_blocks", 60):
        out["cpu_defaults_rank_blocks"] = cpu_defaults_run(args, rank_grid=args.n // 32)
    if not args.no_aggressive and fits("aggressive_1", 20):
        out["aggressive_1"] = aggressive_run(args, hh)
    # BASELINE.json configs 4 and 5 on stand-in data (tools/side_configs.py), each with its oracle iteration check at a small size
    if not args.no_side_configs:
        for name, need in (("gmres_amg_ilu0", 60), ("gmres_mgr", 60)):
            if fits(name, need):
                out[name] = side_config_run(name, hh)
    if skipped:
        out["extras_skipped"] = skipped


def cpu_defaults_run(args, rank_grid=None):
    """(rank_grid = p: the same PDE in the numbering the reference's generator gives it at np = p^3, `-P p p p` -- laplacian.c:504-520 --
    handed over through HYPREDRV_LinearSystemSetMatrixFromCSR; row block q is then rank q's 32^3 sub-cube, which is what the reference's
    HMIS and hybrid sweeps see on p^3 ranks: blocks with an interior however many there are, where slabs of a lexicographic numbering
    lose theirs.)
    SURVEY 8(d)'s second series: the same system and API path with the reference's CPU-build defaults (HMIS coarsening, hybrid l1
    Gauss-Seidel 13 / 14; src/internal/amg.c:141-146, 182-189 -- the configuration its exact pins examples/refOutput/ex1.txt:27 and
    laplacian.txt:34-38 were made with) on V row blocks = the reference at np = V (V: the setup's own choice, in the object), and the
    oracle doing the same on the same V (iters_match) when the budget allows its serial setup."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("series_b", os.path.join(ROOT, "tools", "series_b.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        need = 200 if args.n >= 200 else 40  # the oracle's serial setup at 256^3 is about a minute, its block sweeps run on the host threads
        return mod.run(args.n, steps=max(min(args.steps, 3), 1), warmup=1, oracle=(not args.no_cpu_defaults_oracle) and budget_left() >= need,
                       rank_grid=rank_grid)
    except Exception as e:  # noqa: BLE001 - an extra must not lose the headline
        return {"error": repr(e)[:400]}


def side_config_run(name, hh):
    """BASELINE.json configs[3] (GMRES + MGR, examples/ex3.yml:11-23) and configs[4] (GMRES + AMG with the ILU(0) smoother,
    src/internal/ilu.c:15-28) through the same API path as the headline, on stand-in data (tools/side_configs.py says which and why)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("side_configs", os.path.join(ROOT, "tools", "side_configs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        return getattr(mod, name)(hh)
    except Exception as e:  # noqa: BLE001 - an extra must not lose the headline
        return {"error": repr(e)[:400]}


def aggressive_run(args, hh):
    """The same system and API path with ONE aggressive-coarsening level (reference option `preconditioner: amg: aggressive: num_levels: 1`,
    src/internal/amg.c:160-173, 938-944; NOT the benchmark's configuration, which is the reference's default preset): what the option
    buys on this workload -- a hierarchy of operator complexity ~1.33 instead of ~2.77, more iterations, each much cheaper."""
    from hypredrive_amd import hypredrv as hd
    n = args.n
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: 1\n")
    h.set_laplacian7((n, n, n))
    ts = []
    for rep in range(2):
        hh.sync()
        t0 = time.perf_counter()
        h.create_and_setup()
        hh.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 0:
            h.destroy_solver()
    A, amg = hh._lib.borrow(h)
    g, o = amg.complexities
    for _ in range(args.warmup):
        h.apply()
    hh.sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = h.apply()
    hh.sync()
    ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)
    res = {"what": "same system, same API path, `aggressive: num_levels: 1` (multipass interpolation on level 0): a reference OPTION, not the "
                   "benchmark's configuration", "ms_per_step": ms, "value": n ** 3 / (ms * 1e-3), "iters": last["iters"], "converged": last["converged"],
           "setup_ms": ts[1], "operator_complexity": o, "grid_complexity": g, "num_levels": amg.num_levels}
    del A, amg
    h.destroy_solver()
    h.close()
    return res


def plain_csr_child(args, window="0"):
    """The same bench in a child process with HDA_CODED=0 HDA_WINDOW=0 (no stencil / value coding, no windowed column indices), so
    the level-0 product is the plain CSR stream kernel of the north-star 'CSR SpMV >= 40 % of roofline' claim.  window="1": only the
    codings are off -- what a variable-coefficient operator of the same sparsity gets with default settings."""
    env = dict(os.environ, HDA_CODED="0", HDA_WINDOW=window)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--grid", str(args.n), "--no-cpu-baseline", "--no-kernel-table", "--no-plain-csr", "--no-aggressive", "--no-traffic", "--no-cpu-defaults", "--no-side-configs", "--child"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=max(min(300.0, budget_left()), 30.0))
    except subprocess.TimeoutExpired:
        return {"error": "child run did not finish within its share of the extras budget"}
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            c = json.loads(ln)
            k1 = c["level0_spmv"]
            return {"what": "child run with HDA_CODED=0 HDA_WINDOW=0: every operator in plain CSR (int32 col + fp64 val), the plain streaming kernel" if window == "0" else
                            "child run with HDA_CODED=0: no stencil / value coding (nothing that depends on the operator's VALUES repeating); windowed "
                            "column indices as for any matrix -- list form on the Galerkin levels, run form on the structured level 0",
                    "ms_per_step": c["ms_per_step"], "value": c["value"], "iters": c["iters"],
                    "level0_spmv_kernel": k1["kernel"], "level0_spmv_ms": k1["avg_ms"], "level0_spmv_gbs": k1["csr_equiv_gbs"],
                    "level0_spmv_frac": k1["csr_equiv_frac"], "solve_phase_hbm_frac": c["solve_phase_hbm_frac"],
                    "level0_prolongation_ms": c.get("level0_prolongation", {}).get("avg_ms"),
                    "level0_restriction_ms": c.get("level0_restriction", {}).get("avg_ms")}
    return {"error": f"child exited {r.returncode}: {r.stderr[-500:]}"}


def measure_traffic(args, limit_s=240):
    """HBM bytes per launch of the kernels the line quotes, measured NOW: two rocprofv3 counter passes (FETCH_SIZE, then WRITE_SIZE:
    separate runs, counters only, as MI355X_MICROARCH.md's HBM section prescribes) over a one-solve child of this script, reduced by
    tools/pmc_traffic.py (its k_cg_dir calibration of the gfx950 FETCH_SIZE unit included).  Returns (dict, None) or (None, why)."""
    import importlib.util
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, "this run is itself being profiled"
    tmp = tempfile.mkdtemp(prefix="hda_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    env["TMPDIR"] = "/tmp"
    try:
        t0 = time.perf_counter()
        for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            # the program itself after `--`: no env / shell hop between the profiler's preloaded library and python
            cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, sub), "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "0", "--grid", str(args.n), "--child",
                   "--no-cpu-baseline", "--no-kernel-table", "--no-plain-csr", "--no-aggressive", "--no-traffic", "--no-cpu-defaults", "--no-side-configs"]
            left = limit_s - (time.perf_counter() - t0)
            if left < 20:
                return None, f"counter passes exceeded {limit_s} s"
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=left)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {ctr} pass did not finish within {left:.0f} s"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} pass exited {r.returncode}: {r.stderr[-300:]}"
        spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        t = mod.compute(os.path.join(tmp, "fetch"), os.path.join(tmp, "write"), "live")
        t["seconds"] = time.perf_counter() - t0
        return t, None
    except Exception as e:  # noqa: BLE001 - the committed profile stays as the fallback
        return None, repr(e)[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
              "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
              "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCHELASTIC_ERROR_FILE")


def child_bench(extra, nranks, timeout_s):
    """One more bench in a CHILD launch (never a re-exec: this process has used the GPU), outside this job's rendezvous; returns
    its JSON line as a dict, or {"error": ...}.  nranks > 1: `python -m torch.distributed.run` like the driver's own launch."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    me = os.path.abspath(__file__)
    cmd = [sys.executable, me] if nranks == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
                                                    "--master-addr", "127.0.0.1", "--master-port", str(free_port()), me]
    cmd += ["--gpus", str(nranks), "--child", "--no-cpu-baseline", "--no-kernel-table", "--no-plain-csr", "--no-aggressive", "--no-traffic", "--no-cpu-defaults", "--no-side-configs"] + extra
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"child launch did not finish within {timeout_s} s"}
    err = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
        if ln.startswith("{") and '"error"' in ln and err is None:
            err = json.loads(ln)
    return err or {"error": f"child exited {r.returncode}: {r.stderr[-400:]}"}


def wait_for_sibling_ranks(limit_s=120.0):
    """Rank 0 starts child launches once its peers -- the other children of this job's launcher -- have exited and released their
    GPUs (on a box where several ranks share one card, the old and the new ranks together would exceed what the card admits)."""
    me, parent = os.getpid(), os.getppid()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < limit_s:
        alive = 0
        for d in os.listdir("/proc"):
            if not d.isdigit() or int(d) == me:
                continue
            try:
                with open(f"/proc/{d}/stat") as f:
                    st = f.read()
                ppid = int(st[st.rindex(")") + 2:].split()[1])
                if ppid != parent:
                    continue
                with open(f"/proc/{d}/cmdline", "rb") as f:
                    if b"bench.py" in f.read():
                        alive += 1
            except (OSError, ValueError):
                continue
        if alive == 0:
            return True
        time.sleep(0.5)
    return False


def multi_extras(args, out):
    """N > 1, rank 0, after the timed weak-scaling run: the OTHER curve and the same-node N = 1 point, so that one driver invocation
    per N yields both series BASELINE.md's >= 6x target can be read from (reference scripts/node_scaling.sh:1275-1292: the fixed-size
    series 256^3 at P = 1 .. 2x2x2, partition examples/src/C_laplacian/laplacian.c:561-582).
      strong_<n>:   the global <n>^3 problem (= one rank's block of the weak run) cut into N blocks, child launch of N ranks
      n1_reference: <n>^3 on ONE GPU of this node (rank 0's), same run, same code
    Failures of either child are recorded in their object and never touch the headline."""
    n, N = args.n, out["n_gpus"]
    base = ["--grid", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    keep = ("ms_per_step", "value", "iters", "converged", "setup_ms", "solve_timer_ms", "allreduces_per_iter", "halo_exchanges_per_iter",
            "collectives_per_iter", "halo_exchanges_overlapped_per_iter", "partitioned_levels", "levels_total", "transport", "ranks_seen")
    wait_for_sibling_ranks()
    st = child_bench(base + ["--strong"], N, 900)
    strong = {k: st.get(k) for k in keep if k in st} if "error" not in st else dict(st)
    strong["what"] = (f"child launch of {N} ranks: lap7 {n}^3 GLOBAL cut into {N} blocks (strong scaling), same API path and timing as the headline")
    if "config" in st:
        strong["parallelism"] = st["config"].get("parallelism")
    n1 = child_bench(base, 1, 900)
    ref = {k: n1.get(k) for k in keep if k in n1} if "error" not in n1 else dict(n1)
    ref["what"] = f"child run on ONE GPU of this node (rank 0's): lap7 {n}^3, the N = 1 point of both series, same run"
    out[f"strong_{n}"] = strong
    out["n1_reference"] = ref
    # the model the first measured curve is to be read against: what one PCG iteration of the weak run exchanges (counted by the library,
    # rank 0), priced with DESIGN section 5's assumptions, next to the measured kernel time of the same block on one GPU
    if ref.get("ms_per_step") and out.get("iters"):
        lat_x, lat_ar, link = 30.0, 25.0, 153.0  # us per grouped send/recv, us per small all-reduce, GB/s per xGMI link (MI355X_MICROARCH.md)
        xb = 8.0 * out["halo_doubles_per_iter"]
        hidden = out["halo_exchanges_overlapped_per_iter"]
        exposed_us = (out["halo_exchanges_per_iter"] - hidden) * lat_x + out["allreduces_per_iter"] * lat_ar + xb / (link * 1e3)
        out["comm_model"] = {
            "per_iteration": {"halo_exchanges": out["halo_exchanges_per_iter"], "of_them_under_a_product_kernel": hidden, "halo_bytes": xb,
                              "allreduces": out["allreduces_per_iter"], "allreduce_doubles": out["allreduce_doubles_per_iter"]},
            "assumed": {"exchange_latency_us": lat_x, "allreduce_latency_us": lat_ar, "xgmi_link_GBs": link,
                        "note": "design estimates (DESIGN section 5), never measured: this run is their first measurement"},
            "projected_exposed_comm_ms_per_solve": exposed_us * out["iters"] * 1e-3,
            "n1_kernel_ms_per_solve": ref["ms_per_step"],
            "projected_weak_ms_per_solve": ref["ms_per_step"] + exposed_us * out["iters"] * 1e-3,
            "measured_weak_ms_per_solve": out["ms_per_step"]}
    if ref.get("value") and out.get("value") and out["scaling"] == "weak":
        out["speedup_weak_dofs"] = out["value"] / ref["value"]  # DOF/s at N GPUs (N x the rows) over DOF/s at 1 GPU: ideal = N
    if ref.get("ms_per_step") and strong.get("ms_per_step"):
        out["speedup_strong"] = ref["ms_per_step"] / strong["ms_per_step"]  # same problem, N GPUs over 1: ideal = N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)  # (a solve is 34 ms: ten of them average out a one-off host hiccup that five do not)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", dest="n", type=int, default=0, help="grid points per dimension (per rank; global with --strong or "
                    "--workload aniso); default 256 (lap7) / 128 (aniso)")
    ap.add_argument("--workload", choices=["lap7", "aniso"], default="lap7", help="lap7: BASELINE configs 2 / 3 (the headline metric). aniso: "
                    "BASELINE config 5 stand-in, GMRES + AMG with the ILU(0) smoother on a heterogeneous anisotropic reservoir operator")
    ap.add_argument("--cpu-sample", type=int, default=0, help="grid size of the CPU-baseline run (default: the benchmark's own --grid; "
                    "256^3 takes about 2 minutes of host time, most of it the oracle's setup)")
    ap.add_argument("--strong", action="store_true", help="N > 1: --grid is the GLOBAL problem, cut into N blocks (fixed-size series). Default is "
                    "weak scaling: --grid is the block of every rank (global grid = block x rank grid; 256 on 8 GPUs = "
                    "BASELINE config 3, 512^3); the weak line of N > 1 also carries the strong series' point (strong_<grid>) and the "
                    "same-node N = 1 point (n1_reference) from child launches")
    ap.add_argument("--weak", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--no-plain-csr", action="store_true")
    ap.add_argument("--no-aggressive", action="store_true", help="N = 1: skip the side run with one aggressive-coarsening level")
    ap.add_argument("--no-cpu-defaults", action="store_true", help="N = 1: skip the side run with the reference's CPU-build defaults (HMIS, hybrid l1 Gauss-Seidel)")
    ap.add_argument("--no-cpu-defaults-oracle", action="store_true", help="cpu_defaults without the oracle's run on the same row blocks (iters_match)")
    ap.add_argument("--no-side-configs", action="store_true", help="N = 1: skip gmres_amg_ilu0 / gmres_mgr (BASELINE configs 5 and 4 on stand-in data)")
    ap.add_argument("--no-traffic", action="store_true", help="N = 1, 256^3: skip the two rocprofv3 --pmc passes behind roofline.traffic "
                    "(the committed profiles/traffic.json is quoted instead)")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: skip the strong_<grid> and n1_reference child launches")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # a launch made by another bench.py: no further children
    args = ap.parse_args()
    signal.signal(signal.SIGTERM, _on_term)
    if args.n <= 0:
        args.n = 128 if args.workload == "aniso" else 256
    world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world is None:
        sys.exit(spawn_ranks(args))
    if world is not None and int(world) != args.gpus:
        print(f"bench.py: launcher started {world} ranks, --gpus says {args.gpus}: using {world}", file=sys.stderr)
    if world is not None and int(world) > 1:
        # a multi-rank run that hangs (a collective one rank never enters) must say where before the launcher's own limit ends it silently
        import threading
        limit = float(os.environ.get("HDA_BENCH_TIMEOUT", "900"))

        def hang():
            error_line("hang:" + PHASE["name"], int(os.environ.get("RANK", "0")), f"bench.py did not finish within {limit:.0f} s")
            os._exit(4)

        wd = threading.Timer(limit, hang)
        wd.daemon = True
        wd.start()
    else:
        wd = None
    out = run(args)
    if wd is not None:
        wd.cancel()  # (the child launches below have limits of their own)
    if out is not None:
        PHASE["name"] = "extras (strong / n1 child launches)"
        if out["n_gpus"] > 1 and not (args.child or args.no_extras or args.strong or args.workload != "lap7" or out.get("rccl_error")):
            try:
                multi_extras(args, out)
            except Exception as e:  # noqa: BLE001 - the headline is measured: extras must not lose it
                out["extras_error"] = repr(e)[:400]
        print_line(out)
        if out.get("rccl_error"):
            sys.exit(2)  # measured over the host-staged transport because RCCL could not be joined: a result line, and a failure


if __name__ == "__main__":
    main()
