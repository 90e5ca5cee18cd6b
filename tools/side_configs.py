#!/usr/bin/env python3
"""BASELINE.json configs 4 and 5 beside the headline, on STAND-IN DATA (the reference's data sets are unreachable offline):

  gmres_mgr       config 4, examples/ex3.yml:11-23 -- GMRES(30) + MGR (two reduction levels: jacobi prolongation; l1-hsgs global
                  relaxation + column-lumped restriction; BoomerAMG on the coarsest system) -- on the three-field model system of
                  tools/make_threefield.py scaled up (compflow6k, 5625 rows, is not in the reference tree)
  gmres_amg_ilu0  config 5, src/internal/ilu.c:15-28 + amg.c:899-921 -- GMRES(30) + BoomerAMG with the ILU(0) complex smoother on level
                  0 (bj-iluk, Jacobi-iterative triangular solves) -- on the heterogeneous anisotropic reservoir operator of
                  hypredrive_amd/synthetic.py (SPE10 model 2 is not reachable)

Each runs through HYPREDRV_LinearSolverSetup / Apply like the headline and returns: iterations, ms per solve, setup, the dominant
kernel's SURVEY 8(d) bytes / time, and `iters_match`: the same configuration on a size the oracle finishes in seconds, device
iteration count == oracle's.  bench.py carries them as budgeted extras; `python tools/side_configs.py [mgr|ilu0] [size]` runs one."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0
YAML_MGR = ("solver: gmres\npreconditioner:\n  mgr:\n    level:\n      0:\n        f_dofs: [2]\n        prolongation_type: jacobi\n"
            "      1:\n        f_dofs: [1]\n        g_relaxation: l1-hsgs\n        restriction_type: columped\n    coarsest_level: amg\n")
YAML_ILU0 = ("solver:\n  gmres:\n    relative_tol: 1.0e-6\npreconditioner:\n  amg:\n    smoother:\n      type: ilu\n      num_levels: 1\n"
             "      ilu:\n        type: bj-iluk\n        tri_solve: 0\n")
MGR_LEVELS = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")]


def _threefield(cells):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_threefield", os.path.join(ROOT, "tools", "make_threefield.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.system(cells)


def _timed(hh, h, steps, warmup):
    ts = []
    for rep in range(2):  # the first setup of a process also pays for the allocator (bench.py's protocol)
        hh.sync()
        t0 = time.perf_counter()
        h.create_and_setup()
        hh.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 0:
            h.destroy_solver()
    return ts


def _solves(hh, h, steps, warmup):
    for _ in range(warmup):
        h.apply()
    hh.sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = h.apply()
    hh.sync()
    return (time.perf_counter() - t0) * 1e3 / max(steps, 1), last


def gmres_mgr(hh, cells=512, steps=3, warmup=1, oracle_cells=24, with_oracle=True):
    import ctypes as C
    from hypredrive_amd import hypredrv as hd
    S, labels = _threefield(cells)
    N = S.shape[0]
    h = hd.Hypredrv(YAML_MGR)
    h.set_matrix_csr(0, N - 1, S.indptr, S.indices, S.data)
    h.set_rhs_array(0, N - 1, np.ones(N))
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    hd.check(hd.lib().HYPREDRV_LinearSystemSetDofmap(h.h, N, lab.ctypes.data_as(C.POINTER(C.c_int))))
    h.finish_system()
    setup = _timed(hh, h, steps, warmup)
    A = hh._lib.borrow_matrix(h)
    hh.probe_spmv(None, 0)
    k1 = hh._lib.probe_add(A, 0)
    ms, last = _solves(hh, h, steps, warmup)
    k1_ms, k1_n = hh._lib.probe_read_id(k1)
    hh.probe_spmv(None, 0)
    nnz = int(S.nnz)
    by = 12.0 * nnz + 4.0 * (N + 1) + 16.0 * N
    out = {"what": "BASELINE config 4 on STAND-IN DATA: GMRES(30) + MGR with the solver / preconditioner block of examples/ex3.yml:11-23 on the "
                   f"three-field model system of tools/make_threefield.py at {cells} x {cells} cells (compflow6k is not in the reference tree); "
                   "the l1-hsgs global relaxation runs on the row blocks the setup announces on stderr (the reference at np = V, as BoomerAMG's "
                   "hybrid sweeps; HDA_BLOCKS=1: the sequential np = 1 sweep, 39 ms per solve); parity unpinned (device == oracle only)",
           "rows": N, "nnz": nnz, "iters": last["iters"], "converged": last["converged"], "final_rel": last["final_rel"],
           "ms_per_step": ms, "value": N / (ms * 1e-3), "unit": "DOF/s", "setup_ms": setup[1], "setup_cold_ms": setup[0],
           "dominant_kernel": {"kernel": "level-0 product of the GMRES iteration (K1)", "bytes_per_launch": by, "avg_ms": k1_ms, "launches": k1_n,
                               "gbs": by / k1_ms / 1e6 if k1_ms else None, "frac": by / k1_ms / 1e6 / HBM_PEAK_GBS if k1_ms else None}}
    del A
    h.destroy_solver()
    h.close()
    if with_oracle:
        from oracle import oracle_ffi as o
        So, lo = _threefield(oracle_cells)
        Ao, Ah = o.Csr.from_scipy(So), hh.Csr.from_scipy(So)
        b = np.ones(So.shape[0])
        ro, rh = o.gmres(Ao, b, o.MgrPrecond(Ao, lo, MGR_LEVELS)), hh.gmres(Ah, b, hh.Mgr(Ah, lo, MGR_LEVELS))
        out["oracle_check"] = {"cells": oracle_cells, "rows": int(So.shape[0]), "device_iters": rh["iters"], "oracle_iters": ro["iters"]}
        out["iters_match"] = bool(rh["converged"] and rh["iters"] == ro["iters"])
    return out


def gmres_amg_ilu0(hh, n=128, steps=3, warmup=1, oracle_n=20, with_oracle=True):
    from hypredrive_amd import hypredrv as hd
    from hypredrive_amd.synthetic import spe10_like
    ip, ix, v, b = spe10_like(n)
    N = n ** 3
    nnz0 = int(ip[-1])
    h = hd.Hypredrv(YAML_ILU0)
    h.set_matrix_csr(0, N - 1, ip, ix, v)
    h.set_rhs_array(0, N - 1, b)
    h.finish_system()
    del ip, ix, v, b
    setup = _timed(hh, h, steps, warmup)
    A, amg = hh._lib.borrow(h)
    L = amg.num_levels
    dims = [amg.level_matrix(l, 0).dims for l in range(L)]
    dom = max(range(1, L), key=lambda l: dims[l][2]) if L > 1 else 0
    Ad = amg.level_matrix(dom, 0)
    hh.probe_spmv(None, 0)
    pd, p0 = hh._lib.probe_add(Ad, 2), hh._lib.probe_add(A, 1)
    ms, last = _solves(hh, h, steps, warmup)
    d_ms, d_n = hh._lib.probe_read_id(pd)
    r_ms, r_n = hh._lib.probe_read_id(p0)
    hh.probe_spmv(None, 0)
    nd, _, zd = dims[dom]
    by = 12.0 * zd + 4.0 * (nd + 1) + 32.0 * nd          # B_spmv + 16 n
    by0 = 12.0 * nnz0 + 4.0 * (N + 1) + 24.0 * N         # residual: B_spmv + 8 n
    g, oc = amg.complexities
    out = {"what": "BASELINE config 5 on STAND-IN DATA: GMRES(30) + BoomerAMG with the ILU(0) complex smoother on level 0 (bj-iluk, Jacobi-iterative "
                   f"triangular solves; src/internal/ilu.c:15-28, amg.c:899-921) on the heterogeneous anisotropic reservoir operator of "
                   f"hypredrive_amd/synthetic.py at {n}^3 (SPE10 is not reachable offline)",
           "rows": N, "nnz": nnz0, "iters": last["iters"], "converged": last["converged"], "final_rel": last["final_rel"],
           "ms_per_step": ms, "value": N / (ms * 1e-3), "unit": "DOF/s", "setup_ms": setup[1], "setup_cold_ms": setup[0],
           "num_levels": L, "operator_complexity": oc,
           "dominant_kernel": {"kernel": f"l1-Jacobi sweep on the level-{dom} operator ({nd} rows, {zd} entries)", "bytes_per_launch": by, "avg_ms": d_ms,
                               "launches": d_n, "gbs": by / d_ms / 1e6 if d_ms else None, "frac": by / d_ms / 1e6 / HBM_PEAK_GBS if d_ms else None},
           "level0_residual": {"kernel": "residual on the level-0 operator (the ILU smoother's and GMRES' products)", "bytes_per_launch": by0, "avg_ms": r_ms,
                               "launches": r_n, "gbs": by0 / r_ms / 1e6 if r_ms else None, "frac": by0 / r_ms / 1e6 / HBM_PEAK_GBS if r_ms else None}}
    del A, amg, Ad
    h.destroy_solver()
    h.close()
    if with_oracle:
        import scipy.sparse as sp
        from oracle import oracle_ffi as o
        ip, ix, v, b = spe10_like(oracle_n)
        S = sp.csr_matrix((v, ix, ip), shape=(oracle_n ** 3, oracle_n ** 3))
        Ao, Ah = o.Csr.from_scipy(S), hh.Csr.from_scipy(S)
        ao = o.Amg(Ao, o.amg_params(True))
        ao.set_ilu_smoother(num_levels=1, num_sweeps=1, tri_solve=0)
        ah = hh.Amg(Ah, hh.AmgParams.default(smooth_num_levels=1, smooth_num_sweeps=1, ilu_tri_solve=0))
        ro, rh = o.gmres(Ao, b, ao), hh.gmres(Ah, b, ah)
        out["oracle_check"] = {"grid": oracle_n, "rows": oracle_n ** 3, "device_iters": rh["iters"], "oracle_iters": ro["iters"]}
        out["iters_match"] = bool(rh["converged"] and rh["iters"] == ro["iters"])
    return out


if __name__ == "__main__":
    import hypredrive_amd as hh
    which = sys.argv[1] if len(sys.argv) > 1 else "mgr"
    if which == "mgr":
        print(json.dumps(gmres_mgr(hh, int(sys.argv[2]) if len(sys.argv) > 2 else 512)))
    else:
        print(json.dumps(gmres_amg_ilu0(hh, int(sys.argv[2]) if len(sys.argv) > 2 else 128)))
