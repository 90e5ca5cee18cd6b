"""ILU(0) measurements on one MI355X: factorisation and application of the block-Jacobi ILU(0) of the 7-pt
Laplacian (level-scheduled exact substitutions vs Jacobi-iterative triangular solves, ilu.c:21-23), and
AMG-PCG with ILU as level-0 complex smoother next to the l1-Jacobi baseline.  usage: gpurun_ilu.py [n=256]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import time
import numpy as np
import hypredrive_amd as h

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = h.lap7(n, n, n, want_rhs=False)
N, _, nnz = A.dims
print(f"lap7 {n}^3: {N} rows, {nnz} nnz", flush=True)
r = np.ones(N)
for ts, name in ((1, "exact substitutions (level scheduled)"), (0, "5+5 Jacobi iterations")):
    h.sync(); t0 = time.perf_counter()
    F = h.Ilu(A, tri_solve=ts)
    h.sync(); t_setup = (time.perf_counter() - t0) * 1e3
    F.apply(r)  # warm
    reps = 5
    h.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        z = F.apply(r)
    h.sync(); t_apply = (time.perf_counter() - t0) * 1e3 / reps  # includes the 2 x 134 MB host transfers of the test entry
    print(f"ILU(0) {name}: setup {t_setup:.1f} ms, apply (with host copies) {t_apply:.2f} ms", flush=True)
    del F
kp = h.KrylovParams.default(False)
for label, prm in (("l1-Jacobi V(1,1) (baseline)", {}),
                   ("ILU(0) smoother on level 0, Jacobi-iterative solves", dict(smooth_num_levels=1, ilu_tri_solve=0)),
                   ("ILU(0) smoother on level 0, exact solves", dict(smooth_num_levels=1, ilu_tri_solve=1))):
    h.sync(); t0 = time.perf_counter()
    amg = h.Amg(A, h.AmgParams.default(**prm))
    h.sync(); ts = (time.perf_counter() - t0) * 1e3
    h.solve_device(A, amg, kp, nsolves=1, profile_k1=False)
    res = h.solve_device(A, amg, kp, nsolves=3, profile_k1=False)
    print(f"AMG-PCG, {label}: setup {ts:.0f} ms, {res['iters']} iterations, {np.median(res['solve_ms']):.1f} ms per solve, "
          f"true rel res {res['true_rel']:.2e}", flush=True)
    del amg
