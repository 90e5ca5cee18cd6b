"""Chebyshev smoother (relax type 16) against l1-Jacobi on the benchmark problem.  usage: gpurun_cheby.py [n=256]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import time
import numpy as np
import hypredrive_amd as h

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = h.lap7(n, n, n, want_rhs=False)
kp = h.KrylovParams.default(False)
for label, prm in (("l1-Jacobi V(1,1)", {}), ("Chebyshev order 2", dict(relax_down=16, relax_up=16)),
                   ("Chebyshev order 3", dict(relax_down=16, relax_up=16, cheby_order=3)),
                   ("Chebyshev order 4, fraction 0.1", dict(relax_down=16, relax_up=16, cheby_order=4, cheby_fraction=0.1))):
    h.sync(); t0 = time.perf_counter()
    amg = h.Amg(A, h.AmgParams.default(**prm))
    h.sync(); ts = (time.perf_counter() - t0) * 1e3
    h.solve_device(A, amg, kp, nsolves=1, profile_k1=False)
    res = h.solve_device(A, amg, kp, nsolves=3, profile_k1=False)
    print(f"{n}^3 AMG-PCG, {label}: setup {ts:.0f} ms, {res['iters']} iterations, {np.median(res['solve_ms']):.1f} ms per solve", flush=True)
    del amg
