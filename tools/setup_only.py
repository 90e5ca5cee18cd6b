#!/usr/bin/env python3
"""Two BoomerAMG setups of the benchmark system through HYPREDRV_LinearSolverSetup (the "prec" timer's content), the second one
bracketed by marker kernels: the program tools/setup_accounting.py's trace and counter passes run (GPU).  usage: setup_only.py [n=256]"""
import json
import os
import sys
import time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hypredrive_amd as hh
from hypredrive_amd import hypredrv as hd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h = hd.Hypredrv("solver: pcg\npreconditioner:\n  preset: poisson\n")
h.set_laplacian7((n, n, n))
ms = []
for rep in range(2):
    hh.sync()
    if rep == 1:
        hh.load().hda_marker(1)
    t0 = time.perf_counter()
    h.create_and_setup()
    hh.sync()
    ms.append((time.perf_counter() - t0) * 1e3)
    if rep == 1:
        hh.load().hda_marker(2)
    if rep == 0:
        h.destroy_solver()
A, amg = hh._lib.borrow(h)
L = amg.num_levels
dims = []
for l in range(L):
    a = amg.level_matrix(l, 0).dims
    p = amg.level_matrix(l, 1).dims if l < L - 1 else (0, 0, 0)
    dims.append({"level": l, "rows": a[0], "nnz": a[2], "P_cols": p[1], "P_nnz": p[2]})
print(json.dumps({"n": n, "setup_ms": ms, "levels": dims}))
