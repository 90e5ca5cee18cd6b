#!/usr/bin/env python3
"""AMG-PCG with the hybrid l1 Gauss-Seidel smoother (13 down / 14 up, the reference's CPU-build default) on an n^3 Laplacian: ms per solve.
usage: gs_probe.py <n> [solves]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypredrive_amd import _lib  # noqa: E402
from hypredrive_amd import hypredrv as hd  # noqa: E402

n = int(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    relaxation:\n      down_type: 13\n      up_type: 14\n")
h.set_laplacian7((n, n, n))
h.create_and_setup()
h.apply()
_lib.sync()
t0 = time.perf_counter()
for _ in range(k):
    r = h.apply()
_lib.sync()
print("hl1GS %d^3: %.2f ms per solve, %d iterations, converged %s" % (n, (time.perf_counter() - t0) * 1e3 / k, r["iters"], r["converged"]), flush=True)
h.destroy_solver()
h.close()
