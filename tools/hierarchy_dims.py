#!/usr/bin/env python3
"""Rows / entries of every level's A, P (and R) of the benchmark hierarchy: python tools/hierarchy_dims.py [n]  (GPU)."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hypredrive_amd as hh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = hh.lap7(n, n, n)
amg = hh.Amg(A)
L = amg.num_levels
L = L if isinstance(L, int) else L()
for l in range(L):
    a = amg.level_matrix(l, 0).dims
    line = f"level {l}: A {a[0]} rows {a[2]} entries ({a[2] / max(a[0], 1):.1f} per row)"
    if l < L - 1:
        p = amg.level_matrix(l, 1).dims
        line += f"; P {p[0]} x {p[1]}, {p[2]} entries"
    print(line, flush=True)
