import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import numpy as np
import hypredrive_amd as h
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
A = h.lap7(n, n, n, want_rhs=False)
amg = h.Amg(A)
for l in range(0, min(4, amg.num_levels)):
    for which, name in ((0, "A"), (1, "P")):
        if which == 1 and l >= amg.num_levels - 1:
            continue
        rp, cj, v = amg.level_matrix(l, which).download()
        u, c = np.unique(v, return_counts=True)
        c.sort()
        top = c[::-1]
        cov = lambda k: top[:k].sum() / v.size
        print(f"L{l} {name}: nnz {v.size} distinct values {u.size} ({u.size / v.size:.4f}); coverage top255 {cov(255):.3f} top65535 {cov(65535):.3f}", flush=True)
