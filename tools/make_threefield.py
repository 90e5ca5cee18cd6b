#!/usr/bin/env python3
"""Write a small 3-field model system in the file layout of the reference's MGR examples (examples/ex3.yml reads
data/compflow6k/np1/{IJ.out.A, IJ.out.b, dofmap.out}; that data set is not in the reference tree and Zenodo is unreachable
offline): hypre ASCII IJ matrix / vector parts and a dofmap part (count, then one label per row --
src/internal/containers.c:443-620 of the reference).  Three unknowns per cell, interleaved, labels 0 (pressure-like,
diffusive), 1, 2 (cell-local fields): the structure ex3.yml's MGR block eliminates level by level.
usage: make_threefield.py [root=.] [n=16]  ->  <root>/data/threefield/np1/"""
import os
import sys

import numpy as np
import scipy.sparse as sp


def system(n=16, seed=0):
    A1 = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    L2 = sp.kronsum(A1, A1).tocsr()
    nc = n * n
    I = sp.identity(nc)
    rng = np.random.default_rng(seed)
    d = lambda lo, hi: sp.diags(rng.uniform(lo, hi, nc))
    K = sp.bmat([[L2 + d(0.5, 1.0), d(0.1, 0.3), d(0.05, 0.1)],
                 [d(0.1, 0.2), 2.0 * I + 0.1 * L2, d(0.05, 0.1)],
                 [d(0.05, 0.1), d(0.1, 0.2), d(2.5, 3.5)]]).tocsr()
    perm = np.arange(3 * nc).reshape(3, nc).T.ravel()
    Kp = K[perm][:, perm].tocsr()
    Kp.sort_indices()
    return Kp, np.tile([0, 1, 2], nc)


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "."
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    S, labels = system(n)
    N = S.shape[0]
    d = os.path.join(root, "data", "threefield", "np1")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "IJ.out.A.00000"), "w") as f:
        f.write(f"0 {N - 1} 0 {N - 1}\n")
        for i in range(N):
            for k in range(S.indptr[i], S.indptr[i + 1]):
                f.write(f"{i} {S.indices[k]} {S.data[k]:.17e}\n")
    with open(os.path.join(d, "IJ.out.b.00000"), "w") as f:
        f.write(f"0 {N - 1}\n" + "".join(f"{i} 1.0\n" for i in range(N)))
    with open(os.path.join(d, "dofmap.out.00000"), "w") as f:
        f.write(f"{N}\n" + "".join(f"{v}\n" for v in labels))
    print(f"wrote {d}: {N} rows, {S.nnz} nonzeros, labels 0/1/2")


if __name__ == "__main__":
    main()
