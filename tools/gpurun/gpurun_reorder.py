"""Experiment: does a hierarchy-induced clustered numbering of the level-1 unknowns speed up the
level-1 products?  parent(i) = column of the largest |P_ij|; rank recursively from the coarsest level."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import time
import numpy as np
import scipy.sparse as sp
import hypredrive_amd as h
n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
A0 = h.lap7(n, n, n, want_rhs=False)
amg = h.Amg(A0)
L = amg.num_levels
def get(l, which):
    rp, cj, v = amg.level_matrix(l, which).download()
    m = amg.level_matrix(l, which)
    return sp.csr_matrix((v, cj, rp), shape=(m.nrows, m.ncols))
t0 = time.time()
P = {l: get(l, 1) for l in range(1, min(L - 1, 1 + depth))}
A1 = get(1, 0)
print(f"download {time.time()-t0:.1f}s; level1 {A1.shape[0]} rows {A1.nnz} nnz", flush=True)
# parent of every level-l point = strongest interpolation source (C points: themselves)
def parent(Pl):
    Pa = abs(Pl).tocsr()
    rp, cj, v = Pa.indptr, Pa.indices, Pa.data
    par = np.zeros(Pa.shape[0], dtype=np.int64)
    # argmax per row
    rowid = np.repeat(np.arange(Pa.shape[0]), np.diff(rp))
    order = np.lexsort((-v, rowid))
    first = np.concatenate([[0], np.cumsum(np.diff(rp))[:-1]])
    nonempty = np.diff(rp) > 0
    par[nonempty] = cj[order[first[nonempty]]]
    return par
lv = sorted(P.keys())
rank = np.arange(P[lv[-1]].shape[1], dtype=np.int64)  # natural order on the deepest level used
for l in reversed(lv):
    par = parent(P[l])
    key = rank[par] * (P[l].shape[0] + 1) + np.arange(P[l].shape[0])
    perm = np.argsort(key, kind="stable")          # new position -> old index
    rank = np.empty_like(perm); rank[perm] = np.arange(perm.size)
print(f"ordering {time.time()-t0:.1f}s", flush=True)
perm1 = np.argsort(rank)
A1p = A1[perm1][:, perm1].tocsr(); A1p.sort_indices()
def bandstat(M):
    rowid = np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))
    d = np.abs(M.indices - rowid)
    return np.median(d), np.percentile(d, 90), (d < 32768).mean(), (d < 2048).mean()
print("natural   |col-row| median/p90/frac<32768/frac<2048:", bandstat(A1), flush=True)
print("clustered |col-row| median/p90/frac<32768/frac<2048:", bandstat(A1p), flush=True)
for name, M in (("natural", A1), ("clustered", A1p)):
    Md = h.Csr.from_arrays(M.shape[0], M.shape[1], M.indptr, M.indices, M.data)
    for kind, kn in ((0, "spmv"), (1, "jacobi")):
        ms, by = h.time_kernel(kind, Md, None, 30)
        print(f"{name:10s} {kn}: {ms*1e3:8.1f} us  {by/ms/1e6:6.0f} GB/s", flush=True)
