"""Synthetic inputs for BASELINE config 5 (SPE10 / anisotropic diffusion: the real data set is not reachable offline).

spe10_like(): a 7-point finite-volume discretisation of -div(K grad u) = f on an n^3 Cartesian grid with a heterogeneous,
anisotropic diagonal permeability tensor in the manner of SPE10 model 2: log-normal horizontal permeability varying
over several decades from cell to cell, layered in z, and a vertical permeability `kv_kh` times the horizontal one;
face transmissibilities are harmonic means of the two cell values (the discretisation reservoir simulators use).
Dirichlet conditions by truncation (as the reference's generator, examples/src/C_laplacian/laplacian.c:719-921) keep
the operator nonsingular.  Nothing about it is constant-coefficient: no operator of the hierarchy can be stencil-coded,
so the plain-CSR kernels and the ILU(0) smoother run on level 0.  Deterministic in (n, seed).
"""
import numpy as np


def spe10_like(n, kv_kh=1.0e-3, decades=3.0, seed=10, dtype=np.float64):
    """Returns (indptr int64, indices int64, data, rhs) of the n^3 system, rows in lexicographic order (x fastest)."""
    rng = np.random.default_rng(seed)
    # layered log-normal field: a layer mean per z plane plus cell-wise variation, clipped to `decades` decades
    layer = rng.normal(0.0, 1.0, size=n)[:, None, None]
    cell = rng.normal(0.0, 1.0, size=(n, n, n))
    logk = np.clip(0.6 * layer + 0.8 * cell, -2.0, 2.0) * (decades / 4.0) * np.log(10.0)
    kh = np.exp(logk)            # [z, y, x]
    kz = kv_kh * kh

    def harm(a, b):
        return 2.0 * a * b / (a + b)

    N = n ** 3
    idx = np.arange(N, dtype=np.int64).reshape(n, n, n)
    tx = harm(kh[:, :, :-1], kh[:, :, 1:])     # faces between x and x+1
    ty = harm(kh[:, :-1, :], kh[:, 1:, :])
    tz = harm(kz[:-1, :, :], kz[1:, :, :])
    rows, cols, vals = [], [], []
    diag = np.zeros((n, n, n))
    for t, lo, hi in ((tx, idx[:, :, :-1], idx[:, :, 1:]), (ty, idx[:, :-1, :], idx[:, 1:, :]), (tz, idx[:-1, :, :], idx[1:, :, :])):
        rows += [lo.ravel(), hi.ravel()]
        cols += [hi.ravel(), lo.ravel()]
        vals += [-t.ravel(), -t.ravel()]
    # diagonal: sum of the transmissibilities of all six faces; boundary faces use the cell's own value (Dirichlet by truncation)
    diag[:, :, :-1] += tx; diag[:, :, 1:] += tx; diag[:, :, 0] += kh[:, :, 0]; diag[:, :, -1] += kh[:, :, -1]
    diag[:, :-1, :] += ty; diag[:, 1:, :] += ty; diag[:, 0, :] += kh[:, 0, :]; diag[:, -1, :] += kh[:, -1, :]
    diag[:-1, :, :] += tz; diag[1:, :, :] += tz; diag[0, :, :] += kz[0, :, :]; diag[-1, :, :] += kz[-1, :, :]
    rows.append(idx.ravel()); cols.append(idx.ravel()); vals.append(diag.ravel())
    r = np.concatenate(rows); c = np.concatenate(cols); v = np.concatenate(vals).astype(dtype)
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    indptr = np.zeros(N + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    indptr = np.cumsum(indptr)
    # source / sink pair (an injector and a producer column), as a quarter five-spot would have
    rhs = np.zeros(N)
    rhs[idx[:, 0, 0]] = 1.0
    rhs[idx[:, -1, -1]] = -1.0
    return indptr, c.astype(np.int64), v, rhs
