"""Pin the CPU oracle against every numeric anchor the reference holds for this path
(numbers extracted into tests/golden/ref_pins.json by
tests/golden/make_ref_pins.py) and against scipy as an independent check."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def test_lap7_shape_and_values(orc, pins):
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    assert A.nrows == pins["ex1"]["rows"] and A.nnz == pins["ex1"]["nnz"]
    S = A.to_scipy()
    # row sums min 0 / max 3, entries/row 4..7 (examples/refOutput/ex2.txt:125)
    rs = np.asarray(S.sum(axis=1)).ravel()
    assert rs.min() == 0.0 and rs.max() == 3.0
    cnt = np.diff(S.indptr)
    assert cnt.min() == 4 and cnt.max() == 7
    assert np.all(S.diagonal() == 6.0)
    assert abs(np.linalg.norm(b) - pins["ex1"]["stats"][0]["r0"]) < 0.05  # 3.16e+01
    assert (S != S.T).nnz == 0


def test_lap7_generator_rhs_and_blocks(orc, pins):
    A, b = orc.lap7(10, 10, 10, b_mode=0)
    assert np.linalg.norm(b) == pytest.approx(pins["laplacian"]["stats"][0]["r0"])  # 1.00e+01
    # block-partitioned numbering is a symmetric permutation of the lexicographic one
    A1 = A.to_scipy()
    A2, b2 = orc.lap7(10, 10, 10, P=(2, 2, 1))
    A2 = A2.to_scipy()
    assert A2.nnz == A1.nnz and (A2 != A2.T).nnz == 0
    e1 = np.sort(spla.eigsh(A1, k=3, which="LA", return_eigenvectors=False))
    e2 = np.sort(spla.eigsh(A2, k=3, which="LA", return_eigenvectors=False))
    assert np.allclose(e1, e2, rtol=1e-10)
    assert b2.sum() == b.sum()
    lo0, hi0 = orc.lap7_partition(10, 10, 10, (2, 2, 1), 0)
    lo3, hi3 = orc.lap7_partition(10, 10, 10, (2, 2, 1), 3)
    assert (lo0, hi0) == (0, 249) and (lo3, hi3) == (750, 999)


def test_pin_ex1_cpu_defaults(orc, pins):
    """examples/refOutput/ex1.txt:27 -- 6 iterations, final 4.98e-08."""
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    amg = orc.Amg(A, orc.amg_params(False))
    r = orc.pcg(A, b, amg)
    ref = pins["ex1"]["stats"][0]
    assert r["converged"] and r["iters"] == ref["iters"]
    S = A.to_scipy()
    true_rel = np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b)
    # hypre's list/hash orders on coarse levels are upstream detail: 2 % on the 6-iteration residual
    assert true_rel == pytest.approx(ref["rel"], rel=0.02)


def test_pin_laplacian_driver(orc, pins):
    """examples/refOutput/laplacian.txt:34-38 -- 5 iterations, 6.12e-07."""
    A, b = orc.lap7(10, 10, 10, b_mode=0)
    amg = orc.Amg(A, orc.amg_params(False))
    r = orc.pcg(A, b, amg)
    ref = pins["laplacian"]["stats"][0]
    assert r["converged"] and r["iters"] == ref["iters"]
    S = A.to_scipy()
    true_rel = np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b)
    assert true_rel == pytest.approx(ref["rel"], rel=0.02)


def test_pin_ex2_pmis_hierarchy(orc, pins):
    """examples/refOutput/ex2.txt:124-139: PMIS/ext+i hierarchy 1000/351/62, nnz 6400/7485/1986,
    P rows 1..4 entries, complexities 1.413 / 2.48.  hypre's RNG stream (and the 4-rank
    partition) cannot be reproduced, so sizes are compared to a few percent."""
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    amg = orc.Amg(A, orc.amg_params(True))
    ref = pins["ex2"]
    assert amg.num_levels == len(ref["operators"])
    for l, op in enumerate(ref["operators"]):
        Al = amg.level_A(l)
        assert Al.nrows == pytest.approx(op["rows"], rel=0.12), (l, Al.nrows)
        assert Al.nnz == pytest.approx(op["nnz"], rel=0.15), (l, Al.nnz)
    assert amg.grid_complexity == pytest.approx(ref["grid_complexity"], rel=0.02)
    assert amg.operator_complexity == pytest.approx(ref["operator_complexity"], rel=0.05)
    for l, ip in enumerate(ref["interp"]):
        P = amg.level_P(l)
        cnt = np.diff(P.rowptr)
        assert cnt.max() == ip["max"]
        rs = np.asarray(P.to_scipy().sum(axis=1)).ravel()
        assert rs.max() <= 1.0 + 1e-12
    # <b,b> = 1000 (ex2.txt:160)
    assert float(b @ b) == ref["bdotb"]
    # the level-0 weight extremes of the reference's table are exact (ex2.txt:134: 5.263e-02 = 1/19 and 4.255e-01 = 20/47; hypre's
    # table leaves the identity entries of the C rows out): the extended+i weights and their truncation are the reference's
    w = amg.level_P(0).to_scipy().data
    w = w[w != 1.0]
    assert f"{w.min():.3e}" == "5.263e-02" and f"{w.max():.3e}" == "4.255e-01"


def test_hypre_pmis_stream_is_the_minimal_standard_generator(orc):
    """hypre_Rand = Park & Miller's minimal standard generator (seed <- 16807 seed mod 2^31 - 1): the published
    check value -- starting from 1 the 10 000th seed is 1 043 618 065 -- and the first draws of rank 0 (seed 2747) and rank 3."""
    m = 2147483647
    s = 1
    for _ in range(10000):
        s = (16807 * s) % m
    assert s == 1043618065
    r = orc.pmis_hypre_stream(8, part=[0, 3, 3, 5, 8])          # ranks of 3, 0, 2, 3 rows
    want = []
    for q, cnt in enumerate([3, 0, 2, 3]):
        s = 2747 + q
        for _ in range(cnt):
            s = (16807 * s) % m
            want.append(s / m)
    assert np.array_equal(r, np.array(want))
    assert r[0] == (16807 * 2747) / m


def test_pin_ex2_with_hypres_own_random_stream_is_a_negative_result(orc, pins):
    """Round 5 experiment (tools/pmis_rng_experiment.py, profiles/r05_pmis_rng_experiment.txt): PMIS with hypre's own tie-break
    stream -- per rank seed 2747 + rank, one draw per local row, every level anew, the ranks being row blocks -- does NOT reproduce
    examples/refOutput/ex2.txt:122-139 exactly on any partition of the 10^3 grid with 250 rows on the first rank (the Zenodo np4
    files are not in the reference tree, so their partition is one of the unknowns).  Closest: the generator's 2 x 2 x 1 rank grid:
    rows 360 / 61 (351 / 62), nonzeros 7478 / 1931 (7485 / 1986), operator complexity 2.470 (2.480), row-sum minimum 4.194e-01 exact --
    every figure within 3 %, where the partition-independent hash the product uses sits at 8-10 % on the coarsest level.  The split it
    produces is a valid PMIS split either way."""
    P = (2, 2, 1)
    A, _ = orc.lap7(10, 10, 10, P=P, b_mode=1)
    part = [orc.lap7_partition(10, 10, 10, P, r)[0] for r in range(4)] + [1000]
    amg = orc.Amg(A, orc.amg_params(True, blocks=4, block_part=part, pmis_rng=1))
    ref = pins["ex2"]
    assert amg.num_levels == len(ref["operators"])
    got = [(amg.level_A(l).nrows, amg.level_A(l).nnz) for l in range(3)]
    assert got == [(1000, 6400), (360, 7478), (61, 1931)]                       # not 351 / 7485 and 62 / 1986
    for (n, nnz), op in zip(got, ref["operators"]):
        assert n == pytest.approx(op["rows"], rel=0.03) and nnz == pytest.approx(op["nnz"], rel=0.03)
    assert amg.operator_complexity == pytest.approx(ref["operator_complexity"], rel=0.005)
    rs = np.asarray(amg.level_P(0).to_scipy().sum(axis=1)).ravel()
    assert f"{rs.min():.3e}" == "4.194e-01"                                     # ex2.txt:134, exact
    # a valid split: C points independent, every F point has a strong C neighbour
    sm = orc.strength(A)
    cf = amg.level_cf(0)
    S = A.to_scipy()
    S.data = sm.astype(float)
    S.eliminate_zeros()
    G = ((S + S.T) > 0).tocsr()
    C = cf == 1
    assert G[C][:, C].nnz == 0 and np.all((S[cf == -1][:, C].getnnz(axis=1)) > 0)


def test_pmis_is_valid_mis(orc):
    A, _ = orc.lap7(12, 9, 7)
    sm = orc.strength(A)
    cf = orc.pmis(A, sm)
    S = A.to_scipy()
    S.data = sm.astype(float)
    S.eliminate_zeros()
    G = ((S + S.T) > 0).tocsr()
    C = cf == 1
    # independence: no two C points strongly connected; maximality: every F has a strong C
    assert G[C][:, C].nnz == 0
    dep = S.tocsr()
    for i in np.where(cf == -1)[0]:
        nb = dep.indices[dep.indptr[i]:dep.indptr[i + 1]]
        assert C[nb].any()
    assert set(np.unique(cf)) <= {1, -1, -3}


def test_rap_matches_scipy(orc):
    A, _ = orc.lap7(9, 8, 7)
    sm = orc.strength(A)
    cf = orc.pmis(A, sm)
    P = orc.interp_extpi(A, sm, cf)
    Ac = orc.rap(A, P)
    Ps, As = P.to_scipy(), A.to_scipy()
    ref = (Ps.T @ As @ Ps).tocsr()
    d = (Ac.to_scipy() - ref)
    assert abs(d).max() < 1e-13
    rp, cj = Ac.rowptr, Ac.col
    for i in range(Ac.nrows):  # rows column-sorted, no duplicates
        assert np.all(np.diff(cj[rp[i]:rp[i + 1]]) > 0)


def test_interp_partition_of_unity_interior(orc):
    """Interior rows of a zero-row-sum operator interpolate constants exactly."""
    A, _ = orc.lap7(10, 10, 10)
    sm = orc.strength(A)
    cf = orc.pmis(A, sm)
    P = orc.interp_extpi(A, sm, cf).to_scipy()
    rs = np.asarray(P.sum(axis=1)).ravel()
    S = A.to_scipy()
    interior = np.asarray(S.sum(axis=1)).ravel() == 0.0
    fpts = (cf == -1) & interior
    assert fpts.sum() > 50
    assert np.allclose(rs[fpts], 1.0, atol=1e-12)
    assert np.all(rs[cf == 1] == 1.0)


def test_pcg_matches_scipy_solution(orc):
    A, b = orc.lap7(8, 8, 8, b_mode=1)
    r = orc.pcg(A, b, None, orc.krylov_params(False, rtol=1e-12, max_iter=500))
    x = spla.spsolve(A.to_scipy().tocsc(), b)
    assert r["converged"]
    assert np.linalg.norm(r["x"] - x) / np.linalg.norm(x) < 1e-10


def test_gmres_amg_converges(orc):
    A, b = orc.lap7(10, 10, 10, b_mode=0)
    amg = orc.Amg(A, orc.amg_params(True))
    r = orc.gmres(A, b, amg)
    S = A.to_scipy()
    assert r["converged"]
    assert np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b) < 1e-6
    assert r["iters"] <= 12


def test_unit_anchor_one_by_one(orc, pins):
    """tests/test_setmatrix_from_csr.c:397-417: 3x = 6 -> ||x|| = 2."""
    u = pins["unit"]["one_by_one"]
    A = orc.Csr.from_arrays(1, 1, [0, 1], [0], [u["a"]])
    amg = orc.Amg(A, orc.amg_params(True))
    assert amg.num_levels == 1
    r = orc.pcg(A, np.array([u["b"]]), amg)
    assert abs(np.linalg.norm(r["x"]) - u["x_norm"]) < u["tol"]


def test_unit_anchor_1d_laplacian(orc):
    """tests/test_setmatrix_from_csr.c:168-199: 1-D Laplacian n=16 solves."""
    n = 16
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(n, n), format="csr")
    A = orc.Csr.from_scipy(T)
    b = np.ones(n)
    r = orc.pcg(A, b, orc.Amg(A, orc.amg_params(True)))
    assert r["converged"] and np.linalg.norm(r["x"]) > 0
    assert np.allclose(T @ r["x"], b, atol=1e-5)


def test_zero_rhs_returns_zero(orc):
    A, _ = orc.lap7(5, 5, 5)
    r = orc.pcg(A, np.zeros(A.nrows), None)
    assert r["iters"] == 0 and np.all(r["x"] == 0)


def test_init_guess_exact_zero_iters(orc):
    """tests/test_init_guess.c:247-270: x0 = ones on b = A*1 -> residual 0."""
    A, _ = orc.lap7(6, 6, 6)
    S = A.to_scipy()
    b = S @ np.ones(A.nrows)
    r = orc.pcg(A, b, None, x0=np.ones(A.nrows))
    assert r["hist"][0] == 0.0 or r["iters"] <= 1


# ------------------------------------------------------------------ ILU(0) (parity unpinned: analytic anchors only)

def _lu_split(F):
    import scipy.sparse as sp
    LU = F.factors.to_scipy()
    n = LU.shape[0]
    return (sp.tril(LU, -1) + sp.eye(n)).tocsr(), sp.triu(LU).tocsr()


def test_ilu0_defining_property(orc):
    """ILU(0): (L U)_ij = a_ij on the pattern of A, nothing stored outside it (Saad, Prop. 10.4)."""
    import scipy.sparse as sp
    for A in (orc.lap7(7, 6, 5, b_mode=1)[0],
              orc.Csr.from_scipy((sp.random(120, 120, density=0.06, random_state=3, format="csr")
                                  + sp.diags(np.full(120, 4.0))).tocsr())):  # second one: non-symmetric pattern
        S = A.to_scipy()
        F = orc.Ilu(A)
        L, U = _lu_split(F)
        assert (F.factors.to_scipy() != 0).nnz <= S.nnz
        E = (L @ U - S).tocsr()
        pat = (abs(S) > 0).astype(float)
        assert abs(E.multiply(pat)).max() < 1e-13 * abs(S).max()


def test_ilu0_tridiagonal_is_exact_lu(orc):
    """No fill on a tridiagonal matrix: ILU(0) = LU, so the preconditioned solve ends after one iteration."""
    import scipy.sparse as sp
    n = 50
    T = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    A = orc.Csr.from_scipy(T)
    b = np.arange(1.0, n + 1.0)
    F = orc.Ilu(A)
    assert np.allclose(T @ F.apply(b), b, rtol=0, atol=1e-11)
    k = np.arange(2, n + 1)  # pivots of the 1-D Laplacian: u_kk = (k + 1) / k, k = 1..n
    U = _lu_split(F)[1]
    assert np.allclose(U.diagonal()[1:], (k + 1.0) / k)
    r = orc.pcg(A, b, orc.IluPrecond(A))
    assert r["converged"] and r["iters"] == 1


def test_ilu_apply_variants(orc):
    """exact substitutions = scipy's triangular solves; the Jacobi-iterative form (ilu.c:21-23) converges to it
    (strictly triangular iteration matrices are nilpotent) and block Jacobi = ILU of the block diagonal."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    A, b = orc.lap7(6, 6, 6, b_mode=0)
    b = b + np.linspace(0.0, 1.0, A.nrows)
    F = orc.Ilu(A)
    L, U = _lu_split(F)
    z = F.apply(b)
    z_ref = spla.spsolve_triangular(U, spla.spsolve_triangular(L, b, lower=True), lower=False)
    assert np.allclose(z, z_ref, rtol=1e-13, atol=0)
    errs = [np.linalg.norm(orc.Ilu(A, tri_solve=0, lower_it=k, upper_it=k).apply(b) - z) for k in (1, 3, 5, 9, 40)]
    assert all(e1 > e2 for e1, e2 in zip(errs, errs[1:-1])) and errs[-1] < 1e-12 * np.linalg.norm(z)
    part = [0, 70, 150, A.nrows]
    S = A.to_scipy().tolil()
    for p in range(3):  # cut the couplings between blocks
        lo, hi = part[p], part[p + 1]
        S[lo:hi, :lo] = 0
        S[lo:hi, hi:] = 0
    Fb = orc.Ilu(A, part=part)
    Fd = orc.Ilu(orc.Csr.from_scipy(S.tocsr()))
    assert np.array_equal(Fb.apply(b), Fd.apply(b))


def test_ilu_as_preconditioner_and_smoother(orc):
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    plain = orc.pcg(A, b)
    ilu = orc.pcg(A, b, orc.IluPrecond(A))
    assert ilu["converged"] and ilu["iters"] < plain["iters"]
    two = orc.gmres(A, b, orc.IluPrecond(A, max_iter=2))
    one = orc.gmres(A, b, orc.IluPrecond(A, max_iter=1))
    assert two["converged"] and two["iters"] < one["iters"]
    amg = orc.Amg(A, orc.amg_params(True))
    base = orc.pcg(A, b, amg)
    amg.set_ilu_smoother(num_levels=1, num_sweeps=1)
    sm = orc.pcg(A, b, amg)
    assert sm["converged"] and sm["iters"] < base["iters"]
    S = A.to_scipy()
    assert np.linalg.norm(b - S @ sm["x"]) / np.linalg.norm(b) < 1e-6


def test_ilu_rejects_missing_diagonal(orc):
    import scipy.sparse as sp
    M = sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 2.0]]))
    M.eliminate_zeros()
    with pytest.raises(ValueError):
        orc.Ilu(orc.Csr.from_scipy(M))


def test_fgmres_and_bicgstab_restatements(orc):
    """FlexGMRES = GMRES in exact arithmetic for a fixed preconditioner; BiCGSTAB converges to the same solution
    and, unpreconditioned on an SPD matrix, needs about half of CG's iterations (two products each)."""
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    S = A.to_scipy()
    amg = orc.Amg(A, orc.amg_params(True))
    g, f = orc.gmres(A, b, amg), orc.fgmres(A, b, amg)
    assert f["converged"] and f["iters"] == g["iters"] and np.allclose(f["hist"], g["hist"], rtol=1e-9)
    kp = orc.krylov_params(False, max_iter=400, rtol=1e-8)
    bs = orc.bicgstab(A, b, None, kp)
    cg = orc.pcg(A, b, None, kp)
    assert bs["converged"] and 0.3 * cg["iters"] <= bs["iters"] <= cg["iters"]
    assert np.linalg.norm(b - S @ bs["x"]) / np.linalg.norm(b) <= 1e-8
    ba = orc.bicgstab(A, b, amg)
    assert ba["converged"] and ba["iters"] < orc.pcg(A, b, amg)["iters"]
    assert orc.bicgstab(A, np.zeros(A.nrows))["iters"] == 0


# ------------------------------------------------------------------ MGR (parity unpinned: analytic anchors only)

def three_field_system(n=10, seed=0):
    """Interleaved 3-field model of a reservoir-like system (labels 0 = pressure-like, 1, 2 = local fields)."""
    import scipy.sparse as sp
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


def test_mgr_transfer_operators_and_schur_complement(orc):
    """P = [W; I], R = [Z I], A_c = R A P.  With a diagonal A_FF, jacobi prolongation and injection restriction give
    the exact Schur complement; jacobi restriction gives the same; columped divides by the column sums of A_FF."""
    import scipy.sparse as sp
    S, labels = three_field_system()
    A = orc.Csr.from_scipy(S)
    F, Cm = labels == 2, labels != 2
    Aff, Afc, Acf, Acc = S[F][:, F], S[F][:, Cm], S[Cm][:, F], S[Cm][:, Cm]
    schur = (Acc - Acf @ sp.diags(1.0 / Aff.diagonal()) @ Afc).tocsr()
    for rt in ("injection", "jacobi", "columped"):
        M = orc.MgrPrecond(A, labels, [dict(f_dofs=[2], prolongation_type="jacobi", restriction_type=rt)])
        P, R = M.matrix(0, 1).to_scipy(), M.matrix(0, 2).to_scipy()
        assert abs(P[Cm] - sp.identity(Cm.sum())).max() == 0 and abs(R[:, Cm] - sp.identity(Cm.sum())).max() == 0
        assert abs(P[F] + sp.diags(1.0 / Aff.diagonal()) @ Afc).max() < 1e-15
        assert abs(M.matrix(1, 0).to_scipy() - schur).max() < 1e-13   # A_FF diagonal: every variant reduces exactly
        if rt == "columped":
            cs = np.asarray(Aff.sum(axis=0)).ravel()
            assert abs(R[:, F] + Acf @ sp.diags(1.0 / cs)).max() < 1e-15
    M0 = orc.MgrPrecond(A, labels, [dict(f_dofs=[2])])                 # injection both ways: A_c = A_CC
    assert abs(M0.matrix(1, 0).to_scipy() - Acc).max() == 0


def test_mgr_two_reduction_levels_as_preconditioner(orc):
    """The structure of the reference's examples/ex3.yml (level 0: f_dofs [2], jacobi prolongation; level 1: f_dofs [1],
    l1-hsgs global relaxation, columped restriction; coarsest AMG) under GMRES."""
    S, labels = three_field_system(12, seed=1)
    A = orc.Csr.from_scipy(S)
    b = np.ones(S.shape[0])
    lev = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")]
    M = orc.MgrPrecond(A, labels, lev)
    assert [M.matrix(l, 0).nrows for l in range(3)] == [432, 288, 144]
    r = orc.gmres(A, b, M)
    plain = orc.gmres(A, b, None)
    assert r["converged"] and r["iters"] < plain["iters"] / 2
    assert np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b) < 1e-6
    assert orc.fgmres(A, b, M)["iters"] == r["iters"]


# ------------------------------------------------------------------ Chebyshev smoother (relax type 16; parity unpinned)

def test_cheby_polynomial_and_eigenvalue_estimate(orc):
    """The coefficients are those of the residual polynomial T_k((theta - t)/delta) / T_k(theta/delta); the CG / Lanczos
    estimate approaches the largest eigenvalue of D^-1/2 A D^-1/2 from below, Gershgorin bounds it from above; one sweep
    is the matrix polynomial applied to the residual."""
    import scipy.sparse as sp
    from numpy.polynomial.chebyshev import Chebyshev
    A, b = orc.lap7(9, 8, 7, b_mode=1)
    S = A.to_scipy()
    D = sp.diags(1.0 / np.sqrt(S.diagonal()))
    lam = np.linalg.eigvalsh((D @ S @ D).toarray())
    for order in (1, 2, 3, 4):
        Cb = orc.Cheby(A, order=order, fraction=0.3)
        assert 0.9 * lam[-1] < Cb.max_eig <= lam[-1] * (1 + 1e-12) and Cb.min_eig >= lam[0] * (1 - 1e-12)
        upper = 1.1 * Cb.max_eig
        lower = Cb.min_eig + 0.3 * (upper - Cb.min_eig)
        th, de = 0.5 * (upper + lower), 0.5 * (upper - lower)
        t = np.linspace(lower, upper, 9)
        q = sum(Cb.coefs[i] * t ** i for i in range(order))
        T = Chebyshev.basis(order)
        assert np.abs((1.0 - t * q) - T((th - t) / de) / T(th / de)).max() < 1e-13
        # the sweep: u1 = u0 + D^-1/2 q(D^-1/2 A D^-1/2) D^-1/2 (b - A u0)
        u0 = np.linspace(0.0, 1.0, A.nrows)
        Ss = (D @ S @ D).toarray()
        Q = sum(Cb.coefs[i] * np.linalg.matrix_power(Ss, i) for i in range(order))
        ref = u0 + D @ (Q @ (D @ (b - S @ u0)))
        assert np.allclose(Cb.apply(b, u0), ref, rtol=1e-12, atol=1e-13)
    G = orc.Cheby(A, eig_est=0)
    assert G.max_eig >= lam[-1]
    plain = orc.pcg(A, b, orc.Amg(A, orc.amg_params(True)))
    cheb = orc.pcg(A, b, orc.Amg(A, orc.amg_params(True, relax_down=16, relax_up=16)))
    assert cheb["converged"] and cheb["iters"] < plain["iters"]


# (interp_type 17: mm-ext+i as examples/ex8.yml names it -- since round 4 the matrix-matrix operator itself, not the classical formula)
EX8_VARIANTS = [dict(coarsen_type=10, interp_type=17, strong_th=0.25, relax_down=16, relax_up=16),
                dict(coarsen_type=10, interp_type=17, strong_th=0.5, relax_down=16, relax_up=16, cheby_order=4, cheby_fraction=0.1),
                dict(coarsen_type=10, interp_type=17, strong_th=0.8, relax_down=8, relax_up=8),
                dict(coarsen_type=10, interp_type=17, strong_th=0.9, relax_down=16, relax_up=16)]


def ex8_system(orc):
    """The system behind examples/refOutput/ex8.txt: the 10^3 operator with the right-hand side of the first np4 part only
    (its initial residual 1.58e+01 = sqrt(250); ex2 reads the same files on 4 ranks and reports sqrt(1000))."""
    A, _ = orc.lap7(10, 10, 10, b_mode=1)
    return A, np.r_[np.ones(250), np.zeros(750)]


def test_pin_ex8_chebyshev_l1symgs_ilu_variants(orc, pins):
    """examples/refOutput/ex8.txt:92-95 -- the reference's only checked-in numbers for Chebyshev relaxation (type 16), the
    symmetric l1 Gauss-Seidel sweep (l1sym-hgs) and the ILU(0) complex smoother, all with mm-ext+i interpolation (the
    matrix-matrix form of extended+i) on HMIS grids, PCG to 1e-9: 7 / 6 / 6 / 7 iterations.  Tolerance SURVEY 8(c): +-1
    (hypre's coarse grids come from its own random stream, its Chebyshev eigenvalue estimate from a random start vector).
    The fifth variant of that output used standard interpolation, which is not restated."""
    A, b = ex8_system(orc)
    ref = pins["ex8"]["stats"]
    assert np.linalg.norm(b) == pytest.approx(ref[0]["r0"], rel=2e-3)
    S = A.to_scipy()
    got = []
    for k, v in enumerate(EX8_VARIANTS):
        amg = orc.Amg(A, orc.amg_params(False, **v))
        if k == 3:
            amg.set_ilu_smoother(1, 1)
        r = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
        assert r["converged"]
        assert np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b) < 1e-9
        assert abs(r["iters"] - ref[k]["iters"]) <= 1, (k, r["iters"], ref[k]["iters"])
        got.append(r["iters"])
    assert got == [6, 5, 6, 6]  # the oracle's own counts, so that a change of the restatement shows up here


def test_pin_ex8_on_four_row_blocks(orc, pins):
    """The same four pins with HMIS and the sweeps on FOUR row blocks (orc_amg_params.blocks = 4: what hypre computes on the four
    part files of data/ps3d10pt7/np4 the reference's ex8 run read): 7 / 5 / 7 / 6 iterations against 7 / 6 / 6 / 7, every one within
    the +-1 of SURVEY 8(c), and the per-iteration rates (final true residual)^(1/iterations) move from the one-block restatement's
    0.030 / 0.010 / 0.022 / 0.025 to 0.040 / 0.014 / 0.033 / 0.030 against the reference's 0.046 / 0.016 / 0.031 / 0.031 -- closer in
    all four: the first coarse grid has 436 points (PMIS on the rows next to a block boundary) where one block gives the red-black
    500.  DESIGN section 3 left this as the open explanation of round 3's systematic -1."""
    A, b = ex8_system(orc)
    ref = pins["ex8"]["stats"]
    S = A.to_scipy()
    got, rates = [], []
    for k, v in enumerate(EX8_VARIANTS):
        amg = orc.Amg(A, orc.amg_params(False, blocks=4, **v))
        assert amg.level_A(1).nrows == 436
        if k == 3:
            amg.set_ilu_smoother(1, 1)  # (one block: that run was ONE rank reading four part files -- its r0 is sqrt(250), the first part's right-hand side)
        r = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
        tr = np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b)
        assert r["converged"] and tr < 1e-9 and abs(r["iters"] - ref[k]["iters"]) <= 1, (k, r["iters"], ref[k]["iters"])
        got.append(r["iters"])
        rates.append(tr ** (1.0 / r["iters"]))
    assert got == [7, 5, 7, 6]
    want = [ref[k]["rel"] ** (1.0 / ref[k]["iters"]) for k in range(4)]
    assert all(abs(g - w) / w < 0.2 for g, w in zip(rates, want)), (rates, want)


def test_mm_extpi_properties(orc):
    """mm-ext+i (type 17), W = -D^-1 (I + B) A^s_FC: rows of a constant-coefficient M-matrix interior sum to 1 like extended+i's, C rows
    inject, the pattern is the distance-two interpolatory set, and where no F point has a strong F neighbour it IS direct interpolation
    with the weak entries lumped (B = 0)."""
    import scipy.sparse as sp
    A, _ = orc.lap7(9, 8, 7)
    sm = orc.strength(A, 0.25)
    cf = orc.hmis_blocks(A, sm, [0, A.nrows])
    P17, P6 = orc.interp_mm_extpi(A, sm, cf, 0, 0.0), orc.interp_extpi(A, sm, cf, 0, 0.0)
    S17, S6 = P17.to_scipy(), P6.to_scipy()
    assert S17.shape == S6.shape
    crow = np.flatnonzero(cf == 1)
    assert np.allclose(S17[crow].toarray(), S6[crow].toarray())          # injection
    inner = [i for i in np.flatnonzero(cf == -1) if A.rowptr[i + 1] - A.rowptr[i] == 7]
    assert np.allclose(np.asarray(S17[inner].sum(axis=1)).ravel(), 1.0, atol=1e-12)
    assert set(zip(*S17.nonzero())) <= set(zip(*S6.nonzero()))           # inside extended+i's distance-two set
    # red-black grid: every neighbour of an F point is C -> B = 0 -> W = -A_FC / a_ii (all connections strong here)
    M = A.to_scipy().tocsr()
    for i in inner[:50]:
        cols = S17[i].indices
        fine = [j for j in M[i].indices if cf[j] == 1]
        assert len(cols) == len(fine) and np.allclose(S17[i].data, 1.0 / 6.0)


def test_pin_ex8_fifth_variant_direct_interpolation(orc, pins):
    """examples/refOutput/ex8.txt:96 -- the fifth variant: PMIS 0.5, two l1 symmetric Gauss-Seidel sweeps (type 8) down and
    up, 6 iterations.  That output echoes `prolongation_type: standard` (ex8.txt:74) while the examples/ex8.yml in the tree
    says `direct_sep_weights` (ex8.yml:76): the pin constrains the restated direct interpolation (type 3) only as far as the
    two operators agree on this grid -- SURVEY 8(c)'s +-1."""
    A, b = ex8_system(orc)
    ref = pins["ex8"]["stats"][4]
    amg = orc.Amg(A, orc.amg_params(False, coarsen_type=8, interp_type=3, strong_th=0.5, relax_down=8, relax_up=8, sweeps_down=2, sweeps_up=2))
    r = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
    assert r["converged"] and abs(r["iters"] - ref["iters"]) <= 1, (r["iters"], ref["iters"])
    assert r["iters"] == 6  # the oracle's own count
    # direct interpolation: an F row names strong C neighbours only, rows sum to ~1 on this M-matrix interior
    P = amg.level_P(0)
    cf = amg.level_cf(0)
    sm = orc.strength(A, 0.5, 0.9)
    S = A.to_scipy().tocsr()
    cidx = np.cumsum(cf == 1) - 1
    for i in np.flatnonzero(cf == -1)[:200]:
        strongC = {int(cidx[j]) for k, j in zip(range(S.indptr[i], S.indptr[i + 1]), S.indices[S.indptr[i]:S.indptr[i + 1]]) if sm[k] and cf[j] == 1}
        assert set(P.col[P.rowptr[i]:P.rowptr[i + 1]]) <= strongC



# ---- row blocks: V contiguous row blocks on one rank = the reference at np = V (orc_amg_params.blocks) -------------------------

def _mmat(n, density, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="csr")
    M = sp.csr_matrix(M + M.T)
    M.data = -np.abs(M.data)
    M.setdiag(0)
    M.eliminate_zeros()
    return (M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 0.1)).tocsr()


def test_block_forms_with_one_block_are_the_sequential_forms(orc):
    """blocks <= 1 must reproduce the one-rank restatement bit for bit (the ex1 / laplacian pins above run through it)"""
    M = _mmat(600, 0.01, 3)
    A = orc.Csr.from_scipy(M)
    rng = np.random.default_rng(0)
    b, x = rng.standard_normal(600), rng.standard_normal(600)
    for opt in (1, 4):
        assert np.array_equal(orc.l1_norms_blocks(A, opt, [0, 600]), orc.l1_norms(A, opt))
    for t in (3, 4, 6, 8, 13, 14, 18, 0):
        l1 = orc.l1_norms(A, 1 if t == 18 else 4)
        assert np.array_equal(orc.relax_blocks(A, l1, t, 0.9, b, x, [0, 600]), orc.relax(A, l1, t, 0.9, b, x))
    # a hierarchy with blocks = 1 is the hierarchy without the parameter
    Al, bl = orc.lap7(9, 8, 7)
    h0, h1 = orc.Amg(Al, orc.amg_params(False)), orc.Amg(Al, orc.amg_params(False, blocks=1))
    assert h0.num_levels == h1.num_levels
    for l in range(h0.num_levels - 1):
        assert np.array_equal(h0.level_cf(l), h1.level_cf(l))
    assert np.allclose(orc.pcg(Al, bl, h0)["hist"], orc.pcg(Al, bl, h1)["hist"], rtol=1e-12, atol=0)  # (threaded inner products)


def test_block_l1_divisor_formula(orc):
    """hypre_ParCSRComputeL1Norms option 4: |a_ii| + half the absolute sum of the entries leaving the block, truncated to |a_ii|
    when within 4/3 of it"""
    M = _mmat(300, 0.03, 5)
    A = orc.Csr.from_scipy(M)
    part = np.array([0, 70, 70, 190, 300])
    got = orc.l1_norms_blocks(A, 4, part)
    Mc = M.tocoo()
    blk = np.searchsorted(part, np.arange(300), side="right") - 1
    # (searchsorted with the empty block: rows 70..189 belong to block 2)
    off = np.zeros(300)
    for i, j, v in zip(Mc.row, Mc.col, Mc.data):
        if i != j and not (part[blk[i]] <= j < part[blk[i] + 1]):
            off[i] += abs(v)
    d = M.diagonal()
    want = d + 0.5 * off
    want[want <= 4.0 / 3.0 * d] = d[want <= 4.0 / 3.0 * d]
    assert np.allclose(got, want, rtol=1e-15)
    assert (got > d).any() and (got == d).any()  # both branches of the truncation are exercised


def test_block_gauss_seidel_limits(orc):
    """every row its own block = a Jacobi sweep with the option-4 divisor; blocks that do not couple = independent sequential
    sweeps; a sweep over V blocks never reads a value another block wrote in the same sweep"""
    M = _mmat(250, 0.04, 7)
    A = orc.Csr.from_scipy(M)
    rng = np.random.default_rng(2)
    b, x = rng.standard_normal(250), rng.standard_normal(250)
    rows = np.arange(251)
    l1 = orc.l1_norms_blocks(A, 4, rows)
    jac = x + (b - M @ x) / l1
    for t in (13, 14):
        assert np.allclose(orc.relax_blocks(A, l1, t, 1.0, b, x, rows), jac, rtol=1e-13, atol=1e-14)
    # block-diagonal operator: blocks = components, the hybrid sweep is the sequential sweep
    import scipy.sparse as sp
    B = sp.block_diag([_mmat(80, 0.1, 8), _mmat(50, 0.1, 9), _mmat(120, 0.05, 10)]).tocsr()
    Bo = orc.Csr.from_scipy(B)
    bb, xx = rng.standard_normal(250), rng.standard_normal(250)
    part = [0, 80, 130, 250]
    l1b = orc.l1_norms_blocks(Bo, 4, part)
    assert np.array_equal(l1b, orc.l1_norms(Bo, 4))
    for t in (3, 4, 6, 13, 14, 8):
        assert np.array_equal(orc.relax_blocks(Bo, l1b, t, 1.0, bb, xx, part), orc.relax(Bo, l1b, t, 1.0, bb, xx))
    # two blocks: the second block's forward sweep sees the OLD first block
    part2 = [0, 100, 250]
    l12 = orc.l1_norms_blocks(A, 4, part2)
    got = orc.relax_blocks(A, l12, 13, 1.0, b, x, part2)
    want = x.copy()
    Ml = M.tolil()
    for lo, hi in ((0, 100), (100, 250)):
        cur = x.copy()
        for i in range(lo, hi):
            r = b[i] - sum(v * (cur[j] if lo <= j < hi else x[j]) for j, v in zip(Ml.rows[i], Ml.data[i]))
            cur[i] += r / l12[i]
        want[lo:hi] = cur[lo:hi]
    assert np.allclose(got, want, rtol=1e-13, atol=1e-14)


def test_block_hmis_properties(orc):
    """HMIS on V blocks (De Sterck / Yang / Heys: Ruge first pass per block, interior C points kept, PMIS on the rest): components
    as blocks = the first pass of every component; every F point ends up with a strong C neighbour or without dependants; the C points
    are an independent set of the strength graph wherever PMIS placed them; thin blocks (every row on a block boundary) = PMIS"""
    import scipy.sparse as sp
    B = sp.block_diag([orc.lap7(6, 5, 4)[0].to_scipy(), orc.lap7(5, 5, 5)[0].to_scipy()]).tocsr()
    Bo = orc.Csr.from_scipy(B)
    sm = orc.strength(Bo, 0.25)
    cf = orc.hmis_blocks(Bo, sm, [0, 120, 245])
    a = orc.lap7(6, 5, 4)[0]
    c = orc.lap7(5, 5, 5)[0]
    assert np.array_equal(cf[:120], orc.hmis_blocks(a, orc.strength(a, 0.25), [0, 120]))
    assert np.array_equal(cf[120:], orc.hmis_blocks(c, orc.strength(c, 0.25), [0, 125]))
    # structure of the result on a coupled problem
    A, _ = orc.lap7(12, 10, 8)
    n = A.nrows
    sm = orc.strength(A, 0.25)
    S = sp.csr_matrix((sm.astype(float), A.col, A.rowptr), shape=(n, n))
    S.eliminate_zeros()
    for part in ([0, n], [0, n // 3, n], [(q * n) // 8 for q in range(9)]):
        cf = orc.hmis_blocks(A, sm, part)
        assert set(np.unique(cf)) <= {1, -1, -3}
        C = (cf == 1).astype(float)
        has_c = (S @ C) > 0
        dependants = np.asarray(S.sum(axis=0)).ravel() > 0
        assert np.all(has_c[cf == -1] | ~dependants[cf == -1])
    thin = [(q * n) // 80 for q in range(81)]  # blocks of one x-line ... 1.2 lines: every row has a strong neighbour outside
    assert np.array_equal(orc.hmis_blocks(A, sm, thin), orc.pmis(A, sm))


def test_block_hierarchy_partitions_follow_the_c_points(orc):
    A, b = orc.lap7(12, 12, 12)
    h = orc.Amg(A, orc.amg_params(False, blocks=6))
    p0 = h.level_block_part(0)
    assert np.array_equal(p0, orc.even_blocks(A.nrows, 6))
    for l in range(h.num_levels - 1):
        cf = h.level_cf(l)
        p, pn = h.level_block_part(l), h.level_block_part(l + 1)
        assert np.array_equal(pn, np.concatenate([[0], np.cumsum(cf == 1)])[p])
    r1, r6 = orc.pcg(A, b, orc.Amg(A, orc.amg_params(False))), orc.pcg(A, b, h)
    assert r6["converged"] and abs(r6["iters"] - r1["iters"]) <= 2


def test_pin_ex8_fifth_variant_standard_interpolation(orc, pins):
    """examples/refOutput/ex8.txt:63-79,96 -- the fifth variant AS THE OUTPUT ECHOES IT: PMIS 0.5, `prolongation_type: standard`
    (interpolation type 8), two l1 symmetric Gauss-Seidel sweeps down and up: 6 iterations, final relative residual 1.69e-11.
    The restated standard interpolation takes 5 (its fifth residual, 6.9e-10, is just under the 1e-9 the reference's fifth must
    have been just over) -- SURVEY 8(c)'s +-1 -- and its convergence rate per iteration, 0.0147, is the reference's
    (1.69e-11)^(1/6) = 0.0161 to 10 %; direct interpolation (type 3, what examples/ex8.yml names today) has 0.027."""
    A, b = ex8_system(orc)
    ref = pins["ex8"]["stats"][4]
    amg = orc.Amg(A, orc.amg_params(False, coarsen_type=8, interp_type=8, strong_th=0.5, relax_down=8, relax_up=8, sweeps_down=2, sweeps_up=2))
    r = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
    assert r["converged"] and abs(r["iters"] - ref["iters"]) <= 1, (r["iters"], ref["iters"])
    assert r["iters"] == 5  # the oracle's own count
    rate = (r["hist"][-1] / r["hist"][0]) ** (1.0 / r["iters"])
    ref_rate = ref["rel"] ** (1.0 / ref["iters"])  # 1.69e-11 after 6
    assert abs(rate / ref_rate - 1.0) < 0.15, (rate, ref_rate)


def test_standard_interp_properties(orc):
    """Standard interpolation restated from the paper: C rows are identity rows; an F row names its extended interpolatory set;
    on an M-matrix interior row (zero row sum) the weights sum to 1 (constants are interpolated exactly: sum of a-hat over all
    m != i equals -a-hat_ii there); and on a 1-D chain F - C - F the two-sided formula gives the known weights."""
    A, _ = orc.lap7(9, 9, 9)
    sm = orc.strength(A, 0.25, 0.9)
    cf = orc.pmis(A, sm)
    P = orc.interp_standard(A, sm, cf, 0, 0.0)
    S = A.to_scipy().tocsr()
    rows = np.diff(P.rowptr)
    assert np.all(rows[cf == 1] == 1) and np.all(P.val[P.rowptr[:-1][cf == 1]] == 1.0)
    interior = [i for i in np.flatnonzero(cf == -1) if S.indptr[i + 1] - S.indptr[i] == 7 and
                all(S.indptr[j + 1] - S.indptr[j] == 7 for j in S.indices[S.indptr[i]:S.indptr[i + 1]])]
    assert interior
    for i in interior[:100]:
        assert abs(P.val[P.rowptr[i]:P.rowptr[i + 1]].sum() - 1.0) < 1e-12
    # chain  C F F C  with rows (-1, 2, -1): F point 1 has strong C {0} and strong F {2} whose strong C is {3}
    import scipy.sparse as sp
    T = sp.diags([-np.ones(3), 2.0 * np.ones(4), -np.ones(3)], [-1, 0, 1]).tocsr()
    T.sort_indices()
    At = orc.Csr.from_scipy(T)
    smt = np.ones(T.nnz, np.uint8)
    smt[T.indices == np.repeat(np.arange(4), np.diff(T.indptr))] = 0
    Pt = orc.interp_standard(At, smt, np.array([1, -1, -1, 1], np.int32), 0, 0.0)
    # row 1: a-hat_11 = 2 - (-1)(-1)/2 = 1.5, a-hat_10 = -1, a-hat_13 = -(-1/2)(-1) = -0.5; all of the stencil is in C-hat: alfa = 1/1.5
    assert np.allclose(Pt.val[Pt.rowptr[1]:Pt.rowptr[2]], [1.0 / 1.5, 0.5 / 1.5])
    assert np.allclose(Pt.val[Pt.rowptr[2]:Pt.rowptr[3]], [0.5 / 1.5, 1.0 / 1.5])


def test_mgr_global_relaxation_on_row_blocks(orc):
    """orc_mgr_level_params.grelax_blocks (round 5): the hybrid Gauss-Seidel global relaxation of an MGR level on V row blocks = the
    reference at np = V.  One block is the sequential sweep bit for bit; with every row a block of its own the sweep is the Jacobi sweep
    with hypre's option-4 l1 divisor (what orc_relax_blocks is anchored on), a different -- still convergent -- preconditioner."""
    S, labels = three_field_system(9, seed=2)
    A = orc.Csr.from_scipy(S)
    b = np.ones(S.shape[0])
    base = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")]
    r = np.random.default_rng(3).standard_normal(S.shape[0])
    z0 = orc.MgrPrecond(A, labels, base).vcycle(r)
    one = [dict(base[0]), dict(base[1], g_blocks=1)]
    assert np.array_equal(orc.MgrPrecond(A, labels, one).vcycle(r), z0)
    nrows1 = int((np.asarray(labels) != 2).sum())   # rows of the second reduction level
    for V in (3, nrows1):
        lev = [dict(base[0]), dict(base[1], g_blocks=V)]
        M = orc.MgrPrecond(A, labels, lev)
        assert not np.array_equal(M.vcycle(r), z0)
        res = orc.gmres(A, b, M)
        assert res["converged"] and res["iters"] <= 3 * orc.gmres(A, b, orc.MgrPrecond(A, labels, base))["iters"]
