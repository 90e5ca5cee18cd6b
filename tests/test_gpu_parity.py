"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same inputs.

Bars: bit-exact for integer/index work (strength mask, C/F splitting, sparsity patterns) and
for the setup arithmetic that is defined in a fixed order (interpolation weights, RAP, l1
norms); 1e-13 relative for wave-parallel fp64 reductions (SpMV, smoothers, V-cycle);
identical iteration counts and residual histories to 1e-10 relative for AMG-PCG on an
identical hierarchy.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

RTOL_REDUCE = 1e-13


@pytest.fixture(scope="module")
def hd():
    import hypredrive_amd as h
    assert h.device_count() >= 1, "no HIP device"
    return h


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def rand_spd(n, density, seed):
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="csr")
    M = M + M.T
    M = M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 1.0)
    return M.tocsr()


def both(orc, hd, M):
    return orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)


# ---------------------------------------------------------------- K1 SpMV

@pytest.mark.parametrize("shape", [(10, 10, 10), (33, 17, 9), (64, 64, 64)])
def test_spmv_lap7(orc, hd, shape):
    Ao, _ = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    n = Ao.nrows
    # generator parity: identical CSR
    rp, cj, v = Ah.download()
    assert np.array_equal(rp, Ao.rowptr) and np.array_equal(cj, Ao.col) and np.array_equal(v, Ao.val)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(n)
    assert rel(Ah.spmv(x), orc.spmv(Ao, x)) < RTOL_REDUCE
    assert rel(Ah.spmv(x, -1.0, 1.0, y0), orc.spmv(Ao, x, -1.0, 1.0, y0)) < RTOL_REDUCE


@pytest.mark.parametrize("n,density", [(1, 1.0), (7, 0.5), (300, 0.02), (2000, 0.01), (500, 0.2)])
def test_spmv_irregular(orc, hd, n, density):
    """ragged rows, empty rows, all lane-group widths"""
    M = rand_spd(n, density, 3)
    M = M.tolil()
    if n > 5:
        M[3, :] = 0  # an empty row
    M = M.tocsr()
    M.eliminate_zeros()
    Ao, Ah = both(orc, hd, M)
    x = np.random.default_rng(2).standard_normal(n)
    assert rel(Ah.spmv(x), M @ x) < 1e-12
    assert rel(Ah.spmv(x), orc.spmv(Ao, x)) < 1e-12


@pytest.mark.parametrize("n,density,seed", [(3000, 0.01, 5), (900, 0.2, 6), (20000, 0.0008, 7)])
def test_windowed_csr_matches_plain_csr(hd, monkeypatch, n, density, seed):
    """The windowed form of the column indices (distinct columns of a 1024-entry chunk gathered once into LDS, 2-byte positions
    per entry, next chunk requested one chunk ahead) performs the plain streaming kernel's products; a row sum is reduced by the
    lane group its chunk assigns (chunks are half as long as the plain kernel's, so the grouping -- not the set -- of additions
    can differ): products, residuals and Jacobi sweeps agree to the 1e-13 of every wave-parallel reduction here, on irregular
    rows (an empty row, a 900-entry row that spans chunks, ragged lengths)."""
    rng = np.random.default_rng(seed)
    M = rand_spd(n, density, seed).tolil()
    M[7, :] = 0
    M[11, rng.choice(n, size=min(900, n - 1), replace=False)] = 0.5  # a long row
    M = M.tocsr()
    M.eliminate_zeros()
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    out = {}
    for form in ("plain", "windowed"):
        monkeypatch.setenv("HDA_WINDOW", "0" if form == "plain" else "1")
        monkeypatch.setenv("HDA_WINDOW_MIN_NNZ", "0")
        monkeypatch.setenv("HDA_WINDOW_RATIO", "2")  # take the form whatever it saves
        A = hd.Csr.from_scipy(M)
        out[form] = (A.spmv(x), A.spmv(x, -1.0, 1.0, b), A.relax(b, x, 18, 1.0, sweeps=2), hd.format_bytes(A)["spmv"])
    for a, c in zip(out["plain"][:3], out["windowed"][:3]):
        assert rel(c, a) < RTOL_REDUCE
    assert rel(out["windowed"][0], M @ x) < 1e-12
    assert out["windowed"][3] != out["plain"][3]  # the second run did take the windowed form


@pytest.mark.parametrize("shape,c", [((64, 64, 64), (1.0, 1.0, 1.0)), ((80, 60, 56), (1.0, 0.5, 0.001))])
def test_row_class_coding_is_bit_identical(hd, monkeypatch, shape, c):
    """A stencil-coded operator read one class byte per row (row-class coding) performs the products and additions of the
    entry-coded kernel -- and of the CSR row -- in the same order: bit-identical products, residuals and Jacobi sweeps, for
    the isotropic benchmark operator and an anisotropic one (three distinct off-diagonal values, 27 boundary classes)."""
    n = shape[0] * shape[1] * shape[2]
    rng = np.random.default_rng(11)
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    out = {}
    for form in ("entry", "row"):
        monkeypatch.setenv("HDA_ROWCLASS", "0" if form == "entry" else "1")
        A = hd.lap7(*shape, c=c)
        out[form] = (A.spmv(x), A.spmv(x, -1.0, 1.0, b), A.relax(b, x, 18, 1.0, sweeps=2), hd.format_bytes(A)["spmv"])
    for a, q in zip(out["entry"][:3], out["row"][:3]):
        assert np.array_equal(a, q)
    assert out["row"][3] < 0.7 * out["entry"][3]  # and it does stream fewer bytes
    S = hd.lap7(*shape, c=c).to_scipy()
    assert rel(out["row"][0], S @ x) < 1e-13


@pytest.mark.parametrize("sweeps", [(1, 1), (2, 1), (1, 2)])
def test_pcg_kernel_fusions_are_bitwise_neutral(hd, monkeypatch, sweeps):
    """Two launch fusions of the PCG loop (round 3): (a) on one rank the finalize kernels of <s,p> and of the <r,z>, <r,r> pair ride on
    the kernels that consume the sums (each block repeats k_finalize's arithmetic); (b) the cycle's zero-guess first sweep
    z0 = dinv .* r is written by the update kernel that produces r (Amg::apply_offering / FirstSweepFusion), into the vector the
    cycle would have put it -- the output vector or the level's second buffer, depending on the parity of the sweep count.  Same
    multiplications and additions: iterates, histories and iteration counts are bit-identical with either fusion off."""
    n = 40
    A = hd.lap7(n, n, n)
    b = np.random.default_rng(5).standard_normal(n ** 3)
    out = {}
    for fin, z0 in (("1", "1"), ("2", "1"), ("0", "1"), ("2", "0"), ("0", "0")):
        monkeypatch.setenv("HDA_FUSE_FINALIZE", fin)
        monkeypatch.setenv("HDA_FUSE_Z0", z0)
        amg = hd.Amg(A, hd.AmgParams.default(sweeps_down=sweeps[0], sweeps_up=sweeps[1]))
        out[(fin, z0)] = hd.pcg(A, b, amg, hd.KrylovParams.default(False, rtol=1e-10))
    ref = out[("0", "0")]
    assert ref["converged"]
    for k, r in out.items():
        assert r["iters"] == ref["iters"] and np.array_equal(r["hist"], ref["hist"]) and np.array_equal(r["x"], ref["x"]), k


@pytest.mark.parametrize("coded", ["1", "0"])
def test_first_sweep_fusion_on_the_windowed_restriction_is_bitwise_neutral(hd, monkeypatch, coded):
    """The zero-guess Jacobi sweep u = dinv .* f of a coarse level rides on the restriction kernel that produces f (spmv_with_scaled_copy);
    since round 5 also when the restriction runs on the WINDOWED kernel -- the level-0 restriction of a large problem (value-coded with
    HDA_CODED=1, plain values without): at 96^3 R0 has 1.4 M entries, above the lane-group and window thresholds.  One multiplication per
    row either way: V-cycles, PCG iterates and histories are bit-identical with HDA_FUSE_FIRST_SWEEP=0 (a separate k_mul launch).
    (profiles/r05_cycle_tail.md: what else was tried on the small-kernel tail of the cycle, and why it was removed.)"""
    monkeypatch.setenv("HDA_CODED", coded)
    n = 96
    A = hd.lap7(n, n, n)
    b = np.random.default_rng(9).standard_normal(n ** 3)
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("HDA_FUSE_FIRST_SWEEP", fuse)
        amg = hd.Amg(A)
        out[fuse] = (amg.vcycle(b), hd.pcg(A, b, amg, hd.KrylovParams.default(False, rtol=1e-10)))
    (z1, r1), (z0, r0) = out["1"], out["0"]
    assert r0["converged"] and np.array_equal(z1, z0)
    assert r1["iters"] == r0["iters"] and np.array_equal(r1["hist"], r0["hist"]) and np.array_equal(r1["x"], r0["x"])


@pytest.mark.parametrize("shape", [(24, 24, 24), (40, 32, 20)])
def test_single_reduction_pcg_is_the_same_iteration(hd, monkeypatch, shape):
    """HDA_PCG_SINGLE_REDUCE=1 (opt-in): the Chronopoulos-Gear form of PCG -- <r,u>, <w,u> and <r,r> in ONE reduction per iteration,
    alpha from the recurrence -- is the same Krylov iteration in exact arithmetic: iteration count within 1 of hypre's recurrence,
    residual history within 1e-6 relative step by step while both run, the same solution to the stopping tolerance."""
    A = hd.lap7(*shape)
    b = np.random.default_rng(7).standard_normal(shape[0] * shape[1] * shape[2])
    out = {}
    for sr in ("0", "1"):
        monkeypatch.setenv("HDA_PCG_SINGLE_REDUCE", sr)
        out[sr] = hd.pcg(A, b, hd.Amg(A), hd.KrylovParams.default(False, rtol=1e-9))
    a, c = out["0"], out["1"]
    assert a["converged"] and c["converged"] and abs(a["iters"] - c["iters"]) <= 1
    m = min(len(a["hist"]), len(c["hist"]))
    assert np.allclose(a["hist"][:m], c["hist"][:m], rtol=1e-6)
    assert rel(c["x"], a["x"]) < 1e-7
    S = A.to_scipy()
    assert np.linalg.norm(b - S @ c["x"]) / np.linalg.norm(b) < 2e-9


def test_spmv_rectangular(orc, hd):
    rng = np.random.default_rng(5)
    M = sp.random(123, 57, density=0.1, random_state=rng, format="csr")
    Ao, Ah = both(orc, hd, M)
    x = rng.standard_normal(57)
    assert rel(Ah.spmv(x), M @ x) < 1e-12


def test_dot_and_norm_anchor(hd, pins):
    """tests/test_linsys.c:4126-4155: L2 norm of [1,-2,3] = sqrt(14)."""
    import ctypes as C
    v = np.array([1.0, -2.0, 3.0])
    out = C.c_double()
    L = hd.load()
    assert L.hda_dot(3, v.ctypes.data_as(C.POINTER(C.c_double)), v.ctypes.data_as(C.POINTER(C.c_double)),
                     C.byref(out)) == 0
    assert np.sqrt(out.value) == pytest.approx(pins["unit"]["norms_of_1_m2_3"]["L2"], rel=1e-15)
    big = np.random.default_rng(0).standard_normal(1_000_003)
    assert L.hda_dot(big.size, big.ctypes.data_as(C.POINTER(C.c_double)),
                     big.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out)) == 0
    assert out.value == pytest.approx(float(big @ big), rel=1e-13)


# ------------------------------------------------------------- K2 smoothers

@pytest.mark.parametrize("rtype,weight", [(18, 1.0), (0, 0.8), (7, 1.0)])
def test_relax_jacobi(orc, hd, rtype, weight):
    Ao, b = orc.lap7(12, 11, 10, b_mode=1)
    Ah = hd.lap7(12, 11, 10)
    x0 = np.random.default_rng(4).standard_normal(Ao.nrows)
    l1 = orc.l1_norms(Ao, 1 if rtype == 18 else 4)
    xo = x0
    for _ in range(3):
        xo = orc.relax(Ao, l1, rtype, weight, b, xo)
    xh = Ah.relax(b, x0, rtype, weight, sweeps=3)
    assert rel(xh, xo) < RTOL_REDUCE
    assert np.array_equal(Ah.l1_norms(1), orc.l1_norms(Ao, 1))  # bit-exact


@pytest.mark.parametrize("rtype", [3, 4, 6, 13, 14, 8])
def test_relax_hybrid_gauss_seidel(orc, hd, rtype):
    """K3: level-scheduled sweeps reproduce the sequential sweep of the oracle (one rank:
    l1 divisor = a_ii, so 13/14/8 equal 3/4/6)."""
    Ao, b = orc.lap7(11, 9, 8, b_mode=1)
    Ah = hd.lap7(11, 9, 8)
    x0 = np.random.default_rng(7).standard_normal(Ao.nrows)
    l1 = orc.l1_norms(Ao, 4)
    xo = x0
    for _ in range(2):
        xo = orc.relax(Ao, l1, rtype, 1.0, b, xo)
    assert rel(Ah.relax(b, x0, rtype, 1.0, sweeps=2), xo) < RTOL_REDUCE
    # irregular symmetric pattern
    M = rand_spd(700, 0.01, 21)
    Mo, Mh = both(orc, hd, M)
    bb = np.random.default_rng(8).standard_normal(700)
    xo = orc.relax(Mo, orc.l1_norms(Mo, 4), rtype, 0.9, bb, np.zeros(700))
    assert rel(Mh.relax(bb, np.zeros(700), rtype, 0.9, sweeps=1), xo) < 1e-12


def test_gauss_seidel_nonsymmetric_pattern(orc, hd):
    """anti-dependencies of a structurally nonsymmetric matrix are honoured (levels come from A u A^T)"""
    rng = np.random.default_rng(5)
    M = sp.random(400, 400, density=0.02, random_state=rng, format="csr") + sp.diags(np.full(400, 5.0))
    M = M.tocsr()
    Mo, Mh = both(orc, hd, M)
    bb = rng.standard_normal(400)
    for rtype in (3, 4):
        xo = orc.relax(Mo, orc.l1_norms(Mo, 4), rtype, 1.0, bb, np.ones(400))
        assert rel(Mh.relax(bb, np.ones(400), rtype, 1.0, sweeps=1), xo) < 1e-12


# ----------------------------------------------------------- K4 strength/PMIS

@pytest.mark.parametrize("shape", [(10, 10, 10), (20, 7, 13), (40, 40, 40)])
def test_strength_and_pmis_bit_exact(orc, hd, shape):
    Ao, _ = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    so = orc.strength(Ao)
    sh = Ah.strength()
    assert np.array_equal(so, sh)
    co = orc.pmis(Ao, so, seed=2747, level=0)
    ch = Ah.pmis(sh, seed=2747, level=0)
    assert np.array_equal(co, ch)
    assert set(np.unique(ch)) <= {1, -1, -3}


def test_strength_pmis_irregular(orc, hd):
    M = rand_spd(1500, 0.004, 9)
    M = (-abs(M) + 2 * sp.diags(M.diagonal())).tocsr()  # M-matrix-like: negative off-diagonals
    Ao, Ah = both(orc, hd, M)
    for theta, mrs in ((0.25, 0.9), (0.5, 1.0), (0.1, 0.5)):
        so, sh = orc.strength(Ao, theta, mrs), Ah.strength(theta, mrs)
        assert np.array_equal(so, sh)
        assert np.array_equal(orc.pmis(Ao, so, 7, 2, 100), Ah.pmis(sh, 7, 2, 100))


@pytest.mark.parametrize("avg", [18, 40, 90])
def test_long_row_setup_kernels_bit_exact(orc, hd, avg):
    """Rows of 18 / 40 / 90 entries on average (one of 400, one empty apart from its diagonal): the lanes-per-row forms of the
    strength and l1-norm kernels (16 / 32 / 64 lanes; the row sums handed round in entry order), PMIS, the wavefront interpolation
    kernel and the Galerkin product with its register sort, each bit for bit against the oracle on an operator with off-diagonals of
    both signs."""
    n = 1500
    rng = np.random.default_rng(avg)
    M = sp.random(n, n, density=avg / n, random_state=rng, format="lil")
    M[3, rng.choice(n, size=400, replace=False)] = -0.25
    M[9, :] = 0
    M = M.tocsr()
    M = (M + M.T) * 0.5
    M.data = np.where(rng.random(M.nnz) < 0.7, -1.0, 0.3) * np.abs(M.data)
    M = (M + M.T).tolil()
    M.setdiag(np.asarray(abs(M.tocsr()).sum(axis=1)).ravel() * 0.6 + 0.5)
    M = M.tocsr()
    M.sort_indices()
    Ao, Ah = both(orc, hd, M)
    for opt in (1, 4):
        assert np.array_equal(Ah.l1_norms(opt), orc.l1_norms(Ao, opt)), opt
    for theta, mrs in ((0.25, 0.9), (0.6, 1.0)):
        so, sh = orc.strength(Ao, theta, mrs), Ah.strength(theta, mrs)
        assert np.array_equal(so, sh)
    cf = orc.pmis(Ao, so)
    assert np.array_equal(cf, Ah.pmis(sh))
    Po = orc.interp_extpi(Ao, so, cf, 4, 0.0)
    Ph = Ah.interp_extpi(sh, cf, 4, 0.0)
    rp, cj, v = Ph.download()
    assert np.array_equal(rp, Po.rowptr) and np.array_equal(cj, Po.col) and np.array_equal(v, Po.val)
    Aco = orc.rap(Ao, Po)
    rp, cj, v = Ah.rap(Ph).download()
    assert np.array_equal(rp, Aco.rowptr) and np.array_equal(cj, Aco.col) and np.array_equal(v, Aco.val)


# ------------------------------------------------------- K5 interpolation, K6 RAP

@pytest.mark.parametrize("shape,pmax,tf", [((10, 10, 10), 4, 0.0), ((16, 9, 12), 4, 0.0),
                                           ((12, 12, 12), 0, 0.0), ((12, 12, 12), 3, 0.2)])
def test_interp_and_rap_bit_exact(orc, hd, shape, pmax, tf):
    Ao, _ = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    sm = orc.strength(Ao)
    cf = orc.pmis(Ao, sm)
    Po = orc.interp_extpi(Ao, sm, cf, pmax, tf)
    Ph = Ah.interp_extpi(sm, cf, pmax, tf)
    rp, cj, v = Ph.download()
    assert np.array_equal(rp, Po.rowptr) and np.array_equal(cj, Po.col)
    assert np.array_equal(v, Po.val), np.abs(v - Po.val).max()
    Aco = orc.rap(Ao, Po)
    Ach = Ah.rap(Ph)
    rp, cj, v = Ach.download()
    assert np.array_equal(rp, Aco.rowptr) and np.array_equal(cj, Aco.col)
    assert np.array_equal(v, Aco.val), np.abs(v - Aco.val).max()


@pytest.mark.parametrize("pmax,tf", [(4, 0.0), (0, 0.0), (2, 0.3)])
def test_direct_interp_bit_exact(orc, hd, pmax, tf):
    """Interpolation type 3 (`direct_sep_weights`, the fifth variant of the reference's examples/ex8.yml): strong C neighbours
    only, negative and positive entries scaled separately.  Same pattern and the same bits as the oracle on a stencil operator
    and on operators with off-diagonals of both signs (beta != 1), with and without truncation."""
    mats = [both(orc, hd, orc.lap7(12, 11, 10)[0].to_scipy())]
    for seed in (3, 4):
        M = rand_spd(600, 0.02, seed)  # positive off-diagonals
        mats.append(both(orc, hd, M))
        rng = np.random.default_rng(seed)
        M2 = M.copy()
        M2.data = np.where(rng.random(M2.nnz) < 0.5, -1.0, 1.0) * M2.data
        M2 = ((M2 + M2.T) * 0.5).tocsr()
        M2.setdiag(np.asarray(abs(M2).sum(axis=1)).ravel() + 1.0)
        M2.sort_indices()
        mats.append(both(orc, hd, M2.tocsr()))
    for Ao, Ah in mats:
        sm = orc.strength(Ao, 0.25, 0.9)
        cf = orc.pmis(Ao, sm)
        Po = orc.interp_direct(Ao, sm, cf, pmax, tf)
        rp, cj, v = Ah.interp_direct(sm, cf, pmax, tf).download()
        assert np.array_equal(rp, Po.rowptr) and np.array_equal(cj, Po.col)
        assert np.array_equal(v, Po.val), np.abs(v - Po.val).max()
        if pmax:
            assert np.diff(rp).max() <= pmax
        F = np.flatnonzero(cf == -1)
        assert np.all(np.diff(rp)[cf == 1] == 1) and (len(F) == 0 or np.diff(rp)[F].max() > 0)


def test_direct_interp_hierarchy_and_pcg_match_oracle(orc, hd):
    """The whole setup with interpolation type 3 (PMIS 0.5, two symmetric l1 Gauss-Seidel sweeps: examples/ex8.yml:69-79)."""
    Ao, b = orc.lap7(14, 13, 12, b_mode=1)
    Ah = hd.lap7(14, 13, 12)
    kw = dict(coarsen_type=8, interp_type=3, strong_th=0.5, relax_down=8, relax_up=8, sweeps_down=2, sweeps_up=2)
    ho = orc.Amg(Ao, orc.amg_params(False, **kw))
    hh = hd.Amg(Ah, hd.AmgParams.default(relax_coarse=9, **kw))
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels - 1):
        rp, cj, v = hh.level_matrix(l, 1).download()
        Pl = ho.level_P(l)
        assert np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col) and np.array_equal(v, Pl.val)
    ro = orc.pcg(Ao, b, ho, orc.krylov_params(False, rtol=1e-9, max_iter=100))
    rh = hd.pcg(Ah, b, hh, hd.KrylovParams.default(False, rtol=1e-9, max_iter=100))
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9)


@pytest.mark.parametrize("pmax,tf", [(4, 0.0), (0, 0.0), (3, 0.2)])
def test_standard_interp_bit_exact(orc, hd, pmax, tf):
    """Interpolation type 8 (`standard`, what examples/refOutput/ex8.txt:74 echoes for the fifth ex8 variant): the extended
    interpolatory set, strong F neighbours eliminated through their own rows, direct interpolation on the widened stencil.  Same
    pattern and the same bits as the oracle on a stencil operator, an anisotropic one and random operators with off-diagonals
    of both signs, with and without truncation."""
    mats = [both(orc, hd, orc.lap7(12, 11, 10)[0].to_scipy()), both(orc, hd, orc.lap7(9, 10, 11, c=(1.0, 0.1, 10.0))[0].to_scipy())]
    for seed in (5, 6):
        M = rand_spd(500, 0.02, seed)
        rng = np.random.default_rng(seed)
        M2 = M.copy()
        M2.data = np.where(rng.random(M2.nnz) < 0.3, -1.0, 1.0) * M2.data
        M2 = ((M2 + M2.T) * 0.5).tocsr()
        M2.setdiag(np.asarray(abs(M2).sum(axis=1)).ravel() + 1.0)
        M2.sort_indices()
        mats += [both(orc, hd, M), both(orc, hd, M2.tocsr())]
    for Ao, Ah in mats:
        sm = orc.strength(Ao, 0.25, 0.9)
        cf = orc.pmis(Ao, sm)
        Po = orc.interp_standard(Ao, sm, cf, pmax, tf)
        rp, cj, v = Ah.interp_standard(sm, cf, pmax, tf).download()
        assert np.array_equal(rp, Po.rowptr) and np.array_equal(cj, Po.col)
        assert np.array_equal(v, Po.val), np.abs(v - Po.val).max()
        if pmax:
            assert np.diff(rp).max() <= pmax
        # the pattern is the extended one: without truncation the same entries as extended+i names
        if pmax == 0 and tf == 0.0:
            Pe = orc.interp_extpi(Ao, sm, cf, 0, 0.0)
            assert np.array_equal(Pe.rowptr, Po.rowptr) and np.array_equal(Pe.col, Po.col)


def test_standard_interp_hierarchy_and_pcg_match_oracle(orc, hd):
    """The whole setup with interpolation type 8 in the configuration examples/refOutput/ex8.txt:63-79 echoes (PMIS 0.5, two
    symmetric l1 Gauss-Seidel sweeps): identical transfer operators on every level, the oracle's iteration count and history."""
    Ao, b = orc.lap7(14, 13, 12, b_mode=1)
    Ah = hd.lap7(14, 13, 12)
    kw = dict(coarsen_type=8, interp_type=8, strong_th=0.5, relax_down=8, relax_up=8, sweeps_down=2, sweeps_up=2)
    ho = orc.Amg(Ao, orc.amg_params(False, **kw))
    hh = hd.Amg(Ah, hd.AmgParams.default(relax_coarse=9, **kw))
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels - 1):
        rp, cj, v = hh.level_matrix(l, 1).download()
        Pl = ho.level_P(l)
        assert np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col) and np.array_equal(v, Pl.val)
    ro = orc.pcg(Ao, b, ho, orc.krylov_params(False, rtol=1e-9, max_iter=100))
    rh = hd.pcg(Ah, b, hh, hd.KrylovParams.default(False, rtol=1e-9, max_iter=100))
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9)


def test_spgemm_and_transpose_vs_scipy(orc, hd, monkeypatch):
    rng = np.random.default_rng(11)
    X = sp.random(400, 300, density=0.03, random_state=rng, format="csr")
    Y = sp.random(300, 250, density=0.05, random_state=rng, format="csr")
    Xh, Yh = hd.Csr.from_scipy(X), hd.Csr.from_scipy(Y)
    Ch = Xh.matmul(Yh).to_scipy()
    ref = (X @ Y).tocsr()
    assert abs(Ch - ref).max() < 1e-13
    assert (Ch != 0).nnz >= ref.nnz  # structural entries kept
    T = Xh.transpose().to_scipy()
    assert abs(T - X.T.tocsr()).max() == 0.0
    rp, cj, _ = Xh.transpose().download()
    for i in range(300):
        assert np.all(np.diff(cj[rp[i]:rp[i + 1]]) > 0)


# ------------------------------------------------------------------ hierarchy

@pytest.mark.parametrize("shape", [(10, 10, 10), (24, 24, 24), (48, 32, 20)])
def test_hierarchy_identical_to_oracle(orc, hd, shape):
    Ao, b = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    ho = orc.Amg(Ao, orc.amg_params(True))
    hh = hd.Amg(Ah)
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels):
        rp, cj, v = hh.level_matrix(l, 0).download()
        Al = ho.level_A(l)
        assert np.array_equal(rp, Al.rowptr) and np.array_equal(cj, Al.col), f"A pattern level {l}"
        assert np.array_equal(v, Al.val), f"A values level {l}: {np.abs(v - Al.val).max()}"
        if l < ho.num_levels - 1:
            assert np.array_equal(hh.level_cf(l), ho.level_cf(l)), f"C/F level {l}"
            rp, cj, v = hh.level_matrix(l, 1).download()
            Pl = ho.level_P(l)
            assert np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col)
            assert np.array_equal(v, Pl.val)
    g, o = hh.complexities
    assert g == pytest.approx(ho.grid_complexity, rel=1e-15)
    assert o == pytest.approx(ho.operator_complexity, rel=1e-15)
    # one V-cycle from zero
    r = np.random.default_rng(3).standard_normal(Ao.nrows)
    assert rel(hh.vcycle(r), ho.vcycle(r)) < 1e-12


def _agg_mats(orc, hd):
    """A stencil operator, an anisotropic one (strong connections along two axes only) and irregular M-matrices."""
    mats = [both(orc, hd, orc.lap7(12, 11, 10)[0].to_scipy()), both(orc, hd, hd.lap7(14, 12, 10, c=(1.0, 1.0, 0.01)).to_scipy())]
    for seed in (5, 6):
        M = rand_spd(900, 0.01, seed)
        M.data = -np.abs(M.data)
        M.setdiag(0.0)
        M.eliminate_zeros()
        M = (M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 0.1)).tocsr()
        M.sort_indices()
        mats.append(both(orc, hd, M))
    return mats


@pytest.mark.parametrize("num_paths", [1, 2])
def test_aggressive_coarsening_stages_bit_exact(orc, hd, num_paths):
    """Aggressive coarsening (reference AMGagg_args src/internal/amg.c:160-173, forwarded at :938-944; hda_amg_agg.hip), stage by
    stage against the oracle: the second strength graph among the first pass's C points (pattern AND path counts), the C/F
    marker after the second PMIS pass, and the multipass interpolation -- same pattern, same bits (sums in the oracle's order; the
    pass products on the deterministic SpGEMM).  Parity unpinned upstream: no reference output uses these options."""
    for Ao, Ah in _agg_mats(orc, hd):
        sm = orc.strength(Ao, 0.25, 0.9)
        cf1 = orc.pmis(Ao, sm)
        S2o = orc.second_strength(Ao, sm, cf1, num_paths)
        rp, cj, v = Ah.second_strength(sm, cf1, num_paths).download()
        assert np.array_equal(rp, S2o.rowptr) and np.array_equal(cj, S2o.col) and np.array_equal(v, S2o.val)
        assert S2o.nrows == int((cf1 == 1).sum()) and (v >= num_paths).all()
        cfo = orc.coarsen_second_pass(Ao, sm, cf1, num_paths, 2747, 0)
        cfh = Ah.coarsen_second_pass(sm, cf1, num_paths, 2747, 0)
        assert np.array_equal(cfo, cfh)
        assert (cfo == 1).sum() < (cf1 == 1).sum() and set(np.flatnonzero(cfo == 1)) <= set(np.flatnonzero(cf1 == 1))
        Po = orc.interp_multipass(Ao, sm, cfo)
        rp, cj, v = Ah.interp_multipass(sm, cfo).download()
        assert np.array_equal(rp, Po.rowptr) and np.array_equal(cj, Po.col)
        assert np.array_equal(v, Po.val), np.abs(v - Po.val).max()
        # truncation of the finished rows (aggressive.max_nnz_row / trunc_factor): same survivors, same rescaled weights, row sums kept
        for pmax, tf in ((2, 0.0), (0, 0.4), (3, 0.2)):
            Pt_o = orc.truncate_rows(orc.interp_multipass(Ao, sm, cfo), pmax, tf)
            rp2, cj2, v2 = Ah.interp_multipass(sm, cfo).truncate_rows(pmax, tf).download()
            assert np.array_equal(rp2, Pt_o.rowptr) and np.array_equal(cj2, Pt_o.col) and np.array_equal(v2, Pt_o.val), (pmax, tf)
            if pmax:
                assert np.diff(rp2).max() <= pmax
            full, cut = Po.to_scipy().tocsr(), Pt_o.to_scipy().tocsr()
            assert np.allclose(np.asarray(cut.sum(axis=1)).ravel(), np.asarray(full.sum(axis=1)).ravel(), rtol=1e-12, atol=1e-14)
        # what multipass interpolation promises: C rows are identity rows; a row sums to (a_ii - rowsum_i) / a_ii times the mean row
        # sum of the rows it interpolates through -- exactly 1 everywhere on an operator whose rows all sum to zero (below)
        P = Po.to_scipy().tocsr()
        prs = np.asarray(P.sum(axis=1)).ravel()
        assert np.allclose(prs[cfo == 1], 1.0) and (np.diff(P.indptr)[cfo == 1] == 1).all()
        assert (prs[np.diff(P.indptr) > 0] <= 1.0 + 1e-12).all()
    # a graph Laplacian (every row sums to zero): constants are interpolated exactly by every pass
    W = rand_spd(700, 0.012, 9)
    W.setdiag(0.0)
    W.eliminate_zeros()
    W.data = np.abs(W.data)
    L = (sp.diags(np.asarray(W.sum(axis=1)).ravel()) - W).tocsr()
    L.sort_indices()
    Ao, Ah = both(orc, hd, L)
    sm = orc.strength(Ao, 0.25, 2.0)  # (max_row_sum 2: the row-sum test must not weaken rows here)
    cfo = orc.coarsen_second_pass(Ao, sm, orc.pmis(Ao, sm), num_paths, 2747, 0)
    Po = orc.interp_multipass(Ao, sm, cfo)
    rp, cj, v = Ah.interp_multipass(sm, cfo).download()
    assert np.array_equal(rp, Po.rowptr) and np.array_equal(cj, Po.col) and np.array_equal(v, Po.val)
    P = Po.to_scipy().tocsr()
    filled = np.diff(P.indptr) > 0
    assert filled.sum() > 0.9 * L.shape[0]
    assert np.allclose(np.asarray(P.sum(axis=1)).ravel()[filled], 1.0, atol=1e-12)


@pytest.mark.parametrize("shape,agg,trunc", [((16, 16, 16), 1, None), ((20, 18, 16), 2, None), ((24, 24, 24), 1, None), ((20, 20, 20), 1, (2, 0.0)),
                                             ((18, 18, 18), 2, (0, 0.3))])
def test_aggressive_hierarchy_and_pcg_match_oracle(orc, hd, shape, agg, trunc):
    """`aggressive.num_levels` 1 and 2: every operator and interpolation of the hierarchy bit-identical to the oracle's, the coarsening
    really is aggressive (first level coarsens by more than 8 where PMIS + extended+i coarsens by about 3; operator complexity below
    1.5 against 2.6), PCG takes the oracle's iterations with the oracle's history."""
    Ao, b = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    extra = dict(agg_pmax=trunc[0], agg_trunc_factor=trunc[1]) if trunc else {}
    ho = orc.Amg(Ao, orc.amg_params(True, agg_num_levels=agg, **extra))
    hh = hd.Amg(Ah, hd.AmgParams.default(agg_num_levels=agg, **extra))
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels):
        rp, cj, v = hh.level_matrix(l, 0).download()
        Al = ho.level_A(l)
        assert np.array_equal(rp, Al.rowptr) and np.array_equal(cj, Al.col) and np.array_equal(v, Al.val), f"A level {l}"
        if l < ho.num_levels - 1:
            assert np.array_equal(hh.level_cf(l), ho.level_cf(l)), f"C/F level {l}"
            rp, cj, v = hh.level_matrix(l, 1).download()
            Pl = ho.level_P(l)
            assert np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col) and np.array_equal(v, Pl.val), f"P level {l}"
    assert ho.level_A(0).nrows > 8 * ho.level_A(1).nrows and ho.operator_complexity < 1.5
    plain = orc.Amg(Ao, orc.amg_params(True))
    assert plain.operator_complexity > 2.0
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)
    assert rel(rh["x"], ro["x"]) < 1e-10


@pytest.mark.parametrize("n", [128, 256])
def test_parity_at_128_and_256_cubed(orc, hd, n):
    """The benchmark workload itself -- BASELINE config 2, 256^3 = 16.8 M rows (oracle setup about 45 s on the box's host cores) --
    and one size down (128^3, 2.1 M rows, 8 levels): every level has the oracle's rows and entries, the same C/F
    splitting on level 0, the same entry sums (the coarse levels are renumbered for the solve phase -- a permutation
    similarity, so patterns are compared through invariants), the same complexities to 1e-14, and PCG takes the oracle's
    iterations with the oracle's residual history.  Reference tolerance: SURVEY 8(c) "same iteration count +-1"; here: equal,
    history 1e-9.  (bench.py asserts the 256^3 iteration count too: cpu_baseline.iters_match.)"""
    Ao, b = orc.lap7(n, n, n)
    Ah = hd.lap7(n, n, n)
    ho = orc.Amg(Ao, orc.amg_params(True))
    hh = hd.Amg(Ah)
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels):
        Al = ho.level_A(l)
        nr, nc, nnz = hh.level_matrix(l, 0).dims
        assert (nr, nnz) == (Al.nrows, Al.nnz), f"level {l}: {nr} rows / {nnz} entries, oracle {Al.nrows} / {Al.nnz}"
        if l in (0, 2, ho.num_levels - 1):
            _, _, v = hh.level_matrix(l, 0).download()
            assert np.sum(np.abs(v)) == pytest.approx(np.sum(np.abs(Al.val)), rel=1e-12)
    assert np.array_equal(hh.level_cf(0), ho.level_cf(0))
    g, o = hh.complexities
    assert g == pytest.approx(ho.grid_complexity, rel=1e-14) and o == pytest.approx(ho.operator_complexity, rel=1e-14)
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9, atol=0)
    assert rel(rh["x"], ro["x"]) < 1e-9


@pytest.mark.parametrize("shape", [(10, 10, 10), (16, 12, 9)])
def test_hmis_hl1gs_cpu_defaults_identical_to_oracle(orc, hd, shape):
    """The reference's CPU defaults (src/internal/amg.c:120-238: HMIS = Ruge first pass on one rank,
    hybrid l1 Gauss-Seidel 13/14) on the device: same C/F splitting, same hierarchy, same PCG history."""
    Ao, b = orc.lap7(*shape, b_mode=1)
    Ah = hd.lap7(*shape)
    po = orc.amg_params(False)
    ph = hd.AmgParams.default(coarsen_type=po.coarsen_type, relax_down=po.relax_down, relax_up=po.relax_up, relax_coarse=po.relax_coarse)
    assert po.coarsen_type == 10 and (po.relax_down, po.relax_up) == (13, 14)
    ho, hh = orc.Amg(Ao, po), hd.Amg(Ah, ph)
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels - 1):
        assert np.array_equal(hh.level_cf(l), ho.level_cf(l)), f"C/F level {l}"
        rp, cj, v = hh.level_matrix(l + 1, 0).download()
        Al = ho.level_A(l + 1)
        assert np.array_equal(rp, Al.rowptr) and np.array_equal(cj, Al.col) and np.array_equal(v, Al.val)
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)


def test_pin_ex1_reference_cpu_defaults_on_gpu(hd, pins):
    """examples/refOutput/ex1.txt:27 reproduced by the product itself (not only by the oracle):
    10^3 7-pt, b = 1, PCG + BoomerAMG with the reference's CPU defaults -> 6 iterations, 4.98e-08."""
    A = hd.lap7(10, 10, 10)
    b = np.ones(1000)
    amg = hd.Amg(A, hd.AmgParams.default(coarsen_type=10, relax_down=13, relax_up=14, relax_coarse=9))
    r = hd.pcg(A, b, amg)
    ref = pins["ex1"]["stats"][0]
    assert r["converged"] and r["iters"] == ref["iters"] == 6
    S = A.to_scipy()
    true_rel = np.linalg.norm(b - S @ r["x"]) / np.linalg.norm(b)
    assert true_rel == pytest.approx(ref["rel"], rel=0.02)


def test_hmis_refuses_large_systems(hd):
    A = hd.lap7(140, 140, 140)  # 2 744 000 rows > the one-thread limit of 2 500 000
    with pytest.raises(hd.LibraryError, match="HMIS"):
        hd.Amg(A, hd.AmgParams.default(coarsen_type=10))


def test_hmis_cpu_defaults_at_64_cubed_match_oracle(orc, hd):
    """The reference's CPU-build defaults (HMIS + hybrid l1 Gauss-Seidel 13 / 14, src/internal/amg.c:138-146,183-189) one size above
    what round 2 allowed (262 144 rows; the Ruge first pass is one device thread, 2.5 M rows is the limit now): same C/F splitting
    on every level, PCG with the oracle's iterations and history."""
    n = 64
    Ao, b = orc.lap7(n, n, n)
    Ah = hd.lap7(n, n, n)
    po = orc.amg_params(False)
    ph = hd.AmgParams.default(coarsen_type=po.coarsen_type, relax_down=po.relax_down, relax_up=po.relax_up, relax_coarse=po.relax_coarse)
    ho, hh = orc.Amg(Ao, po), hd.Amg(Ah, ph)
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels - 1):
        assert np.array_equal(hh.level_cf(l), ho.level_cf(l)), f"C/F level {l}"
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("cpu_defaults,layout", [(False, "interleaved"), (True, "interleaved"), (False, "contiguous")])
def test_systems_amg_unknown_approach_identical_to_oracle(orc, hd, cpu_defaults, layout):
    """coarsening.num_functions = 3 (presets elasticity_2d/3d, reference src/internal/presets.c:19-27,
    src/internal/amg.c:792-862): couplings between different functions are ignored by the strength
    measure and never lumped into the interpolation diagonal.  3 unknowns per node of an 8^3 grid,
    block-coupled; dof_func interleaved (hypre's default, no map given) or contiguous (map given)."""
    import scipy.sparse as sp
    n = 8
    I = sp.identity(n)
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    L = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr()
    B = np.array([[1.0, 0.85, 0.1], [0.85, 1.2, 0.25], [0.1, 0.25, 0.9]])  # SPD; 0.85 > strong_th: the scalar measure would call it strong
    if layout == "interleaved":
        A = sp.kron(L, B).tocsr()
        dof = None
    else:
        A = sp.kron(B, L).tocsr()
        dof = np.repeat(np.arange(3), n ** 3).astype(np.int32)
    A.sort_indices()
    N = A.shape[0]
    Ao = orc.Csr.from_arrays(N, N, A.indptr, A.indices, A.data)
    Ah = hd.Csr.from_arrays(N, N, A.indptr, A.indices, A.data)
    po = orc.amg_params(not cpu_defaults, num_functions=3, strong_th=0.8)
    ph = hd.AmgParams.default(num_functions=3, strong_th=0.8, coarsen_type=po.coarsen_type, relax_down=po.relax_down,
                              relax_up=po.relax_up, relax_coarse=po.relax_coarse)
    ho, hh = orc.Amg(Ao, po, dof=dof), hd.Amg(Ah, ph, dof=dof)
    assert hh.num_levels == ho.num_levels >= 2
    for l in range(ho.num_levels - 1):
        assert np.array_equal(hh.level_cf(l), ho.level_cf(l)), f"C/F level {l}"
        rp, cj, v = hh.level_matrix(l, 1).download()
        Pl = ho.level_P(l)
        assert np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col) and np.array_equal(v, Pl.val), f"P level {l}"
        rp, cj, v = hh.level_matrix(l + 1, 0).download()
        Al = ho.level_A(l + 1)
        assert np.array_equal(rp, Al.rowptr) and np.array_equal(cj, Al.col) and np.array_equal(v, Al.val)
    # the scalar hierarchy of the same matrix is a different one (the option does something)
    assert not np.array_equal(hd.Amg(Ah, hd.AmgParams.default(strong_th=0.8, coarsen_type=po.coarsen_type)).level_cf(0), hh.level_cf(0))
    b = np.ones(N)
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)


def test_systems_amg_with_direct_interpolation_identical_to_oracle(orc, hd):
    """num_functions = 3 together with interpolation type 3: the direct weights' row sums leave other functions' couplings out
    (positive off-diagonals here, so beta is exercised as well); hierarchy and PCG against the oracle."""
    import scipy.sparse as sp
    n = 7
    I = sp.identity(n)
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    L = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr()
    B = np.array([[1.0, 0.85, 0.1], [0.85, 1.2, 0.25], [0.1, 0.25, 0.9]])
    A = sp.kron(L, B).tocsr()
    A.sort_indices()
    N = A.shape[0]
    Ao, Ah = orc.Csr.from_arrays(N, N, A.indptr, A.indices, A.data), hd.Csr.from_arrays(N, N, A.indptr, A.indices, A.data)
    kw = dict(num_functions=3, strong_th=0.8, interp_type=3)
    po = orc.amg_params(True, **kw)
    ho, hh = orc.Amg(Ao, po), hd.Amg(Ah, hd.AmgParams.default(**kw))
    assert hh.num_levels == ho.num_levels >= 2
    for l in range(ho.num_levels - 1):
        rp, cj, v = hh.level_matrix(l, 1).download()
        Pl = ho.level_P(l)
        assert np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col) and np.array_equal(v, Pl.val), f"P level {l}"
    b = np.ones(N)
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"] and np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)


def _random_system(seed, kind):
    """Irregular test operators: weighted graph Laplacians (M-matrices), a version with positive
    off-diagonals mixed in, an unsymmetric convection-like one, and one with a negative diagonal."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = int(rng.integers(300, 900))
    m = n * int(rng.integers(3, 9))
    i, j = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = i != j
    i, j = i[keep], j[keep]
    w = rng.uniform(0.05, 1.0, i.size) * (10.0 ** rng.integers(-2, 2, i.size))
    if kind == "mixed":
        w *= np.where(rng.uniform(size=i.size) < 0.15, -0.3, 1.0)
    W = sp.coo_matrix((w, (i, j)), shape=(n, n)).tocsr()
    if kind != "unsym":
        W = W + W.T
    else:
        W = W + 0.3 * W.T
    W.sum_duplicates()
    d = np.asarray(abs(W).sum(axis=1)).ravel() + rng.uniform(0.01, 0.5, n)
    A = (sp.diags(d) - W).tocsr()
    if kind == "negdiag":
        A = -A
    A.sort_indices()
    return A


@pytest.mark.parametrize("kind", ["mmatrix", "mixed", "unsym", "negdiag"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_operators_hierarchy_identical_to_oracle(orc, hd, kind, seed):
    """Differential test on irregular sparsity (no grid structure, rows of very different length,
    weights over four decades): strength, PMIS, ext+i with truncation, Galerkin products and the
    solve must still agree with the oracle bit for bit / to 1e-10."""
    A = _random_system(100 * seed + len(kind), kind)
    N = A.shape[0]
    Ao = orc.Csr.from_arrays(N, N, A.indptr, A.indices, A.data)
    Ah = hd.Csr.from_arrays(N, N, A.indptr, A.indices, A.data)
    ho, hh = orc.Amg(Ao, orc.amg_params(True)), hd.Amg(Ah)
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels - 1):
        assert np.array_equal(hh.level_cf(l), ho.level_cf(l)), f"C/F level {l}"
        rp, cj, v = hh.level_matrix(l, 1).download()
        Pl = ho.level_P(l)
        assert np.array_equal(rp, Pl.rowptr) and np.array_equal(cj, Pl.col) and np.array_equal(v, Pl.val), f"P level {l}"
        rp, cj, v = hh.level_matrix(l + 1, 0).download()
        Al = ho.level_A(l + 1)
        assert np.array_equal(rp, Al.rowptr) and np.array_equal(cj, Al.col) and np.array_equal(v, Al.val), f"A level {l + 1}"
    b = np.random.default_rng(seed).standard_normal(N)
    if kind == "unsym":
        ro, rh = orc.gmres(Ao, b, ho), hd.gmres(Ah, b, hh)
    else:
        ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] == ro["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-8, atol=0)


def test_pin_ex2_hierarchy_on_gpu(hd, pins):
    """examples/refOutput/ex2.txt:124-139 against the HIP-built hierarchy."""
    import scipy.sparse as sp  # noqa
    A = hd.lap7(10, 10, 10)
    amg = hd.Amg(A)
    ref = pins["ex2"]
    assert amg.num_levels == len(ref["operators"])
    for l, op in enumerate(ref["operators"]):
        n, _, nnz = amg.level_matrix(l, 0).dims
        assert n == pytest.approx(op["rows"], rel=0.12)
        assert nnz == pytest.approx(op["nnz"], rel=0.15)
    g, o = amg.complexities
    assert g == pytest.approx(ref["grid_complexity"], rel=0.02)
    assert o == pytest.approx(ref["operator_complexity"], rel=0.05)


# --------------------------------------------------------------------- Krylov

@pytest.mark.parametrize("shape,bmode", [((10, 10, 10), 1), ((10, 10, 10), 0), ((32, 32, 32), 0),
                                          ((40, 25, 17), 1)])
def test_amg_pcg_matches_oracle(orc, hd, shape, bmode):
    Ao, b = orc.lap7(*shape, b_mode=bmode)
    Ah = hd.lap7(*shape)
    ro = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    rh = hd.pcg(Ah, b, hd.Amg(Ah))
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)
    assert rel(rh["x"], ro["x"]) < 1e-10
    S = Ao.to_scipy()
    assert np.linalg.norm(b - S @ rh["x"]) / np.linalg.norm(b) < 1e-6


@pytest.mark.parametrize("down,up", [(13, 14), (3, 4), (6, 6)])
def test_amg_pcg_hybrid_gs_matches_oracle(orc, hd, down, up):
    """PMIS + ext+i hierarchy with hybrid (l1) Gauss-Seidel smoothing, the relaxation of the
    reference's examples/ex2.yml:50-51, against the oracle."""
    Ao, b = orc.lap7(14, 14, 14, b_mode=1)
    Ah = hd.lap7(14, 14, 14)
    ro = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True, relax_down=down, relax_up=up)))
    rh = hd.pcg(Ah, b, hd.Amg(Ah, hd.AmgParams.default(relax_down=down, relax_up=up)))
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9)
    assert rel(rh["x"], ro["x"]) < 1e-9


def test_unpreconditioned_cg_matches_oracle(orc, hd):
    Ao, b = orc.lap7(9, 9, 9, b_mode=1)
    Ah = hd.lap7(9, 9, 9)
    kp_o = orc.krylov_params(False, max_iter=200, rtol=1e-10)
    kp_h = hd.KrylovParams.default(False, max_iter=200, rtol=1e-10)
    ro = orc.pcg(Ao, b, None, kp_o)
    rh = hd.pcg(Ah, b, None, kp_h)
    assert rh["iters"] == ro["iters"] and rh["converged"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-8)


def test_gmres_amg_matches_oracle(orc, hd):
    Ao, b = orc.lap7(16, 16, 16, b_mode=0)
    Ah = hd.lap7(16, 16, 16)
    ro = orc.gmres(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    rh = hd.gmres(Ah, b, hd.Amg(Ah))
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-8)
    assert rel(rh["x"], ro["x"]) < 1e-8


@pytest.mark.parametrize("kdim,precond", [(5, False), (4, True), (30, False)])
def test_gmres_restarts_match_oracle(orc, hd, kdim, precond):
    """Restarted GMRES(k) across several cycles (unpreconditioned 12^3 needs ~35 iterations; a
    Jacobi-only "AMG" with max_levels 1 a few cycles of 4): iteration count, history and solution
    against the oracle; also x0 != 0."""
    Ao, b = orc.lap7(12, 12, 12, b_mode=1)
    Ah = hd.lap7(12, 12, 12)
    ko = orc.krylov_params(True, krylov_dim=kdim, max_iter=200)
    kh = hd.KrylovParams.default(True, krylov_dim=kdim, max_iter=200)
    mo = orc.Amg(Ao, orc.amg_params(True, max_levels=1, relax_coarse=18)) if precond else None
    mh = hd.Amg(Ah, hd.AmgParams.default(max_levels=1, relax_coarse=18)) if precond else None
    x0 = np.linspace(-1.0, 1.0, Ao.nrows)
    for guess in (None, x0):
        ro = orc.gmres(Ao, b, mo, ko, x0=guess)
        rh = hd.gmres(Ah, b, mh, kh, x0=guess)
        assert ro["iters"] > kdim or kdim == 30
        assert rh["converged"] == ro["converged"] and rh["iters"] == ro["iters"]
        assert np.allclose(rh["hist"], ro["hist"], rtol=1e-8)
        assert rel(rh["x"], ro["x"]) < 1e-8


def test_solve_phase_renumbering_is_a_similarity(orc):
    """Coarse levels are renumbered for the solve (hda_reorder.hip: nested clusters by strongest
    interpolation parent) above HDA_REORDER rows, 50000 by default -- i.e. never in the small parity
    cases.  Forced down to 300 rows in a fresh process: the preconditioner is the same operator, so
    PCG/GMRES take the oracle's iterations and histories agree to rounding; the renumbered level
    operators are permutations of the oracle's (same multiset of values, same row-length multiset)."""
    import subprocess
    import sys
    code = """
import sys, json, numpy as np
sys.path.insert(0, %r)
import hypredrive_amd as hd
from oracle import oracle_ffi as orc
Ao, b = orc.lap7(24, 20, 18, b_mode=0)
Ah = hd.lap7(24, 20, 18)
ho, hh = orc.Amg(Ao, orc.amg_params(True)), hd.Amg(Ah)
ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
rg, rq = orc.gmres(Ao, b, ho), hd.gmres(Ah, b, hh)
rp, cj, v = hh.level_matrix(1, 0).download()
A1 = ho.level_A(1)
print(json.dumps(dict(it=[ro["iters"], rh["iters"], rg["iters"], rq["iters"]],
      hist=float(np.max(np.abs(np.array(rh["hist"]) / np.array(ro["hist"]) - 1.0))),
      x=float(np.linalg.norm(rh["x"] - ro["x"]) / np.linalg.norm(ro["x"])),
      same_order=bool(np.array_equal(cj, A1.col)), vals=bool(np.array_equal(np.sort(v), np.sort(A1.val))),
      lens=bool(np.array_equal(np.sort(np.diff(rp)), np.sort(np.diff(A1.rowptr)))), nlev=[ho.num_levels, hh.num_levels])))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HDA_REORDER="300"), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["it"][0] == res["it"][1] and res["it"][2] == res["it"][3] and res["nlev"][0] == res["nlev"][1]
    assert res["hist"] < 1e-9 and res["x"] < 1e-9
    assert not res["same_order"] and res["vals"] and res["lens"]


def test_coded_operators_change_no_result():
    """Stencil-coded A (1 B/entry), value-coded level-0 P/R (5 B/entry) and the solve-phase renumbering
    only change how operators are stored: a 96^3 AMG-PCG solve with all of them (default) and with
    none (HDA_CODED=0 HDA_REORDER=0) must take the same iterations, and agree in the solution to
    rounding.  Fresh processes, because the switches are read once."""
    import json
    import subprocess
    import sys
    code = """
import sys, json, numpy as np
sys.path.insert(0, %r)
import hypredrive_amd as hd
A = hd.lap7(96, 96, 96)
amg = hd.Amg(A)
b = np.zeros(96 ** 3); b[:96 * 96] = 1.0
r = hd.pcg(A, b, amg)
fb = hd.format_bytes(A, amg)
print(json.dumps(dict(it=r["iters"], hist=list(map(float, r["hist"])), xn=float(np.linalg.norm(r["x"])), x0=[float(v) for v in r["x"][::50021]],
                      coded=fb["coded"], ratio=fb["vcycle"] / amg.vcycle_bytes)))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for env in ({}, {"HDA_CODED": "0", "HDA_REORDER": "0", "HDA_WINDOW": "0"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b = out
    assert a["coded"] and not b["coded"] and a["ratio"] < 0.9 and b["ratio"] == 1.0   # fewer bytes streamed
    assert a["it"] == b["it"]
    assert np.allclose(a["hist"], b["hist"], rtol=1e-9) and np.allclose(a["x0"], b["x0"], rtol=1e-9, atol=1e-12)
    assert a["xn"] == pytest.approx(b["xn"], rel=1e-10)


def test_edge_cases(orc, hd, pins):
    # 1x1 system 3x = 6 (tests/test_setmatrix_from_csr.c:397-417)
    u = pins["unit"]["one_by_one"]
    A = hd.Csr.from_arrays(1, 1, [0, 1], [0], [u["a"]])
    r = hd.pcg(A, np.array([u["b"]]), hd.Amg(A))
    assert abs(np.linalg.norm(r["x"]) - u["x_norm"]) < u["tol"]
    # 1-D Laplacian n = 16 (tests/test_setmatrix_from_csr.c:168-199): single level, GE
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(16, 16), format="csr")
    A = hd.Csr.from_scipy(T)
    amg = hd.Amg(A)
    assert amg.num_levels == 1
    r = hd.pcg(A, np.ones(16), amg)
    assert r["converged"] and np.allclose(T @ r["x"], np.ones(16), atol=1e-6)
    # zero rhs -> 0 iterations, x = 0
    A = hd.lap7(5, 5, 5)
    r = hd.pcg(A, np.zeros(125), None)
    assert r["iters"] == 0 and np.all(r["x"] == 0)
    # exact initial guess (tests/test_init_guess.c:247-270)
    Ao, _ = orc.lap7(5, 5, 5)
    b = Ao.to_scipy() @ np.ones(125)
    r = hd.pcg(A, b, None, x0=np.ones(125))
    assert r["hist"][0] < 1e-12 * np.linalg.norm(b) or r["iters"] <= 1


def test_spgemm_batched_equals_unbatched(orc, hd):
    """Force the row-batched SpGEMM path with a tiny hash budget (env read once per process,
    so run in a subprocess)."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, hypredrive_amd as h\n"
        "A = h.lap7(14,14,14); sm = A.strength(); cf = A.pmis(sm); P = A.interp_extpi(sm, cf)\n"
        "rp, cj, v = A.rap(P).download(); np.savez('/tmp/_rap_%s.npz' % __import__('os').environ.get('HDA_SPGEMM_SLOTS','d'), rp=rp, cj=cj, v=v)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for slots in ("d", "4096"):
        env = dict(os.environ, PYTHONPATH=root)
        if slots != "d":
            env["HDA_SPGEMM_SLOTS"] = slots
        else:
            env.pop("HDA_SPGEMM_SLOTS", None)
        subprocess.check_call([sys.executable, "-c", code], env=env, cwd=root)
    a, b = np.load("/tmp/_rap_d.npz"), np.load("/tmp/_rap_4096.npz")
    for k in ("rp", "cj", "v"):
        assert np.array_equal(a[k], b[k])


# ------------------------------------------------- full-size property checks

def test_full_size_spmv_properties(hd):
    """256^3 (BASELINE config 2): linearity and A*1 = row sums, no oracle needed."""
    n = 256
    A = hd.lap7(n, n, n, want_rhs=False)
    N, _, nnz = A.dims
    assert N == n ** 3 and nnz == 7 * N - 6 * n * n
    ones = np.ones(N)
    y = A.spmv(ones)
    g = np.arange(n)
    bx = (g == 0).astype(float) + (g == n - 1)
    rs = (bx[None, None, :] + bx[None, :, None] + bx[:, None, None]).ravel()  # x fastest
    assert np.array_equal(y, rs)
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(N), rng.standard_normal(N)
    lhs = A.spmv(2.0 * u - 3.0 * v)
    rhs = 2.0 * A.spmv(u) - 3.0 * A.spmv(v)
    assert rel(lhs, rhs) < 1e-13
    # symmetry: <Au, v> == <u, Av>
    assert abs(A.spmv(u) @ v - u @ A.spmv(v)) / abs(u @ A.spmv(v)) < 1e-12


def test_run_form_of_windowed_csr_changes_no_result():
    """Run form of the windowed CSR (round 3): an uncoded structured-grid operator names a handful of runs of consecutive columns per
    1024-entry chunk, so its column indices shrink to 2-byte positions plus a 128-byte run table per chunk.  HDA_CODED=0 (the general-matrix
    path) with and without it (HDA_WINDOW_RUNS=0), 72^3: the level-0 operator takes the run form, same iterations, histories and solutions
    to rounding; and as a bare product against the plain kernel on a random vector."""
    import json
    import subprocess
    import sys
    code = """
import sys, json, numpy as np
sys.path.insert(0, %r)
import hypredrive_amd as hd
n = 72
A = hd.lap7(n, n, n)
x = np.random.default_rng(3).standard_normal(n ** 3)
y = A.spmv(x)
amg = hd.Amg(A)
b = np.zeros(n ** 3); b[:n * n] = 1.0
r = hd.pcg(A, b, amg)
fb = hd.format_bytes(A, amg)
print(json.dumps(dict(it=r["iters"], hist=list(map(float, r["hist"])), xn=float(np.linalg.norm(r["x"])), yn=float(np.linalg.norm(y)),
                      y0=[float(v) for v in y[::40009]], spmv_bytes=fb["spmv"], windowed=bool(fb["windowed"]))))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for env in ({"HDA_CODED": "0"}, {"HDA_CODED": "0", "HDA_WINDOW_RUNS": "0"}, {"HDA_CODED": "0", "HDA_WINDOW": "0"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append(json.loads(r.stdout.strip().splitlines()[-1]))
    runs, lists, plain = out
    assert runs["spmv_bytes"] < 0.9 * lists["spmv_bytes"]          # fewer bytes streamed by the level-0 product
    assert runs["it"] == lists["it"] == plain["it"]
    assert np.allclose(runs["hist"], plain["hist"], rtol=1e-9) and runs["xn"] == pytest.approx(plain["xn"], rel=1e-10)
    assert np.allclose(runs["y0"], plain["y0"], rtol=1e-13, atol=1e-13) and runs["yn"] == pytest.approx(plain["yn"], rel=1e-13)


def test_allocator_cache_is_bounded_by_the_peak():
    """Released device blocks are kept for the next setup of the same shape, but a process that has solved many differently sized
    systems must not end up holding all of HBM (that starved a child process of scratch memory in the round-3 suite): the cache never
    exceeds max(twice the peak in use, HDA_POOL_CACHE_MIN_GB), oldest blocks are returned first, and results do not depend on it."""
    import json
    import subprocess
    import sys
    code = """
import sys, json, numpy as np
sys.path.insert(0, %r)
import hypredrive_amd as hd
from hypredrive_amd import _lib
seen, its = [], []
for n in (24, 40, 28, 44, 32, 36, 48, 26, 30, 34, 38, 42, 46, 24):
    A = hd.lap7(n, n, n)
    amg = hd.Amg(A)
    b = np.zeros(n ** 3); b[:n * n] = 1.0
    r = hd.pcg(A, b, amg)
    its.append([n, int(r["iters"]), float(r["hist"][-1])])
    del A, amg
    iu, pk = _lib.memory_stats()
    seen.append([iu, pk, _lib.memory_cached()])
print(json.dumps(dict(seen=seen, its=its)))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for floor in ("0.001", "64"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HDA_POOL_CACHE_MIN_GB=floor), timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    small, big = outs
    assert small["its"] == big["its"]                                  # same iterations and residuals, bit for bit
    for iu, pk, cached in small["seen"]:
        assert cached <= max(2 * pk, 0.001 * 2 ** 30)
    assert small["seen"][-1][2] < big["seen"][-1][2]                   # without the bound the cache only grows
    tot_small = small["seen"][-1][0] + small["seen"][-1][2]
    assert tot_small <= 3 * small["seen"][-1][1] + 0.001 * 2 ** 30


def test_random_hierarchies_bit_identical_to_oracle():
    """tests/fuzz_hierarchies.py, 40 random operators (40-3000 rows, 2-70 entries a row, both signs, empty rows; theta 0.25-0.7, Pmax 0-6,
    ext+i / direct interpolation, 0-2 aggressive levels; every other seed since round 4: standard / mm-ext+i / direct / ext+i
    interpolation, PMIS or HMIS, Jacobi or a hybrid Gauss-Seidel pair, on 1 / 3 / 7 row blocks) through the whole setup under
    HDA_GUARD=1 HDA_POISON=1: every operator, C/F marker, block start and interpolation of every hierarchy bit-identical to the oracle's
    (240 of them ran clean in round 4, gpurun_out/r04fuzz)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_hierarchies.py"), "40"], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=root, HDA_GUARD="1", HDA_POISON="1"), timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_int32_size_guard(hd):
    """Operators are int32-indexed (HYPRE_Int in hypre's default build): a setup stage whose output
    would pass 2^31-1 entries must stop with an error, never wrap around."""
    from hypredrive_amd import _lib
    _lib.check_row_total(1 << 20, 2047)            # 2^31 - 2^20: fits
    with pytest.raises(_lib.LibraryError, match="int32"):
        _lib.check_row_total(1 << 20, 2048)        # exactly 2^31
    with pytest.raises(_lib.LibraryError, match="int32"):
        _lib.check_row_total(1 << 22, 1 << 20)


# ------------------------------------------------- ILU(0) (reference src/internal/ilu.c, amg.c:899-921)

def _ilu_cases(orc):
    import scipy.sparse as sp
    conv = sp.random(400, 400, density=0.02, random_state=11, format="csr")
    conv = (conv + sp.diags(np.asarray(abs(conv).sum(axis=1)).ravel() + np.asarray(abs(conv).sum(axis=0)).ravel() + 1.0)).tocsr()
    return [("lap7", orc.lap7(13, 11, 9, b_mode=1)[0].to_scipy()), ("spd", rand_spd(700, 0.01, 5)), ("nonsym", conv)]


def test_ilu0_factors_bit_exact(orc, hd):
    """Level-scheduled factorisation on the device = sequential IKJ ILU(0) of the oracle, bit for bit
    (updates reach an entry in ascending pivot order on both sides)."""
    for name, M in _ilu_cases(orc):
        Ao, Ah = both(orc, hd, M)
        Fo, Fh = orc.Ilu(Ao), hd.Ilu(Ah)
        lo, lh = Fo.factors.to_scipy(), Fh.factors.to_scipy()
        assert np.array_equal(lo.indptr, lh.indptr) and np.array_equal(lo.indices, lh.indices), name
        assert np.array_equal(lo.data, lh.data), name


@pytest.mark.parametrize("tri_solve,lower_it,upper_it", [(1, 5, 5), (0, 5, 5), (0, 1, 1), (0, 4, 7), (0, 6, 2)])
def test_ilu_apply_matches_oracle(orc, hd, tri_solve, lower_it, upper_it):
    for name, M in _ilu_cases(orc):
        Ao, Ah = both(orc, hd, M)
        r = np.random.default_rng(2).standard_normal(M.shape[0])
        zo = orc.Ilu(Ao, tri_solve=tri_solve, lower_it=lower_it, upper_it=upper_it).apply(r)
        zh = hd.Ilu(Ah, tri_solve=tri_solve, lower_it=lower_it, upper_it=upper_it).apply(r)
        assert rel(zh, zo) < RTOL_REDUCE, name


@pytest.mark.parametrize("tri_solve,max_iter", [(1, 1), (0, 1), (1, 2)])
def test_ilu_preconditioned_krylov_matches_oracle(orc, hd, tri_solve, max_iter):
    """'preconditioner: ilu' under PCG and GMRES: same iteration counts and residual histories as the oracle."""
    Ao, b = orc.lap7(14, 14, 14, b_mode=1)
    Ah = hd.lap7(14, 14, 14)
    po = orc.IluPrecond(Ao, max_iter=max_iter, tri_solve=tri_solve)
    ph = hd.Ilu(Ah, max_iter=max_iter, tri_solve=tri_solve)
    if max_iter == 1:  # M^-1 applied once is symmetric for a symmetric A; two residual corrections are too, but keep PCG to the plain case
        ro, rh = orc.pcg(Ao, b, po), hd.pcg(Ah, b, ph)
        assert rh["converged"] and rh["iters"] == ro["iters"]
        assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9)
    ro, rh = orc.gmres(Ao, b, po), hd.gmres(Ah, b, ph)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-8)
    assert rel(rh["x"], ro["x"]) < 1e-8


@pytest.mark.parametrize("levels,sweeps,tri_solve", [(1, 1, 1), (1, 2, 1), (2, 1, 0), (25, 1, 1)])
def test_amg_ilu_smoother_matches_oracle(orc, hd, levels, sweeps, tri_solve):
    """amg.smoother.type ilu (amg.c:899-921): ILU(0) replaces the relaxation on the first levels of an
    otherwise identical hierarchy; V-cycle, iteration count and history against the oracle."""
    Ao, b = orc.lap7(16, 15, 14, b_mode=1)
    Ah = hd.lap7(16, 15, 14)
    ao = orc.Amg(Ao, orc.amg_params(True))
    ao.set_ilu_smoother(num_levels=levels, num_sweeps=sweeps, tri_solve=tri_solve)
    ah = hd.Amg(Ah, hd.AmgParams.default(smooth_num_levels=levels, smooth_num_sweeps=sweeps, ilu_tri_solve=tri_solve))
    assert ah.num_levels == ao.num_levels
    Fo = orc.Ilu(ao.level_A(0))  # (keep the handle: factors is a borrowed view)
    lo, lh = Fo.factors.to_scipy(), ah.ilu_factors(0).to_scipy()
    assert np.array_equal(lo.data, lh.data)
    r = np.random.default_rng(4).standard_normal(Ao.nrows)
    assert rel(ah.vcycle(r), ao.vcycle(r)) < 1e-12
    ro, rh = orc.pcg(Ao, b, ao), hd.pcg(Ah, b, ah)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9)
    plain = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    assert ro["iters"] < plain["iters"]


def test_ilu_errors_are_loud(hd):
    import scipy.sparse as sp
    M = sp.csr_matrix(np.array([[0.0, 1.0, 0.0], [1.0, 2.0, 1.0], [0.0, 1.0, 2.0]]))
    M.eliminate_zeros()
    with pytest.raises(hd.LibraryError, match="diagonal"):
        hd.Ilu(hd.Csr.from_scipy(M))
    Z = sp.csr_matrix(np.array([[1.0, 1.0], [1.0, 1.0]]))  # second pivot 1 - 1*1 = 0
    with pytest.raises(hd.LibraryError, match="pivot"):
        hd.Ilu(hd.Csr.from_scipy(Z))


@pytest.mark.parametrize("precond", ["amg", "ilu", None])
def test_fgmres_and_bicgstab_match_oracle(orc, hd, precond):
    """solver_ops[SOLVER_FGMRES / SOLVER_BICGSTAB] (reference src/internal/solver.c:229-252): same iteration counts and
    residual histories as the oracle's restatements."""
    Ao, b = orc.lap7(13, 12, 11, b_mode=1)
    Ah = hd.lap7(13, 12, 11)
    po = {"amg": lambda: orc.Amg(Ao, orc.amg_params(True)), "ilu": lambda: orc.IluPrecond(Ao), None: lambda: None}[precond]()
    ph = {"amg": lambda: hd.Amg(Ah), "ilu": lambda: hd.Ilu(Ah), None: lambda: None}[precond]()
    ko, kh = orc.krylov_params(True, rtol=1e-9, krylov_dim=7), hd.KrylovParams.default(True, rtol=1e-9, krylov_dim=7)
    ro, rh = orc.fgmres(Ao, b, po, ko), hd.fgmres(Ah, b, ph, kh)   # krylov_dim 7: restarts included
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-7)
    assert rel(rh["x"], ro["x"]) < 1e-8
    ko, kh = orc.krylov_params(False, rtol=1e-9, max_iter=300), hd.KrylovParams.default(False, rtol=1e-9, max_iter=300)
    ro, rh = orc.bicgstab(Ao, b, po, ko), hd.bicgstab(Ah, b, ph, kh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-5 if precond is None else 1e-7)  # BiCGSTAB amplifies rounding over many steps
    assert rel(rh["x"], ro["x"]) < 1e-7


# ------------------------------------------------- MGR (reference src/internal/mgr.c)

def _three_field(n=10, seed=0):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_pins import three_field_system
    return three_field_system(n, seed)


MGR_CASES = [
    [dict(f_dofs=[2], prolongation_type="jacobi")],
    [dict(f_dofs=[2], prolongation_type="jacobi", restriction_type="jacobi")],
    [dict(f_dofs=[2], prolongation_type="l1-jacobi", restriction_type="columped", f_relaxation="l1-jacobi", f_sweeps=2)],
    [dict(f_dofs=[2])],
    [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")],   # ex3.yml
    # the same with the global relaxation on row blocks (round 5): hybrid l1 Gauss-Seidel as the reference computes it at np = 3 / 5
    [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped", g_blocks=3)],
    [dict(f_dofs=[2], prolongation_type="jacobi", g_relaxation="l1-hfgs", g_blocks=5), dict(f_dofs=[1], g_relaxation="h-bgs", g_sweeps=2, g_blocks=2)],
    [dict(f_dofs=[1, 2], prolongation_type="jacobi", g_relaxation="h-fgs", g_sweeps=2)],
    [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="ilu", restriction_type="columped")],       # ex4.yml / ex5-mgr.yml
    [dict(f_dofs=[0], prolongation_type="jacobi", f_relaxation="amg")],                                                      # AMG on A_FF (diffusive field eliminated first)
    [dict(f_dofs=[1, 2], prolongation_type="jacobi", f_relaxation="amg", f_sweeps=2)],
    [dict(f_dofs=[0], prolongation_type="jacobi", f_relaxation="ilu", ilu=dict(tri_solve=0, lower_jac_iters=3, upper_jac_iters=4))],  # ILU(0) of A_FF
    [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="ilu", ilu=dict(tri_solve=0), restriction_type="columped")],
]


def test_mgr_ilu_on_the_coarsest_level_matches_oracle(orc, hd):
    S, labels = _three_field(11, seed=3)
    Ao, Ah = both(orc, hd, S)
    lev = [dict(f_dofs=[1, 2], prolongation_type="jacobi", coarsest_ilu=dict(max_iter=2, tri_solve=1))]
    Mo, Mh = orc.MgrPrecond(Ao, labels, lev, coarsest="ilu"), hd.Mgr(Ah, labels, lev, coarsest="ilu")
    r = np.random.default_rng(8).standard_normal(S.shape[0])
    assert rel(Mh.vcycle(r), Mo.vcycle(r)) < 1e-12
    b = np.ones(S.shape[0])
    ro, rh = orc.gmres(Ao, b, Mo), hd.gmres(Ah, b, Mh)
    assert rh["converged"] and rh["iters"] == ro["iters"] and np.allclose(rh["hist"], ro["hist"], rtol=1e-8)


@pytest.mark.parametrize("levels", MGR_CASES)
def test_mgr_matches_oracle(orc, hd, levels):
    """Transfer operators and reduced operators bit-identical to the oracle (row kernels sum in column order, the
    Galerkin product is the AMG setup's deterministic SpGEMM); one MGR application to 1e-12; same GMRES iterations."""
    S, labels = _three_field(11, seed=3)
    Ao, Ah = both(orc, hd, S)
    Mo, Mh = orc.MgrPrecond(Ao, labels, levels), hd.Mgr(Ah, labels, levels)
    for l in range(len(levels)):
        for which in (1, 2):
            a, b_ = Mo.matrix(l, which).to_scipy(), Mh.matrix(l, which).to_scipy()
            assert np.array_equal(a.indptr, b_.indptr) and np.array_equal(a.indices, b_.indices) and np.array_equal(a.data, b_.data), (l, which)
        a, b_ = Mo.matrix(l + 1, 0).to_scipy(), Mh.matrix(l + 1, 0).to_scipy()
        assert np.array_equal(a.indptr, b_.indptr) and np.array_equal(a.indices, b_.indices) and np.array_equal(a.data, b_.data), l
    r = np.random.default_rng(6).standard_normal(S.shape[0])
    assert rel(Mh.vcycle(r), Mo.vcycle(r)) < 1e-12
    b = np.ones(S.shape[0])
    ro, rh = orc.gmres(Ao, b, Mo), hd.gmres(Ah, b, Mh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-8)
    assert rel(rh["x"], ro["x"]) < 1e-8


@pytest.mark.parametrize("case", ["frelax_gmres_amg", "coarse_gmres_amg", "frelax_gmres_plain", "both_fgmres"])
def test_mgr_nested_krylov_matches_oracle(orc, hd, case):
    """Nested Krylov components of MGR (reference src/internal/krylov.c; examples/ex3-mgr_Frelax_gmres.yml: GMRES(5, tol 1e-15)
    preconditioned by a one-level BoomerAMG as F-relaxation of the second reduction level; ex3-mgr_coarse_gmres_amg.yml: two
    GMRES iterations preconditioned by BoomerAMG on the coarsest system).  One MGR application against the oracle's, and the
    outer FlexGMRES (the preconditioner is no longer a fixed operator) with the oracle's iteration count.  Parity unpinned:
    compflow6k is not in the tree."""
    S, labels = _three_field(11, seed=3)
    Ao, Ah = both(orc, hd, S)

    def levels(lib, amgp):
        one = amgp(max_levels=1, relax_coarse=18)  # coarsening.max_levels: 1 -- the smoother alone
        l0 = dict(f_dofs=[2], prolongation_type="jacobi")
        l1 = dict(f_dofs=[1], restriction_type="columped")
        if case == "frelax_gmres_amg":
            l1.update(f_relaxation="amg", f_amg=one, f_krylov=dict(method="gmres", max_iter=5, rtol=1e-15))
        elif case == "coarse_gmres_amg":
            l1.update(f_relaxation="amg", f_amg=one, coarsest_krylov=dict(method="gmres", max_iter=2, rtol=0.0))
        elif case == "frelax_gmres_plain":
            l1.update(f_relaxation="amg", f_amg=one, f_krylov=dict(method="gmres", max_iter=3, rtol=1e-15, precond=False))
        else:
            l1.update(f_relaxation="amg", f_amg=one, f_krylov=dict(method="fgmres", max_iter=2, rtol=0.0),
                      coarsest_krylov=dict(method="fgmres", max_iter=2, rtol=0.0))
        return [l0, l1]

    Mo = orc.MgrPrecond(Ao, labels, levels(orc, lambda **kw: orc.amg_params(True, **kw)))
    Mh = hd.Mgr(Ah, labels, levels(hd, lambda **kw: hd.AmgParams.default(**kw)))
    r = np.random.default_rng(6).standard_normal(S.shape[0])
    assert rel(Mh.vcycle(r), Mo.vcycle(r)) < 1e-9
    b = np.ones(S.shape[0])
    ro, rh = orc.fgmres(Ao, b, Mo), hd.fgmres(Ah, b, Mh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-6)
    # the nested solve does something: the plain components give another preconditioner
    plain = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], restriction_type="columped", f_relaxation="amg",
                                                                 f_amg=hd.AmgParams.default(max_levels=1, relax_coarse=18))]
    assert rel(hd.Mgr(Ah, labels, plain).vcycle(r), Mh.vcycle(r)) > 1e-6


@pytest.mark.parametrize("outer", ["pcg", "gmres"])
def test_nested_pcg_inside_an_outer_krylov_matches_oracle(orc, hd, outer):
    """A PCG nested inside the preconditioner call of an outer PCG (YAML `f_relaxation: pcg:` / `coarsest_level: pcg:` under
    `solver: pcg`): the outer recurrence keeps <r,r> block partials and the previous gamma on the device across that call
    (fused dots), the nested solve uses the same scalar block -- it saves and restores it (hda_krylov.hip NestedScope).  Without
    that the outer search directions and stopping tests are silently wrong.  Compared step by step with the oracle, whose
    nested solve has its own scalars by construction; a fixed number of outer iterations, convergence not required (the
    preconditioner is not a fixed operator)."""
    S, labels = _three_field(11, seed=3)
    Ao, Ah = both(orc, hd, S)

    def levels(amgp):
        one = amgp(max_levels=1, relax_coarse=18)
        return [dict(f_dofs=[2], prolongation_type="jacobi"),
                dict(f_dofs=[1], restriction_type="columped", f_relaxation="amg", f_amg=one,
                     f_krylov=dict(method="pcg", max_iter=3, rtol=0.0), coarsest_krylov=dict(method="pcg", max_iter=2, rtol=0.0))]

    Mo = orc.MgrPrecond(Ao, labels, levels(lambda **kw: orc.amg_params(True, **kw)))
    Mh = hd.Mgr(Ah, labels, levels(lambda **kw: hd.AmgParams.default(**kw)))
    b = np.ones(S.shape[0])
    if outer == "pcg":
        ro = orc.pcg(Ao, b, Mo, orc.krylov_params(False, rtol=1e-8, max_iter=8))
        rh = hd.pcg(Ah, b, Mh, hd.KrylovParams.default(False, rtol=1e-8, max_iter=8))
    else:
        ro, rh = orc.fgmres(Ao, b, Mo), hd.fgmres(Ah, b, Mh)
    assert rh["iters"] == ro["iters"] and rh["iters"] >= 3
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-6)
    assert rel(rh["x"], ro["x"]) < 1e-6


@pytest.mark.parametrize("cyc", ["v(0,1)", "v(1,1)", "w", "w(1,1)"])
def test_mgr_cycle_shapes_match_oracle(orc, hd, cyc):
    """`mgr.cycle` (reference src/internal/mgr.c:614-675): smoothing after the coarse correction, on both sides of it, and W-cycles
    (the coarser level visited twice), with global relaxation and F-relaxation on both reduction levels."""
    S, labels = _three_field(11, seed=3)
    Ao, Ah = both(orc, hd, S)
    lev = [dict(f_dofs=[2], prolongation_type="jacobi", g_relaxation="l1-hsgs"),
           dict(f_dofs=[1], g_relaxation="h-fgs", restriction_type="columped", f_relaxation="l1-jacobi", f_sweeps=2, cycle=cyc)]
    Mo, Mh = orc.MgrPrecond(Ao, labels, lev), hd.Mgr(Ah, labels, lev)
    r = np.random.default_rng(6).standard_normal(S.shape[0])
    zh, zo = Mh.vcycle(r), Mo.vcycle(r)
    assert rel(zh, zo) < 1e-11
    base = [dict(lev[0]), {k: v for k, v in lev[1].items() if k != "cycle"}]
    assert rel(hd.Mgr(Ah, labels, base).vcycle(r), zh) > 1e-6  # not the default v(1,0) cycle
    b = np.ones(S.shape[0])
    ro, rh = orc.gmres(Ao, b, Mo), hd.gmres(Ah, b, Mh)
    assert rh["converged"] and rh["iters"] == ro["iters"] and np.allclose(rh["hist"], ro["hist"], rtol=1e-7)


def test_mgr_unimplemented_options_fail_loudly(hd):
    S, labels = _three_field(6)
    Ah = hd.Csr.from_scipy(S)
    with pytest.raises(hd.LibraryError, match="f_dofs"):
        hd.Mgr(Ah, labels, [dict(f_dofs=[7])])           # label that does not occur: nothing to eliminate
    with pytest.raises(hd.LibraryError, match="f_dofs"):
        hd.Mgr(Ah, labels, [dict(f_dofs=[0, 1, 2])])     # everything eliminated: no coarse system


# ------------------------------------------------- Chebyshev smoother (relax type 16; reference src/internal/cheby.c, amg.c:886-890)

@pytest.mark.parametrize("order,eig_est,scale,fraction", [(2, 10, 1, 0.3), (4, 10, 1, 0.1), (3, 0, 1, 0.3), (1, 5, 0, 0.3), (2, 10, 0, 0.2)])
def test_amg_chebyshev_smoother_matches_oracle(orc, hd, order, eig_est, scale, fraction):
    """AMG with the Chebyshev smoother on every level: same hierarchy, V-cycle to 1e-10 (the eigenvalue estimates are dot
    products reduced in a different order), identical PCG iteration counts and histories."""
    Ao, b = orc.lap7(15, 14, 13, b_mode=1)
    Ah = hd.lap7(15, 14, 13)
    kw = dict(relax_down=16, relax_up=16, cheby_order=order, cheby_eig_est=eig_est, cheby_scale=scale, cheby_fraction=fraction)
    ao, ah = orc.Amg(Ao, orc.amg_params(True, **kw)), hd.Amg(Ah, hd.AmgParams.default(**kw))
    assert ah.num_levels == ao.num_levels
    r = np.random.default_rng(12).standard_normal(Ao.nrows)
    assert rel(ah.vcycle(r), ao.vcycle(r)) < 1e-10
    ro, rh = orc.pcg(Ao, b, ao), hd.pcg(Ah, b, ah)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-8)
    if order >= 2:
        assert ro["iters"] < orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))["iters"]


def test_chebyshev_variant_1_is_refused(hd):
    Ah = hd.lap7(6, 6, 6)
    with pytest.raises(hd.LibraryError, match="variant 0"):
        hd.Amg(Ah, hd.AmgParams.default(relax_down=16, relax_up=16, cheby_variant=1))


@pytest.mark.parametrize("kind", ["amg-pcg", "amg-gmres", "mgr-gmres", "ilu-bicgstab", "cheby-pcg"])
def test_runs_are_bitwise_reproducible(hd, kind):
    """Two independent setups + solves of the same problem give bit-identical residual histories and solutions: reductions
    are fixed trees over block partials, setup kernels have a fixed accumulation order, nothing uses floating-point atomics."""
    def once():
        if kind == "mgr-gmres":
            S, labels = _three_field(12, seed=2)
            A = hd.Csr.from_scipy(S)
            M = hd.Mgr(A, labels, [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")])
            return hd.gmres(A, np.ones(S.shape[0]), M)
        A = hd.lap7(20, 19, 18)
        b = np.cos(np.arange(A.dims[0], dtype=np.float64))
        if kind == "amg-pcg":
            return hd.pcg(A, b, hd.Amg(A))
        if kind == "amg-gmres":
            return hd.gmres(A, b, hd.Amg(A), hd.KrylovParams.default(True, krylov_dim=5))
        if kind == "cheby-pcg":
            return hd.pcg(A, b, hd.Amg(A, hd.AmgParams.default(relax_down=16, relax_up=16)))
        return hd.bicgstab(A, b, hd.Ilu(A, tri_solve=0))
    r1, r2 = once(), once()
    assert r1["iters"] == r2["iters"] and np.array_equal(r1["hist"], r2["hist"]) and np.array_equal(r1["x"], r2["x"])
