"""GPU parity tests of the row-block forms: V contiguous row blocks on one GPU = what the reference computes at np = V
(its CPU-build defaults, /root/reference/src/internal/amg.c:141-146 HMIS and :182-189 hybrid l1 Gauss-Seidel 13 / 14, are
rank-block algorithms; pins examples/refOutput/ex1.txt:27, laplacian.txt:34-38 were made with them at np 1).

Bars: C/F splittings, block starts and l1 divisors bit-exact; sweeps 1e-13 (wave-parallel row sums); identical iteration counts
and residual histories to 1e-10 on identical hierarchies.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

GS_TYPES = [3, 4, 6, 13, 14, 8]


@pytest.fixture(scope="module")
def hd():
    import hypredrive_amd as h
    assert h.device_count() >= 1, "no HIP device"
    return h


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def mmatrix(n, density, seed, unsym=False):
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="csr")
    if not unsym:
        M = M + M.T
    M = sp.csr_matrix(M)
    M.data = -np.abs(M.data)
    M.setdiag(0)
    M.eliminate_zeros()
    d = np.asarray(abs(M).sum(axis=1)).ravel() * rng.uniform(0.7, 1.3, n) + 0.05
    return (M + sp.diags(d)).tocsr()


def parts_for(n, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "one":
        return np.array([0, n])
    if kind == "even4":
        return np.array([(q * n) // 4 for q in range(5)])
    if kind == "even37":
        return np.array([(q * n) // 37 for q in range(38)])
    if kind == "ragged":  # blocks of very different sizes, two of them empty
        cuts = np.sort(rng.choice(np.arange(1, n), size=min(9, n - 1), replace=False))
        p = np.concatenate([[0], cuts[:3], [cuts[3], cuts[3]], cuts[4:], [n, n]])
        return np.sort(p)
    if kind == "rows":  # every row its own block: the hybrid sweep is a Jacobi sweep with the option-4 divisor
        return np.arange(n + 1)
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["one", "even4", "even37", "ragged", "rows"])
def test_block_l1_divisors_bit_exact(orc, hd, kind):
    """hypre_ParCSRComputeL1Norms option 4 with the other blocks in the role of the off-processor part"""
    for M in (mmatrix(900, 0.01, 3), mmatrix(400, 0.12, 4), orc.lap7(12, 9, 7)[0].to_scipy()):
        Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
        part = parts_for(M.shape[0], kind, 5)
        for option in (1, 4):
            assert np.array_equal(Ah.l1_norms_blocks(option, part), orc.l1_norms_blocks(Ao, option, part))


@pytest.mark.parametrize("rtype", GS_TYPES)
@pytest.mark.parametrize("kind", ["one", "even4", "even37", "ragged", "rows"])
@pytest.mark.parametrize("sweep_order_copy", [False, True])
def test_block_hybrid_gauss_seidel_matches_oracle(orc, hd, monkeypatch, rtype, kind, sweep_order_copy):
    """Gauss-Seidel inside a block, Jacobi across blocks: every block's workgroup reproduces the sequential sweep of its rows on
    the values the other blocks held when the sweep began.  sweep_order_copy: the kernel big levels get (operator, iterate,
    right-hand side and divisors copied into sweep order, HDA_GS_SORTED_MIN rows and more), forced on these small operators."""
    monkeypatch.setenv("HDA_GS_SORTED_MIN", "0" if sweep_order_copy else "1000000000")
    cases = [(orc.lap7(13, 11, 9, b_mode=1)[0].to_scipy(), 1.0), (mmatrix(1200, 0.008, 11), 0.9), (mmatrix(300, 0.2, 12), 1.0),
             (mmatrix(500, 0.03, 13, unsym=True), 1.0)]
    for M, w in cases:
        n = M.shape[0]
        Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
        part = parts_for(n, kind, 7)
        rng = np.random.default_rng(rtype)
        b, x0 = rng.standard_normal(n), rng.standard_normal(n)
        l1 = orc.l1_norms_blocks(Ao, 4, part)
        xo = x0
        for _ in range(2):
            xo = orc.relax_blocks(Ao, l1, rtype, w, b, xo, part)
        assert rel(Ah.relax_blocks(b, x0, part, rtype, w, sweeps=2), xo) < 1e-12


@pytest.mark.parametrize("kind", ["one", "even4", "ragged"])
def test_block_gauss_seidel_on_long_rows(orc, hd, monkeypatch, kind):
    """Rows of several hundred entries: the barrier-free kernel gives a row 64 lanes and, beyond 256 entries, two chunks per lane;
    rows longer still leave it to the ring / sorted kernels.  Forward, backward and symmetric sweeps against the oracle."""
    monkeypatch.setenv("HDA_GS_SORTED_MIN", "0")
    monkeypatch.setenv("HDA_GS_FREE_CHECK", "1")
    for n, dens in ((500, 0.35), (640, 0.6), (900, 0.75)):
        M = mmatrix(n, dens, 40 + n)
        assert np.diff(M.indptr).max() > 4 * 40
        Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
        part = parts_for(n, kind, 3)
        rng = np.random.default_rng(n)
        b, x0 = rng.standard_normal(n), rng.standard_normal(n)
        l1 = orc.l1_norms_blocks(Ao, 4, part)
        for rtype in (13, 14, 8):
            xo = orc.relax_blocks(Ao, l1, rtype, 1.0, b, x0, part)
            assert rel(Ah.relax_blocks(b, x0, part, rtype, 1.0), xo) < 1e-12, (n, rtype)


@pytest.mark.parametrize("case", ["five_point_200x600", "tridiagonal"])
def test_block_gauss_seidel_short_rows_on_blocks_beyond_16384_rows(orc, hd, monkeypatch, case):
    """Round-4 ADVICE (high): rows of <= 4 off-diagonal entries get one lane each -- G = 512 groups -- and on blocks of more than
    16 384 rows the ring was sized for the dependency reach alone (1024 slots), which broke the guard's invariant
    KG G <= RING - RING / 4 - G - 1: round 0 waited for a progress nobody could make and the sweep ended in its spin limit.  The plan
    now grows the ring (2048 for one lane per row), the kernel asserts the invariant, and the sweep equals the oracle's."""
    monkeypatch.setenv("HDA_GS_SORTED_MIN", "0")
    monkeypatch.setenv("HDA_GS_FREE_CHECK", "1")
    if case == "tridiagonal":
        n = 70000
        M = sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
        parts = [np.array([0, n]), np.array([0, 20000, 50000, n])]
    else:
        nx, ny = 200, 600
        T = lambda m: sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], [-1, 0, 1])
        M = (sp.kron(sp.eye(ny), T(nx)) + sp.kron(T(ny), sp.eye(nx))).tocsr()
        n = nx * ny
        parts = [np.array([0, n]), np.array([0, 40000, 80000, n])]
    Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
    rng = np.random.default_rng(5)
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    for part in parts:
        assert np.diff(part).max() > 16384
        l1 = orc.l1_norms_blocks(Ao, 4, part)
        for rtype in (13, 14):
            xo = orc.relax_blocks(Ao, l1, rtype, 1.0, b, x0, part)
            assert rel(Ah.relax_blocks(b, x0, part, rtype, 1.0), xo) < 1e-12, (case, list(part), rtype)


@pytest.mark.parametrize("dep_copy", ["1", "0"])
def test_block_gauss_seidel_rows_beyond_the_lanes_capacity(orc, hd, monkeypatch, dep_copy):
    """Round 5: the barrier-free kernel sizes a row's lanes for all but 0.3 % of the rows; the longer ones read their further chunks
    inside the update stage.  A five-point operator (one chunk per row) in which one row in five hundred is coupled to 40 more unknowns,
    near and far, earlier and later in the sweep, inside and outside its block.  dep_copy: forward sweeps from the zero guess on the
    copy of the in-block earlier columns alone (the default) or on the whole operator."""
    monkeypatch.setenv("HDA_GS_DEP", dep_copy)
    monkeypatch.setenv("HDA_GS_SORTED_MIN", "0")
    monkeypatch.setenv("HDA_GS_FREE", "1")
    monkeypatch.setenv("HDA_GS_FREE_CHECK", "1")
    nx, ny = 150, 160
    T = lambda m: sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], [-1, 0, 1])
    M = (sp.kron(sp.eye(ny), T(nx)) + sp.kron(T(ny), sp.eye(nx))).tolil()
    n = nx * ny
    rng = np.random.default_rng(11)
    for i in rng.choice(n, n // 500, replace=False):
        near = np.clip(i + rng.integers(-300, 300, 20), 0, n - 1)
        far = rng.integers(0, n, 20)
        for j in np.concatenate([near, far]):
            if j != i:
                M[i, j] = M[j, i] = -0.05
    M = M.tocsr()
    M.setdiag(np.asarray(abs(M).sum(axis=1)).ravel() + 0.5)
    M = M.tocsr(); M.sort_indices()
    Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    for part in ([0, n], [0, 5000, 11000, 17000, n]):
        l1 = orc.l1_norms_blocks(Ao, 4, np.array(part))
        for rtype in (13, 14):
            xo = orc.relax_blocks(Ao, l1, rtype, 1.0, b, x0, np.array(part))
            assert rel(Ah.relax_blocks(b, x0, np.array(part), rtype, 1.0), xo) < 1e-12, (part, rtype)
            xz = orc.relax_blocks(Ao, l1, rtype, 1.0, b, np.zeros(n), np.array(part))
            assert rel(Ah.relax_blocks(b, np.zeros(n), np.array(part), rtype, 1.0), xz) < 1e-12, (part, rtype, "zero guess")


def test_one_block_is_the_sequential_sweep(orc, hd):
    M = mmatrix(800, 0.01, 21)
    Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
    rng = np.random.default_rng(1)
    b, x0 = rng.standard_normal(800), rng.standard_normal(800)
    for rtype in (13, 14, 8):
        seq = orc.relax(Ao, orc.l1_norms(Ao, 4), rtype, 1.0, b, x0)
        assert np.array_equal(orc.relax_blocks(Ao, orc.l1_norms(Ao, 4), rtype, 1.0, b, x0, [0, 800]), seq)
        assert rel(Ah.relax_blocks(b, x0, [0, 800], rtype, 1.0), seq) < 1e-13
        assert rel(Ah.relax(b, x0, rtype, 1.0), seq) < 1e-13


@pytest.mark.parametrize("form", ["lds", "global"])
@pytest.mark.parametrize("kind", ["one", "even4", "even37", "ragged"])
@pytest.mark.parametrize("theta", [0.25, 0.6])
def test_block_hmis_bit_exact(orc, hd, kind, theta, form, monkeypatch):
    """hypre_BoomerAMGCoarsenHMIS at np = V: Ruge first pass per block, interior C points kept, PMIS from there.
    form: the first pass's bucket heads / tails in LDS (default) or in global memory (what in-degrees beyond 4095 get; HDA_RS_LDS=0
    forces it here)."""
    if form == "global":
        monkeypatch.setenv("HDA_RS_LDS", "0")
    for M in (orc.lap7(14, 12, 10)[0].to_scipy(), mmatrix(1500, 0.006, 31), mmatrix(600, 0.05, 32), mmatrix(700, 0.02, 33, unsym=True)):
        Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
        part = parts_for(M.shape[0], kind, 9)
        sm = orc.strength(Ao, theta)
        assert np.array_equal(Ah.strength(theta), sm)
        assert np.array_equal(Ah.hmis_blocks(sm, part), orc.hmis_blocks(Ao, sm, part))


@pytest.mark.parametrize("shape,V", [((12, 12, 12), 1), ((16, 16, 16), 4), ((20, 18, 16), 7), ((24, 24, 24), 12), ((16, 16, 16), 64)])
def test_block_hierarchy_and_pcg_match_oracle(orc, hd, shape, V):
    """the reference's CPU defaults (HMIS, hybrid l1 Gauss-Seidel 13 / 14) on V row blocks: same block starts, C/F splittings and
    operators on every level, PCG with the oracle's iterations and history"""
    Ao, b = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    po = orc.amg_params(False, blocks=V)
    ph = hd.AmgParams.default(coarsen_type=10, relax_down=13, relax_up=14, relax_coarse=9, blocks=V)
    ho, hh = orc.Amg(Ao, po), hd.Amg(Ah, ph)
    assert hh.num_levels == ho.num_levels and hh.blocks == max(V, 1)
    for l in range(ho.num_levels):
        if V > 1:
            assert np.array_equal(hh.level_blocks(l), ho.level_block_part(l)), f"block starts level {l}"
        if l < ho.num_levels - 1:
            assert np.array_equal(hh.level_cf(l), ho.level_cf(l)), f"C/F level {l}"
            rp, cj, v = hh.level_matrix(l + 1, 0).download()
            Al = ho.level_A(l + 1)
            assert np.array_equal(rp, Al.rowptr) and np.array_equal(cj, Al.col) and np.array_equal(v, Al.val)
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)
    assert rel(rh["x"], ro["x"]) < 1e-9


def test_block_part_given_by_the_caller(orc, hd):
    """uneven blocks named by their row starts (what a reference run on ranks of different sizes computes)"""
    shape = (14, 14, 14)
    Ao, b = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    n = Ao.nrows
    part = np.array([0, 300, 301, 1500, 1500, n])
    po = orc.amg_params(False, blocks=5, block_part=part)
    ph = hd.AmgParams.default(coarsen_type=10, relax_down=13, relax_up=14, relax_coarse=9, blocks=5, block_part=part)
    ho, hh = orc.Amg(Ao, po), hd.Amg(Ah, ph)
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels - 1):
        assert np.array_equal(hh.level_blocks(l), ho.level_block_part(l))
        assert np.array_equal(hh.level_cf(l), ho.level_cf(l))
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["iters"] == ro["iters"] and np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("down,up", [(3, 4), (6, 6), (8, 8)])
def test_block_sweeps_with_pmis_grids(orc, hd, down, up):
    """row blocks with the GPU build's coarsening (PMIS is block independent): only the sweeps and their divisors change"""
    Ao, b = orc.lap7(18, 18, 18, b_mode=1)
    Ah = hd.lap7(18, 18, 18)
    ro = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True, relax_down=down, relax_up=up, blocks=9)))
    rh = hd.pcg(Ah, b, hd.Amg(Ah, hd.AmgParams.default(relax_down=down, relax_up=up, blocks=9)))
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9)


def test_blocks_stay_within_one_iteration_of_one_block_at_64_cubed(orc, hd):
    """Blocks four grid planes thick (what the setup chooses by itself above HDA_BLOCKS_MIN_ROWS rows: blocks of four times the
    operator's bandwidth) keep the CPU-default preconditioner within one PCG iteration of its one-block form; device = oracle(V)."""
    n = 64
    Ao, b = orc.lap7(n, n, n)
    Ah = hd.lap7(n, n, n)
    iters = {}
    for V in (1, 16):
        ph = hd.AmgParams.default(coarsen_type=10, relax_down=13, relax_up=14, relax_coarse=9, blocks=V)
        rh = hd.pcg(Ah, b, hd.Amg(Ah, ph))
        ro = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(False, blocks=V)))
        assert rh["converged"] and rh["iters"] == ro["iters"], V
        assert np.allclose(rh["hist"], ro["hist"], rtol=1e-9, atol=0)
        iters[V] = rh["iters"]
    assert abs(iters[16] - iters[1]) <= 1, iters


def test_automatic_blocks(hd, monkeypatch):
    """blocks = 0: one block up to HDA_BLOCKS_MIN_ROWS rows, beyond that blocks of at least four times the bandwidth"""
    A = hd.lap7(40, 40, 40)
    p = hd.AmgParams.default(coarsen_type=10, relax_down=13, relax_up=14, relax_coarse=9, blocks=0)
    assert hd.Amg(A, p).blocks == 1  # 64 000 rows: the sequential algorithms
    B = hd.lap7(96, 96, 96)  # 884 736 rows, bandwidth 9216: 24 blocks of four planes
    h = hd.Amg(B, p)
    assert h.blocks == 24
    part = h.level_blocks(0)
    assert part[0] == 0 and part[-1] == B.nrows and np.all(np.diff(part) == 96 * 96 * 4)
    # Jacobi smoothing + PMIS never asks for blocks
    assert hd.Amg(B, hd.AmgParams.default(blocks=0)).blocks == 1


# ---- mm-ext+i (interpolation type 17) as its own operator, from sparse products ---------------------------------------------------

@pytest.mark.parametrize("pmax,tf", [(4, 0.0), (0, 0.0), (3, 0.2)])
def test_mm_extpi_bit_exact(orc, hd, pmax, tf):
    """hypre's mm-ext+i (/root/reference/src/internal/amg.c:267-268; every pinned variant of examples/refOutput/ex8.txt:26-78):
    W = -D^-1 (I + B) A^s_FC on the deterministic SpGEMM, then InterpTruncation -- pattern and weights bit for bit the oracle's"""
    mats = [orc.lap7(12, 10, 9)[0].to_scipy(), mmatrix(900, 0.01, 41), mmatrix(500, 0.04, 42), mmatrix(600, 0.02, 43, unsym=True)]
    M = mmatrix(400, 0.05, 44).tolil()
    M[7, :] = 0
    M[7, 7] = 1.0  # a row without connections (special F)
    mats.append(M.tocsr())
    for M in mats:
        Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
        for theta in (0.25, 0.6):
            sm = orc.strength(Ao, theta)
            for cf in (orc.pmis(Ao, sm), orc.hmis_blocks(Ao, sm, [0, Ao.nrows])):
                Po, Ph = orc.interp_mm_extpi(Ao, sm, cf, pmax, tf), Ah.interp_mm_extpi(sm, cf, pmax, tf)
                rp, cj, v = Ph.download()
                assert np.array_equal(rp, Po.rowptr) and np.array_equal(cj, Po.col)
                assert np.array_equal(v, Po.val)


@pytest.mark.parametrize("shape,coarsen", [((12, 12, 12), 10), ((20, 16, 14), 8), ((24, 24, 24), 10)])
def test_mm_extpi_hierarchy_and_pcg_match_oracle(orc, hd, shape, coarsen):
    Ao, b = orc.lap7(*shape)
    Ah = hd.lap7(*shape)
    po = orc.amg_params(coarsen == 8, coarsen_type=coarsen, interp_type=17)
    ph = hd.AmgParams.default(coarsen_type=coarsen, interp_type=17, relax_down=po.relax_down, relax_up=po.relax_up, relax_coarse=9)
    ho, hh = orc.Amg(Ao, po), hd.Amg(Ah, ph)
    assert hh.num_levels == ho.num_levels
    for l in range(ho.num_levels - 1):
        assert np.array_equal(hh.level_cf(l), ho.level_cf(l))
        for which, Mo in ((1, ho.level_P(l)), (0, ho.level_A(l + 1))):
            rp, cj, v = hh.level_matrix(l + (0 if which else 1), which).download()
            assert np.array_equal(rp, Mo.rowptr) and np.array_equal(cj, Mo.col) and np.array_equal(v, Mo.val)
    ro, rh = orc.pcg(Ao, b, ho), hd.pcg(Ah, b, hh)
    assert rh["converged"] and rh["iters"] == ro["iters"]
    assert np.allclose(rh["hist"], ro["hist"], rtol=1e-10, atol=0)
    # the operator differs from classical extended+i (type 6): same C/F splitting on level 0, other weights
    h6 = hd.Amg(Ah, hd.AmgParams.default(coarsen_type=coarsen, interp_type=6, relax_down=po.relax_down, relax_up=po.relax_up, relax_coarse=9))
    assert np.array_equal(h6.level_cf(0), hh.level_cf(0))


# ---- block-Jacobi ILU(0) on row blocks: bj-iluk at np = V (reference src/internal/ilu.c:63-115; hypre factors every rank's diagonal block)


@pytest.mark.parametrize("form", ["plain", "sorted", "ring"])
@pytest.mark.parametrize("kind", ["even4", "even37", "ragged"])
def test_block_ilu_factors_bit_exact_and_exact_substitutions_match(orc, hd, monkeypatch, kind, form):
    """ILU(0) of the diagonal blocks of a row partition: the factors carry the oracle's bits (entries leaving a block dropped, updates
    in ascending pivot order), and the exact substitutions -- two zero-guess block sweeps over LU, forward with unit divisors and
    backward with 1 / u_ii -- give the oracle's z = U^-1 L^-1 r to 1e-13 on every form of the block kernel."""
    monkeypatch.setenv("HDA_GS_SORTED_MIN", "100000000" if form == "plain" else "0")
    monkeypatch.setenv("HDA_GS_RING", "1" if form == "ring" else "0")
    cases = [orc.lap7(13, 11, 9, b_mode=1)[0].to_scipy(), mmatrix(900, 0.01, 3), mmatrix(700, 0.012, 4, unsym=True)]
    for M in cases:
        n = M.shape[0]
        part = parts_for(n, kind, seed=n)
        Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
        Fo, Fh = orc.Ilu(Ao, part=part), hd.Ilu(Ah, block_part=part)
        assert Fh.blocks == len(part) - 1
        lo, lh = Fo.factors.to_scipy(), Fh.factors.to_scipy()
        assert np.array_equal(lo.indptr, lh.indptr) and np.array_equal(lo.indices, lh.indices)
        assert np.array_equal(lo.data, lh.data)
        # block-diagonal pattern: no entry leaves its block
        blk = np.searchsorted(part, np.arange(n), side="right")
        rows = np.repeat(np.arange(n), np.diff(lh.indptr))
        assert np.all(blk[rows] == blk[lh.indices])
        r = np.random.default_rng(7).standard_normal(n)
        assert rel(Fh.apply(r), Fo.apply(r)) < 1e-13
        # and it IS the exact solve with the block factors
        L = sp.tril(lh, -1).tocsr() + sp.identity(n)
        U = sp.triu(lh, 0).tocsr()
        assert rel((L @ (U @ Fh.apply(r))), r) < 1e-11


def test_block_ilu_even_split_and_jacobi_iterations(orc, hd):
    """blocks = V without starts = hypre's even split floor(q n / V); the Jacobi-iterative substitutions (tri_solve 0) run on the
    block factors through the streaming kernels as before."""
    M = orc.lap7(12, 12, 12, b_mode=1)[0].to_scipy()
    n = M.shape[0]
    Ao, Ah = orc.Csr.from_scipy(M), hd.Csr.from_scipy(M)
    part = np.array([(q * n) // 5 for q in range(6)])
    r = np.random.default_rng(1).standard_normal(n)
    for ts in (1, 0):
        Fo, Fh = orc.Ilu(Ao, part=part, tri_solve=ts), hd.Ilu(Ah, blocks=5, tri_solve=ts)
        assert Fh.blocks == 5 and np.array_equal(Fo.factors.to_scipy().data, Fh.factors.to_scipy().data)
        assert rel(Fh.apply(r), Fo.apply(r)) < 1e-13


@pytest.mark.parametrize("max_iter", [1, 2])
def test_block_ilu_preconditioned_krylov_matches_oracle(orc, hd, max_iter):
    """'preconditioner: ilu' on four row blocks under PCG / GMRES: the oracle's iteration counts and histories (the oracle at np = 4)."""
    Ao, b = orc.lap7(16, 15, 14, b_mode=1)
    Ah = hd.lap7(16, 15, 14)
    n = Ao.nrows
    part = np.array([(q * n) // 4 for q in range(5)])
    po, ph = orc.IluPrecond(Ao, max_iter=max_iter, part=part), hd.Ilu(Ah, max_iter=max_iter, blocks=4)
    if max_iter == 1:
        ro, rh = orc.pcg(Ao, b, po), hd.pcg(Ah, b, ph)
        assert rh["converged"] and rh["iters"] == ro["iters"] and np.allclose(rh["hist"], ro["hist"], rtol=1e-9)
        one = orc.pcg(Ao, b, orc.IluPrecond(Ao, max_iter=1))
        assert ro["iters"] >= one["iters"]  # dropping the couplings between blocks does not help
    ro, rh = orc.gmres(Ao, b, po), hd.gmres(Ah, b, ph)
    assert rh["converged"] and rh["iters"] == ro["iters"] and np.allclose(rh["hist"], ro["hist"], rtol=1e-8)


def test_amg_ilu_smoother_on_the_hierarchy_blocks(orc, hd):
    """BoomerAMG's complex smoother on a hierarchy with row blocks: the ILU of a level is block-Jacobi over THAT level's blocks
    (at np = V the reference's smoother factors every rank's diagonal block), coarse levels through their C points."""
    Ao, b = orc.lap7(18, 16, 14, b_mode=1)
    Ah = hd.lap7(18, 16, 14)
    kw = dict(coarsen_type=10, strong_th=0.5, relax_down=13, relax_up=14, blocks=4)
    ao = orc.Amg(Ao, orc.amg_params(False, **kw))
    ao.set_ilu_smoother(num_levels=2, num_sweeps=1, part=ao.level_block_part(0))
    ah = hd.Amg(Ah, hd.AmgParams.default(smooth_num_levels=2, smooth_num_sweeps=1, relax_coarse=9, **kw))
    assert ah.num_levels == ao.num_levels
    for l in (0, 1):
        Fo = orc.Ilu(ao.level_A(l), part=ao.level_block_part(l))
        assert np.array_equal(Fo.factors.to_scipy().data, ah.ilu_factors(l).to_scipy().data)
    r = np.random.default_rng(4).standard_normal(Ao.nrows)
    assert rel(ah.vcycle(r), ao.vcycle(r)) < 1e-12
    ro, rh = orc.pcg(Ao, b, ao), hd.pcg(Ah, b, ah)
    assert rh["converged"] and rh["iters"] == ro["iters"] and np.allclose(rh["hist"], ro["hist"], rtol=1e-9)


def test_row_block_announcement(tmp_path):
    """Round-4 review (weak #2) and ADVICE: from 100 000 rows the rank-block algorithms (HMIS, hybrid Gauss-Seidel) run on V row blocks =
    the reference at np = V, not np = 1 -- the setup must SAY so, whatever the print level; HDA_QUIET=1 silences the line; and
    `preconditioner: ilu` keeps the whole matrix in one block unless HDA_BLOCKS asks for blocks (block-Jacobi ILU drops couplings)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ilu_code = ("from hypredrive_amd import hypredrv as hd\n"
                "d = hd.Hypredrv('solver: gmres\\npreconditioner: ilu\\n'); d.set_laplacian7((48, 48, 48)); r = d.solve(); print('ilu', r['converged'])\n")
    code = ("import hypredrive_amd as h\n"
            "A = h.lap7(48, 48, 48)\n"                      # 110 592 rows > HDA_BLOCKS_MIN_ROWS
            "amg = h.Amg(A, h.AmgParams.default(coarsen_type=10, relax_down=13, relax_up=14, blocks=0))\n"
            "print('blocks', h.load().hda_amg_blocks(amg.h))\n" + ilu_code)
    out = {}
    for quiet in ("0", "1"):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("HDA_BLOCKS", None)
        if quiet == "1":
            env["HDA_QUIET"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        out[quiet] = r
    loud, silent = out["0"], out["1"]
    assert "blocks 1" not in loud.stdout and "ilu True" in loud.stdout
    assert "[hypredrv_amd] BoomerAMG setup:" in loud.stderr and "row blocks" in loud.stderr and "chosen by the setup" in loud.stderr
    assert "[hypredrv_amd]" not in silent.stderr
    assert "ILU(0):" not in loud.stderr                     # the ILU preconditioner stayed one block: nothing to announce
    env = dict(os.environ, PYTHONPATH=root, HDA_BLOCKS="8")
    r = subprocess.run([sys.executable, "-c", ilu_code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ILU(0): 8 row blocks" in r.stderr and "couplings between blocks are dropped" in r.stderr, r.stderr[-2000:]
