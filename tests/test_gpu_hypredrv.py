"""GPU tests of the reference-facing boundary (HYPREDRV_* / HYPRE_*): the CSR-ingestion tests
of the reference (tests/test_setmatrix_from_csr.c), its CLI on examples/ex1.yml, the generator
path of examples/src/C_laplacian/laplacian.c, and the row-partitioned path on several ranks."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port  # a port nobody listens on: two suites on one host do not collide
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

YAML_PCG_AMG = "solver: pcg\npreconditioner: amg\n"


@pytest.fixture(scope="module")
def hd():
    import hypredrive_amd as h
    from hypredrive_amd import hypredrv
    assert h.device_count() >= 1
    return hypredrv


def test_csr_1d_laplacian_solves(hd):
    """tests/test_setmatrix_from_csr.c:168-199: 1-D Laplacian n=16 through SetMatrixFromCSR."""
    n = 16
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n), format="csr")
    h = hd.Hypredrv(YAML_PCG_AMG)
    h.set_matrix_csr(0, n - 1, T.indptr, T.indices, T.data)
    h.set_rhs_array(0, n - 1, np.ones(n))
    h.finish_system()
    r = h.solve()
    x = h.solution()
    assert r["converged"] and np.linalg.norm(x) > 0
    assert np.allclose(T @ x, np.ones(n), atol=1e-5)


def test_csr_offset_one_by_one(hd, pins):
    """tests/test_setmatrix_from_csr.c:397-417: 3x = 6 with offset indptr and row_start = 5."""
    u = pins["unit"]["one_by_one"]
    h = hd.Hypredrv(YAML_PCG_AMG)
    h.set_matrix_csr(5, 5, [3, 4], [9, 9, 9, 5], [0.0, 0.0, 0.0, u["a"]])  # indptr[0] = 3: offset CSR
    h.set_rhs_array(5, 5, [u["b"]])
    h.finish_system()
    h.solve()
    assert abs(h.solution_norm("L2") - u["x_norm"]) < u["tol"]
    assert h.solution_norm("bad") == -1.0  # tests/test_linsys.c:4126-4155


def test_csr_direct_device_assembly_equals_the_staged_one(hd, monkeypatch):
    """HYPREDRV_LinearSystemSetMatrixFromCSR uploads the caller's arrays as they are and maps columns / sorts rows on the device
    (round 4; the staged host path took 0.65 s for 49 M entries).  Rows in any column order and an offset indptr give the operator the
    staged path builds (HDA_CSR_DIRECT=0), entry for entry; a row that names a column twice is handed to the staged path, whose rule
    (the later value wins) therefore still holds."""
    import hypredrive_amd as hh
    rng = np.random.default_rng(5)
    n = 400
    M = sp.random(n, n, density=0.03, random_state=rng, format="csr")
    M = (M + M.T + sp.diags(np.full(n, 5.0))).tocsr()
    M.sort_indices()
    ip, ix, v = M.indptr.astype(np.int64), M.indices.astype(np.int64).copy(), M.data.copy()
    for i in range(n):  # shuffle every row's entries
        q = rng.permutation(ip[i + 1] - ip[i]) + ip[i]
        ix[ip[i]:ip[i + 1]], v[ip[i]:ip[i + 1]] = ix[q], v[q]
    pad = 7  # offset CSR: indptr[0] = 7, seven unused leading entries
    ip2, ix2, v2 = ip + pad, np.r_[np.full(pad, 3), ix], np.r_[np.full(pad, 9.0), v]
    # a duplicate: row 11 names its first column again, with another value
    k0 = ip2[11]
    ipd = ip2.copy()
    ipd[12:] += 1
    ixd, vd = np.insert(ix2, ip2[12], ix2[k0]), np.insert(v2, ip2[12], 123.0)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HDA_CSR_DIRECT", mode)
        for tag, (a, b, c) in (("plain", (ip2, ix2, v2)), ("dup", (ipd, ixd, vd))):
            h = hd.Hypredrv(YAML_PCG_AMG)
            h.set_matrix_csr(0, n - 1, a, b, c)
            h.set_rhs_array(0, n - 1, np.ones(n))
            h.finish_system()
            h.create_and_setup()
            A, amg = hh._lib.borrow(h)
            got[mode, tag] = A.to_scipy().copy()
            del A, amg
            h.destroy_solver()
            h.close()
    for tag in ("plain", "dup"):
        a, b = got["1", tag], got["0", tag]
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data), tag
    assert abs(got["1", "plain"] - M).max() == 0.0
    D = got["1", "dup"]
    assert D[11, ix2[k0]] == 123.0 and D.nnz == M.nnz  # the later value won, one entry


def test_norms_known_answer(hd, pins):
    """tests/test_linsys.c:4126-4155 on the solution of I x = [1,-2,3]."""
    k = pins["unit"]["norms_of_1_m2_3"]
    h = hd.Hypredrv(YAML_PCG_AMG)
    h.set_matrix_csr(0, 2, [0, 1, 2, 3], [0, 1, 2], [1.0, 1.0, 1.0])
    h.set_rhs_array(0, 2, [1.0, -2.0, 3.0])
    h.finish_system()
    h.solve()
    assert h.solution_norm("L1") == pytest.approx(k["L1"], rel=1e-12)
    assert h.solution_norm("L2") == pytest.approx(k["L2"], rel=1e-12)
    assert h.solution_norm("Linf") == pytest.approx(k["Linf"], rel=1e-12)


def test_csr_bad_input_is_rejected(hd):
    h = hd.Hypredrv(YAML_PCG_AMG)
    with pytest.raises(hd.HypredrvError):
        h.set_matrix_csr(0, 1, [0, 2, 1], [0, 1], [1.0, 1.0])  # decreasing indptr
    with pytest.raises(hd.HypredrvError) as e:
        hd.check(hd.lib().HYPREDRV_LinearSolverApply(h.h))
    assert e.value.code & hd.ERROR_INVALID_SOLVER


def test_apply_without_setup_is_invalid_precon(hd):
    """reference src/HYPREDRV.c:3142-3151"""
    n = 8
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n), format="csr")
    h = hd.Hypredrv(YAML_PCG_AMG)
    h.set_matrix_csr(0, n - 1, T.indptr, T.indices, T.data)
    h.set_rhs_array(0, n - 1, np.ones(n))
    h.finish_system()
    hd.check(hd.lib().HYPREDRV_LinearSolverCreate(h.h))
    code = hd.lib().HYPREDRV_LinearSolverApply(h.h)
    hd.lib().HYPREDRV_ErrorCodeClear()
    assert code & hd.ERROR_INVALID_PRECON


def test_laplacian_driver_path_matches_oracle(hd, orc, pins):
    """examples/src/C_laplacian/laplacian.c defaults (10^3, presets pcg + poisson, 5 solves):
    r0 = 1.00e+01 as in examples/refOutput/laplacian.txt:34-38; iteration count equals the
    oracle run with the same (hypre-GPU default) parameters; table has one row per solve."""
    Ao, b = orc.lap7(10, 10, 10)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    h = hd.Hypredrv()
    h.presets("pcg", "poisson")
    h.set_laplacian7((10, 10, 10))
    its = [h.solve() for _ in range(5)]
    assert all(r["iters"] == ref["iters"] and r["converged"] for r in its)
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-10
    # the statistics table (Appendix C of SURVEY.md): capture stdout of StatsPrint
    code = ("from hypredrive_amd import hypredrv as hd\n"
            "h = hd.Hypredrv(); h.presets('pcg','poisson'); h.set_laplacian7((10,10,10))\n"
            "[h.solve() for _ in range(5)]; h.stats_print()\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                         env=dict(os.environ, PYTHONPATH=ROOT)).stdout
    rows = re.findall(r"^\|\s+(\d+) \|\s+([\d.]*) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", out, re.M)
    assert len(rows) == 5 and [int(r[0]) for r in rows] == [0, 1, 2, 3, 4]
    assert rows[0][1] != "" and all(r[1] == "" for r in rows[1:])  # LS build only on the first entry
    assert all(r[4] == "1.00e+01" for r in rows)
    assert all(int(r[6]) == ref["iters"] for r in rows)
    assert "times [s]" in out and "STATISTICS SUMMARY" in out


def test_cli_runs_ex1_unchanged(orc, pins):
    """hypredrive-cli examples/ex1.yml: rows/nnz/r0 as examples/refOutput/ex1.txt:17,27; iteration
    count equals the oracle with hypre-GPU defaults (the refOutput's 6 is the CPU-default
    smoother, pinned by tests/test_oracle_pins.py)."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "examples/ex1.yml"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "HYPREDRIVE Failure!!!" not in out + r.stderr
    m = re.search(r"Solving linear system #0 with (\d+) rows and (\d+) nonzeros", out)
    assert m and int(m.group(1)) == pins["ex1"]["rows"] and int(m.group(2)) == pins["ex1"]["nnz"]
    assert "rhs_filename: data/ps3d10pt7/np1/IJ.out.b" in out  # echoed configuration
    row = re.search(r"^\|\s+0 \|\s+([\d.]+) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", out, re.M)
    assert row and row.group(4) == "3.16e+01" and "times [ms]" in out
    Ao, b = orc.lap7(10, 10, 10, b_mode=1)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    assert int(row.group(6)) == ref["iters"]
    assert float(row.group(5)) < 1e-6


@pytest.mark.parametrize("cfg", ["examples/ex1-preset.yml", "examples/ex1-gmres.yml", "examples/ex2-gpu.yml", "examples/ex1-jacobi.yml",
                                 "examples/ex1-gs.yml", "examples/ex2-hl1gs.yml", "examples/ex1-cpudefaults.yml",
                                 "examples/ex1b-gmres-ilu.yml", "examples/ex8-ilu-smoother.yml", "examples/ex1a.yml", "examples/ex1b.yml", "examples/ex3-threefield.yml",
                                 "examples/ex7-scaling-threefield.yml"])
def test_cli_other_examples(cfg):
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", cfg], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    assert row
    if "ex1-gs" in cfg:
        # one forward Gauss-Seidel sweep is a nonsymmetric preconditioner: PCG need not converge, and
        # non-convergence is not an error (reference src/internal/utils.c:33-79)
        assert int(row.group(2)) <= 100  # (the reference file leaves max_iter at its default)
        return
    assert float(row.group(1)) < 1e-6
    if "cpudefaults" in cfg:
        # the reference's own output for this input and these (CPU-default) options:
        # examples/refOutput/ex1.txt:27 -- 6 iterations, 4.98e-08
        assert int(row.group(2)) == 6 and float(row.group(1)) == pytest.approx(4.98e-08, rel=0.02)
    if "ex2" in cfg:  # print_level 2: residual history in hypre's format
        assert re.search(r"^\s+1\s+\d\.\d+e[+-]\d+", r.stdout, re.M)


def test_cli_runs_ex8_multi_unchanged(orc, pins):
    """The reference's examples/ex8-multi-1.yml UNCHANGED (list-valued `include:` of ex8-amg-3 / -2 / -1 / -4.yml: four BoomerAMG
    variants -- HMIS or PMIS grids, "MM-ext+i" interpolation, Chebyshev order 2 and 4, the symmetric l1 Gauss-Seidel sweep, the
    ILU(0) complex smoother -- PCG to 1e-9 on the 10^3 system read from four part files).  Per variant the device takes the
    oracle's iteration count (oracle pinned to examples/refOutput/ex8.txt in tests/test_oracle_pins.py) and stays within
    one iteration of the reference's own numbers for the variants that output holds (ex8.txt:92-95)."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", "examples/ex8-multi-1.yml"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^\|\s+(\d+) \|\s+[\d.]* \|\s+[\d.]+ \|\s+[\d.]+ \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    assert [int(q[0]) for q in rows] == [0, 1, 2, 3], r.stdout
    assert all(q[1] == "3.16e+01" and float(q[2]) < 1e-9 for q in rows)
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    variants = [dict(coarsen_type=10, interp_type=17, strong_th=0.8, relax_down=8, relax_up=8),                                      # ex8-amg-3.yml
                dict(coarsen_type=8, interp_type=17, strong_th=0.5, relax_down=16, relax_up=16, cheby_order=4, cheby_fraction=0.1),  # ex8-amg-2.yml
                dict(coarsen_type=10, interp_type=17, strong_th=0.25, relax_down=16, relax_up=16),                                   # ex8-amg-1.yml
                dict(coarsen_type=10, interp_type=17, strong_th=0.9, relax_down=16, relax_up=16)]                                    # ex8-amg-4.yml (+ ILU)
    ref_entry = [2, None, 0, 3]  # rows of examples/refOutput/ex8.txt made with the same options (its variant 1 used HMIS, not PMIS)
    for k, v in enumerate(variants):
        amg = orc.Amg(A, orc.amg_params(False, **v))
        if k == 3:
            amg.set_ilu_smoother(1, 1)
        ro = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
        assert int(rows[k][3]) == ro["iters"], (k, rows[k], ro["iters"])
        if ref_entry[k] is not None:
            assert abs(int(rows[k][3]) - pins["ex8"]["stats"][ref_entry[k]]["iters"]) <= 1


def test_cli_runs_ex8_unchanged(orc, pins):
    """The reference's examples/ex8.yml UNCHANGED: five BoomerAMG variants as sequence items under `preconditioner: amg` -- the
    fifth with `prolongation_type: direct_sep_weights` (interpolation type 3) and two l1 symmetric Gauss-Seidel sweeps.  Every
    variant takes the oracle's iteration count; the reference's own counts (refOutput/ex8.txt:92-96, made from the first of the
    four part files: half the right-hand side of this run) stay within one."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", "examples/ex8.yml"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^\|\s+(\d+) \|\s+[\d.]* \|\s+[\d.]+ \|\s+[\d.]+ \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    assert [int(q[0]) for q in rows] == [0, 1, 2, 3, 4], r.stdout
    assert all(q[1] == "3.16e+01" and float(q[2]) < 1e-9 for q in rows)
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    variants = [dict(coarsen_type=10, interp_type=17, strong_th=0.25, relax_down=16, relax_up=16),
                dict(coarsen_type=10, interp_type=17, strong_th=0.5, relax_down=16, relax_up=16, cheby_order=4, cheby_fraction=0.1),
                dict(coarsen_type=10, interp_type=17, strong_th=0.8, relax_down=8, relax_up=8),
                dict(coarsen_type=10, interp_type=17, strong_th=0.9, relax_down=16, relax_up=16),
                dict(coarsen_type=8, interp_type=3, strong_th=0.5, relax_down=8, relax_up=8, sweeps_down=2, sweeps_up=2)]
    for k, v in enumerate(variants):
        amg = orc.Amg(A, orc.amg_params(False, **v))
        if k == 3:
            amg.set_ilu_smoother(1, 1)
        ro = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
        assert int(rows[k][3]) == ro["iters"], (k, rows[k], ro["iters"])
        assert abs(int(rows[k][3]) - pins["ex8"]["stats"][k]["iters"]) <= 1


def test_cli_runs_ex8_with_the_standard_interpolation_its_golden_output_echoes(orc, pins, tmp_path):
    """examples/refOutput/ex8.txt:74 echoes `prolongation_type: standard` for the fifth variant (the ex8.yml in the tree says
    direct_sep_weights today): the same file with that one word put back runs through the CLI, takes the oracle's count with
    interpolation type 8, and stays within one of the reference's 6 (refOutput/ex8.txt:96)."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    text = open(os.path.join(ROOT, "examples", "ex8.yml")).read()
    assert text.count("direct_sep_weights") == 1
    cfg = tmp_path / "ex8-standard.yml"
    cfg.write_text(text.replace("direct_sep_weights", "standard"))
    r = subprocess.run([cli, "-q", str(cfg)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^\|\s+(\d+) \|\s+[\d.]* \|\s+[\d.]+ \|\s+[\d.]+ \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    assert [int(q[0]) for q in rows] == [0, 1, 2, 3, 4], r.stdout
    A, b = orc.lap7(10, 10, 10, b_mode=1)
    amg = orc.Amg(A, orc.amg_params(False, coarsen_type=8, interp_type=8, strong_th=0.5, relax_down=8, relax_up=8, sweeps_down=2, sweeps_up=2))
    ro = orc.pcg(A, b, amg, orc.krylov_params(False, rtol=1e-9, max_iter=500))
    assert int(rows[4][3]) == ro["iters"] and abs(int(rows[4][3]) - pins["ex8"]["stats"][4]["iters"]) <= 1, (rows[4], ro["iters"])


@pytest.mark.parametrize("cfg", ["examples/ex3-mgr_Frelax_gmres.yml", "examples/ex3-mgr_coarse_gmres_amg.yml"])
def test_cli_runs_nested_krylov_mgr_examples(cfg):
    """The reference's examples/ex3-mgr_Frelax_gmres.yml (F-relaxation of the second reduction level = GMRES(5) preconditioned by a
    one-level BoomerAMG) and ex3-mgr_coarse_gmres_amg.yml (coarsest level = two GMRES iterations preconditioned by BoomerAMG), the
    files UNCHANGED; their compflow6k data set is not in the tree, so the three file names are overridden on the command line
    (`-a`, reference src/internal/yaml.c:2178) with the generated three-field stand-in of the same layout."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    d = "data/threefield/np1/"
    r = subprocess.run([cli, "-q", cfg, "-a", "--linear_system:rhs_filename", d + "IJ.out.b", "--linear_system:matrix_filename", d + "IJ.out.A",
                        "--linear_system:dofmap_filename", d + "dofmap.out"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    assert row and float(row.group(1)) < 1e-6 and 0 < int(row.group(2)) < 60, r.stdout
    plain = subprocess.run([cli, "-q", "examples/ex3-threefield.yml"], capture_output=True, text=True, cwd=ROOT)
    prow = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", plain.stdout, re.M)
    assert int(row.group(2)) <= int(prow.group(2)) + 2  # a stronger component does not cost iterations


@pytest.mark.parametrize("cyc", ["v(1,1)", "v(0,1)", "w(1,1)"])
def test_cli_mgr_cycle_strings(cyc):
    """`preconditioner.mgr.cycle` through the YAML surface (the strings of the reference's examples/ex7-mgr-cycle-*.yml,
    src/internal/mgr.c:614-675) on the three-field stand-in: every shape converges, the richer ones in no more iterations."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    it = {}
    for c in ("v(1,0)", cyc):
        r = subprocess.run([cli, "-q", "examples/ex3-threefield.yml", "-a", "--preconditioner:mgr:cycle", c], capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
        assert row and float(row.group(1)) < 1e-6, r.stdout
        it[c] = int(row.group(2))
    if cyc != "v(0,1)":
        assert it[cyc] <= it["v(1,0)"]
    assert it[cyc] != it["v(1,0)"] or cyc == "v(0,1)"  # the option reaches the cycle


def test_null_space_projection(hd):
    """HYPREDRV_LinearSystemSetNullSpace + the projection at the end of LinearSolverApply, as the reference's own test drives them
    (tests/test_hypredrv.c:4188-4350): modes before the matrix fail cleanly; two non-orthogonal modes are orthonormalised and the
    computed solution is orthogonal to both INPUT modes; dependent modes are refused; modes of another system size make Apply fail
    with ERROR_INVALID_VAL until they are cleared."""
    import ctypes as C
    L = hd.lib()

    def lap1d(n):
        import scipy.sparse as sp
        return sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()

    def set_ns(h, modes, ncomp):
        m = np.ascontiguousarray(modes, dtype=np.float64)
        return L.HYPREDRV_LinearSystemSetNullSpace(h.h, (m.size // ncomp) if ncomp else 0, ncomp, m.ctypes.data_as(C.POINTER(C.c_double)) if m.size else None)

    def load(h, n):
        A = lap1d(n)
        h.set_matrix_csr(0, n - 1, A.indptr, A.indices, A.data)
        h.set_rhs_array(0, n - 1, np.ones(n))
        h.finish_system()

    n = 8
    h = hd.Hypredrv("solver: pcg\npreconditioner: amg\n")
    assert set_ns(h, np.ones(1), 1) & hd.ERROR_INVALID_VAL  # no matrix yet
    L.HYPREDRV_ErrorCodeClear()
    load(h, n)
    modes = np.r_[1.0 + (np.arange(n) % 2), np.arange(1, n + 1)].astype(float)
    assert set_ns(h, modes, 2) == 0
    r = h.solve()
    assert r["converged"]
    x = h.solution()
    assert abs(x @ modes[:n]) < 1e-9 and abs(x @ modes[n:]) < 1e-9
    # the unprojected solution is not orthogonal to them: the projection did the work
    xf = np.linalg.solve(lap1d(n).toarray(), np.ones(n))
    assert abs(xf @ modes[:n]) > 1.0
    Q, _ = np.linalg.qr(modes.reshape(2, n).T)
    assert np.allclose(x, xf - Q @ (Q.T @ xf), atol=1e-6)
    assert set_ns(h, modes[:n], 1) == 0                           # replacing the modes
    assert set_ns(h, np.r_[modes[:n], 2.0 * modes[:n]], 2) & hd.ERROR_INVALID_VAL  # linearly dependent
    L.HYPREDRV_ErrorCodeClear()
    assert set_ns(h, modes, 2) == 0
    load(h, 4)                                                    # another system size: Apply must refuse, not corrupt
    hd.check(L.HYPREDRV_LinearSystemResetInitialGuess(h.h))
    h.create_and_setup()
    assert L.HYPREDRV_LinearSolverApply(h.h) & hd.ERROR_INVALID_VAL
    L.HYPREDRV_ErrorCodeClear()
    h.destroy_solver()
    assert set_ns(h, np.zeros(0), 0) == 0                         # cleared: solves succeed again
    assert h.solve()["converged"]
    h.close()


def test_cli_output_matches_the_reference_golden_output(tmp_path):
    """the reference keeps its drivers' outputs under examples/refOutput and compares them with
    scripts/compare_output.sh.  The same comparison here (tools/compare_output.py: dates / versions / executable path normalised,
    timing columns masked, the relative residual compared as a number within 2 %): `hypredrive-cli examples/ex1.yml` with the
    reference's CPU-build defaults reproduces examples/refOutput/ex1.txt -- echoed input tree, banners, table frame, 1000 rows /
    6400 nonzeros, r0 3.16e+01, 6 iterations -- line for line.  The golden file is the reference's own test fixture
    (tests/golden/refOutput/ex1.txt)."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "examples/ex1.yml"], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, HYPREDRV_AMD_DEFAULTS="cpu"))
    assert r.returncode == 0, r.stdout + r.stderr
    (tmp_path / "ex1.out").write_text(r.stdout)
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_output.py"), str(tmp_path / "ex1.out"),
                        os.path.join(ROOT, "tests", "golden", "refOutput", "ex1.txt")], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout
    # the reference's own driver, unmodified (oracle/_ref/laplacian_ref), against examples/refOutput/laplacian.txt: five solves, 5 iterations each
    drv = os.path.join(ROOT, "oracle", "_ref", "laplacian_ref")
    if os.path.exists(drv):
        d = subprocess.run([drv], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, HYPREDRV_AMD_DEFAULTS="cpu"))
        assert d.returncode == 0, d.stdout + d.stderr
        (tmp_path / "lap.out").write_text(d.stdout)
        c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_output.py"), str(tmp_path / "lap.out"),
                            os.path.join(ROOT, "tests", "golden", "refOutput", "laplacian.txt")], capture_output=True, text=True)
        assert c.returncode == 0, c.stdout
    # and it does catch a wrong iteration count
    (tmp_path / "bad.out").write_text(re.sub(r"\|      6 \|", "|      7 |", r.stdout))
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_output.py"), str(tmp_path / "bad.out"),
                        os.path.join(ROOT, "tests", "golden", "refOutput", "ex1.txt")], capture_output=True, text=True)
    assert c.returncode == 1 and "iterations differ" in c.stdout


def test_cli_statistics_level_2_aggregate_rows():
    """general.statistics: 2 appends Min. / Max. / Avg. / Std. / Total rows to the table (reference src/internal/stats.c:1262-1358):
    same column widths and formats, Total leaves the residual columns blank and sums the iterations."""
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", "examples/ex8-multi-1.yml", "-a", "--general:statistics", "2"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    its = [int(q) for q in re.findall(r"^\|\s+\d+ \|.*\|\s+(\d+) \|$", r.stdout, re.M)]
    assert len(its) == 4
    agg = dict(re.findall(r"^\|\s+(Min\.|Max\.|Avg\.|Std\.|Total) \|.*\|\s+(\S+) \|$", r.stdout, re.M))
    assert set(agg) == {"Min.", "Max.", "Avg.", "Std.", "Total"}
    assert int(agg["Min."]) == min(its) and int(agg["Max."]) == max(its) and int(agg["Total"]) == sum(its)
    assert float(agg["Avg."]) == pytest.approx(sum(its) / 4, abs=0.05)
    total = re.search(r"^\|\s+Total \|\s+[\d.]+ \|\s+[\d.]+ \|\s+[\d.]+ \|\s+\|\s+\|\s+\d+ \|$", r.stdout, re.M)
    assert total, r.stdout
    # level 1 (the default) prints none of them
    r1 = subprocess.run([cli, "-q", "examples/ex8-multi-1.yml"], capture_output=True, text=True, cwd=ROOT)
    assert "Total" not in r1.stdout


def test_cli_overrides():
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", "examples/ex1.yml", "-a", "--solver:pcg:max_iter", "3"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr  # non-convergence is not an error
    row = re.search(r"\|\s+(\d+) \|$", r.stdout, re.M)
    assert row and int(row.group(1)) == 3


@pytest.mark.parametrize("world,n,solver,rep_rows,setup", [
    (2, 16, "pcg", 100000, "partitioned"), (4, 20, "pcg", 0, "partitioned"), (4, 24, "pcg", 700, "partitioned"),
    (3, 12, "gmres", 0, "partitioned"), (4, 40, "pcg", 2000, "partitioned"), (4, 24, "pcg", 700, "replicated")])
def test_row_partitioned_solve_matches_single_rank(hd, tmp_path, world, n, solver, rep_rows, setup, P=None, reorder=None):
    """Several ranks on one GPU through the staged transport: identical hierarchy (PMIS hashes
    global ids) => same iteration count as one rank, same solution to rounding.  rep_rows
    (HDA_REPLICATE_ROWS) moves the split between partitioned levels and the replicated tail:
    0 = every level partitioned, 700 = two partitioned levels + tail, 100000 = tail from level 1.
    setup = "partitioned": distributed PMIS / ext+i / Galerkin product on the row blocks, run with
    HDA_DIST_CHECK=1, i.e. every level is compared inside the library with the replicated setup
    (same patterns; P, R of level 0 bit-identical; coarser operators to 1e-13 / 1e-10)."""
    out = str(tmp_path / "res.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_REPLICATE_ROWS=str(rep_rows), HDA_DIST_SETUP=setup,
               HDA_DIST_CHECK="1" if setup == "partitioned" else "0")
    if P:
        env["HDA_TEST_P"] = ",".join(str(v) for v in P)
    if reorder is not None:  # solve-phase renumbering forced onto small row blocks; the entry-wise self check is off then
        env["HDA_REORDER"] = str(reorder)
        env["HDA_DIST_CHECK"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "solve", out, str(n), solver]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = json.load(open(out))
    x = np.concatenate([np.load(f"{out}.x{k}.npy") for k in range(world)])
    # single rank reference on the same (block-partitioned) numbering is a permutation of the
    # lexicographic problem: compare through norms and iteration counts
    h = hd.Hypredrv(f"solver: {solver}\npreconditioner:\n  preset: poisson\n")
    h.set_laplacian7((n, n, n))
    ref = h.solve()
    assert res["converged"] and abs(res["iters"] - ref["iters"]) <= 1
    assert res["norm"] == pytest.approx(h.solution_norm("L2"), rel=1e-6)
    assert res["l1"] == pytest.approx(h.solution_norm("L1"), rel=1e-6)
    assert res["linf"] == pytest.approx(h.solution_norm("Linf"), rel=1e-6)
    assert np.linalg.norm(x) == pytest.approx(res["norm"], rel=1e-12)


def _thread_ranks(tmp_path, tag, n, P, solver="pcg", want_x="1", **envx):
    """prod(P) ranks as THREADS of one child process (hda_thread_ranks.hip): a GPU box admits six processes on its card, so this is
    how the eight ranks of BASELINE config 3's 2x2x2 layout share the one GPU."""
    out = str(tmp_path / f"{tag}.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", **envx)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "threads", out, str(n), ",".join(str(v) for v in P), solver, want_x]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    return json.load(open(out)), (np.load(out + ".x.npy") if want_x == "1" else None), r.stderr


@pytest.mark.parametrize("n,rep_rows,check,extra", [(40, 2000, "1", {}), (40, 2000, "0", {}), (32, 0, "0", {}), (48, 100000, "0", {}),
                                                    (40, 2000, "0", {"HDA_OVERLAP": "1"}), (36, 1500, "0", {"HDA_PCG_SINGLE_REDUCE": "1"}),
                                                    (40, 2000, "0", {"HDA_THREAD_TRANSPORT": "device"}), (32, 0, "0", {"HDA_THREAD_TRANSPORT": "device"}),
                                                    (36, 1500, "0", {"HDA_THREAD_TRANSPORT": "device", "HDA_PCG_SINGLE_REDUCE": "1"}),
                                                    (40, 2000, "0", {"HDA_THREAD_TRANSPORT": "device", "HDA_OVERLAP": "0"}),
                                                    (40, 2000, "0", {"HDA_THREAD_TRANSPORT": "device", "HDA_CODED": "0", "HDA_WINDOW_MIN_NNZ": "1000"}),
                                                    (40, 2000, "0", {"HDA_THREAD_TRANSPORT": "device", "HDA_CODED": "0", "HDA_WINDOW_MIN_NNZ": "1000", "HDA_WINDOW_RUNS": "0"})])
def test_eight_ranks_2x2x2_match_single_rank(hd, tmp_path, n, rep_rows, check, extra):
    """BASELINE config 3's layout -- `-P 2 2 2` (reference examples/src/C_laplacian/laplacian.c:561-582, scripts/node_scaling.sh:1275-1292) --
    with eight ranks: blocks have face, EDGE and CORNER neighbours (7 peers), which no 1xPxQ layout produces, on every partitioned
    level.  rep_rows 2000 keeps three levels partitioned at 40^3 (64 000 / ~21 000 / ~4 500 rows) above the replicated tail; 0 keeps
    every level partitioned; 100000 hands level 1 to the tail.  check = "1": HDA_DIST_CHECK, the partitioned setup compared level by
    level with the replicated one inside the library.  Against one rank: iteration count within 1, every rank the same count,
    solution norms to 1e-6 (the stopping tolerance), the gathered solution's norm to rounding."""
    # extra: HDA_OVERLAP=1 -- every product runs its owned-column part while the ghosts travel and adds the ghost-column part afterwards
    # (the RCCL default), here with 7 peers per block; HDA_PCG_SINGLE_REDUCE=1 -- the opt-in one-reduction PCG;
    # HDA_THREAD_TRANSPORT=device -- the thread transport that, like RCCL, only ENQUEUES exchanges and all-reduces on the caller's stream
    # (device-to-device copies ordered by events, hda_comm.hip DeviceThreadComm): overlapped products on by default, nothing waits on the host;
    # HDA_CODED=0 HDA_WINDOW_MIN_NNZ=1000 -- uncoded blocks small enough to be windowed: the owned-column halves of the overlapped products
    # run on the windowed kernel in its run form (ghost columns = further runs) or, with HDA_WINDOW_RUNS=0, its list form
    res, x, err = _thread_ranks(tmp_path, f"t{n}_{rep_rows}_{check}", n, (2, 2, 2), HDA_REPLICATE_ROWS=str(rep_rows), HDA_DIST_CHECK=check, **extra)
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  preset: poisson\n")
    h.set_laplacian7((n, n, n))
    ref = h.solve()
    assert res["world"] == 8 and res["converged"] and res["iters_spread"] == 0
    assert abs(res["iters"] - ref["iters"]) <= 1
    assert res["norm"] == pytest.approx(h.solution_norm("L2"), rel=1e-6)
    assert res["l1"] == pytest.approx(h.solution_norm("L1"), rel=1e-6)
    assert res["linf"] == pytest.approx(h.solution_norm("Linf"), rel=1e-6)
    assert np.linalg.norm(x) == pytest.approx(res["norm"], rel=1e-12)
    if rep_rows == 2000:
        assert res["partitioned_levels"] >= 2
        assert res["exchange"] > 0 and res["allreduce"] > 0
    if check == "1":
        assert "dist check rank 7" in err  # every rank compared its levels


@pytest.mark.parametrize("extra", [{}, {"HDA_PCG_SINGLE_REDUCE": "1"}, {"HDA_OVERLAP": "0"}])
def test_asynchronous_transport_results_do_not_depend_on_timing(tmp_path, extra):
    """The device-direct thread transport only enqueues exchanges and all-reduces (like RCCL), so the ranks' device timelines are held
    together by events alone.  HDA_THREAD_JITTER delays every rank's send-ready / operand-ready events by a rank-and-call dependent
    0-300 us, which pulls the eight timelines apart the way eight GPUs' drift: a missing wait (a ghost tail read before it landed, a send
    buffer repacked while a peer still reads it, a scalar read before its all-reduce) would change the numbers.  They must be
    bit-identical with and without it, 2x2x2 ranks, every level partitioned."""
    base = dict(HDA_REPLICATE_ROWS="0", HDA_DIST_CHECK="0", HDA_THREAD_TRANSPORT="device", **extra)
    r0, x0, _ = _thread_ranks(tmp_path, "calm", 32, (2, 2, 2), **base)
    r1, x1, _ = _thread_ranks(tmp_path, "jitter", 32, (2, 2, 2), HDA_THREAD_JITTER="300", **base)
    assert r0["converged"] and r1["converged"] and r0["iters"] == r1["iters"]
    assert r0["overlapped"] == r1["overlapped"] and (r0["overlapped"] > 0) == (extra.get("HDA_OVERLAP") != "0")
    assert np.array_equal(x0, x1)
    assert r0["final_rel"] == r1["final_rel"]


@pytest.mark.parametrize("transport", ["host", "device"])
def test_config3_full_size_512_cubed_on_eight_thread_ranks(hd, tmp_path, transport):
    """(transport "device": the enqueue-only, event-ordered thread transport that behaves like RCCL towards the library, so the products
    overlap their halo exchanges exactly as they will between GPUs -- at the full size of the configuration.)
    BASELINE config 3 AS NAMED -- the 3-D 7-pt Laplacian 512^3 (134 217 728 rows), fp64, AMG-PCG, row-partitioned 2x2x2 over eight
    ranks, a 256^3 block each (`-n 512 512 512 -P 2 2 2`, reference examples/src/C_laplacian/laplacian.c:561-582) -- on ONE MI355X:
    the eight ranks are threads of one process (hda_thread_ranks.hip) and their messages are host-staged, so this says nothing about
    speed; it says that the partitioned setup, the halo plans with face / edge / corner neighbours on every level and the solve are
    right at the full size: converged, the same iteration count on every rank and within 1 of ONE rank solving the same 512^3
    system on the same GPU (16), the same solution norms.  (What stays unmeasured is RCCL between eight GPUs.)"""
    n = 512
    from hypredrive_amd import _lib
    _lib.memory_trim()  # the eight ranks need most of the 288 GB: nothing of this process's earlier tests may sit cached on the device
    res, _, err = _thread_ranks(tmp_path, "cfg3" + transport, n, (2, 2, 2), want_x="0", **({"HDA_THREAD_TRANSPORT": "device"} if transport == "device" else {}))
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  preset: poisson\n")
    h.set_laplacian7((n, n, n))
    ref = h.solve()
    assert res["world"] == 8 and res["converged"] and res["iters_spread"] == 0
    assert ref["converged"] and abs(res["iters"] - ref["iters"]) <= 1, (res["iters"], ref["iters"])
    assert res["norm"] == pytest.approx(h.solution_norm("L2"), rel=1e-6)
    assert res["linf"] == pytest.approx(h.solution_norm("Linf"), rel=1e-6)
    assert res["partitioned_levels"] >= 3
    h.close()


@pytest.mark.parametrize("reorder", [None, "150"])
def test_prolongation_updates_ghost_copies_one_exchange_less_per_level(tmp_path, reorder):
    """C1 of SURVEY 2.4, exchanges per cycle: the partitioned setup keeps the P rows of every rank's GHOST fine points
    (AmgLevel::Pg), the prolongation updates the ghost copies of the iterate with them, and the first post-smoothing sweep runs
    without a halo exchange of its own: 3 instead of 4 exchanges per partitioned level and V-cycle.  Same arithmetic on the owned
    rows (the ghost copies hold what the owners hold, computed from the same P rows in the same order), so against
    HDA_GHOST_PROLONG=0: same iterations, solution to rounding, and exactly vcycles x partitioned levels fewer exchanges.
    2x2x2 thread ranks: corner and edge neighbours on every level; reorder = 150 forces the solve-phase renumbering onto these
    small blocks (Pg's owned coarse columns are renamed with their level)."""
    env = dict(HDA_REPLICATE_ROWS="2000", HDA_DIST_CHECK="0")
    if reorder:
        env["HDA_REORDER"] = reorder
    a, xa, _ = _thread_ranks(tmp_path, "pg1", 40, (2, 2, 2), HDA_GHOST_PROLONG="1", **env)
    b, xb, _ = _thread_ranks(tmp_path, "pg0", 40, (2, 2, 2), HDA_GHOST_PROLONG="0", **env)
    assert a["converged"] and a["iters"] == b["iters"] and a["vcycles"] == b["vcycles"]
    assert np.linalg.norm(xa - xb) <= 1e-10 * np.linalg.norm(xb)
    assert a["partitioned_levels"] == b["partitioned_levels"] >= 2
    assert b["exchange"] - a["exchange"] == a["vcycles"] * a["partitioned_levels"]


def _dist_solve(tmp_path, tag, world, n, **envx):
    out = str(tmp_path / f"{tag}.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_DIST_CHECK="0", **envx)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "solve", out, str(n), "pcg"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    return json.load(open(out)), np.concatenate([np.load(f"{out}.x{k}.npy") for k in range(world)])


def test_fused_dot_allreduce_is_bitwise_the_unfused_one(tmp_path):
    """C2 of SURVEY 2.4: <r,r> and <r,z> of a PCG iteration travel in ONE two-double all-reduce while the stopping
    test is far away.  Same finalize kernel per slot, elementwise sum: on two ranks (a + b = b + a) the iterate is
    bit-for-bit that of the path with one all-reduce per inner product (HDA_FUSE_DOTS=0)."""
    a, xa = _dist_solve(tmp_path, "fused", 2, 24, HDA_REPLICATE_ROWS="300", HDA_OVERLAP="0")
    b, xb = _dist_solve(tmp_path, "unfused", 2, 24, HDA_REPLICATE_ROWS="300", HDA_OVERLAP="0", HDA_FUSE_DOTS="0")
    assert a["iters"] == b["iters"] and a["final_rel"] == b["final_rel"]
    assert np.array_equal(xa, xb)
    # all-reduces of the solve: <b,b>; <r0,z0> + <r0,r0>; per iteration <s,p> and the fused pair (the last iterations,
    # tested before their V-cycle, keep three); one per V-cycle for the restricted residual of the replicated tail
    it, vc = a["iters"], a["vcycles"]
    assert a["comm"]["allreduce"] < b["comm"]["allreduce"]
    assert a["comm"]["allreduce"] <= 2 + 3 * it + vc and b["comm"]["allreduce"] >= 3 * it + vc
    assert a["comm"]["allreduce"] - (2 + 2 * it + vc) <= 4   # at most a few iterations near convergence are unfused


def test_single_reduction_pcg_one_allreduce_per_iteration(tmp_path):
    """C2 of SURVEY 2.4 with HDA_PCG_SINGLE_REDUCE=1 (opt-in): the three inner products of an iteration travel in ONE all-reduce of
    three doubles; with the replicated tail's restricted residual that is 2 all-reduces per iteration instead of 3 (plus the few
    iterations near convergence that test <r,r> first).  Same iteration count within 1, same solution to the stopping tolerance."""
    a, xa = _dist_solve(tmp_path, "sr1", 4, 24, HDA_REPLICATE_ROWS="700", HDA_PCG_SINGLE_REDUCE="1")
    b, xb = _dist_solve(tmp_path, "sr0", 4, 24, HDA_REPLICATE_ROWS="700")
    assert a["converged"] and abs(a["iters"] - b["iters"]) <= 1
    assert np.linalg.norm(xa - xb) <= 1e-5 * np.linalg.norm(xb)
    it, vc = a["iters"], a["vcycles"]
    # the untimed r0 and final-residual norms of Apply (2); <b,b> and the first triple (2); one triple per iteration; one per V-cycle
    # (the tail's restricted residual); up to three <r,r>-first tests near convergence
    assert a["comm"]["allreduce"] <= 4 + it + vc + 3
    assert a["comm"]["allreduce"] < b["comm"]["allreduce"] - (it - 3)


def test_overlapped_halo_exchange_matches_the_serial_one(tmp_path):
    """C1 of SURVEY 2.4: every product of the cycle and the PCG product start on the rows' owned columns while the ghost
    values travel on the communication stream; the ghost-column part is added afterwards (k_offd_fix).  Against the
    path that finishes every exchange first (HDA_OVERLAP=0): same iteration count, solution to rounding (the row sums
    are split in two), and every exchange of the solve was an overlapped one."""
    a, xa = _dist_solve(tmp_path, "ovl", 4, 24, HDA_REPLICATE_ROWS="700", HDA_OVERLAP="1")
    b, xb = _dist_solve(tmp_path, "ser", 4, 24, HDA_REPLICATE_ROWS="700", HDA_OVERLAP="0")
    assert a["iters"] == b["iters"]
    assert np.linalg.norm(xa - xb) <= 1e-12 * np.linalg.norm(xb)
    assert a["comm"]["exchange"] == b["comm"]["exchange"] > 0
    assert b["comm"]["overlapped"] == 0 and a["comm"]["overlapped"] >= a["comm"]["exchange"] - 2  # (untimed r0 / final residual products)


@pytest.mark.parametrize("world,n,rep_rows,setup", [(4, 24, 700, "partitioned"), (2, 20, 300, "partitioned"),
                                                        (4, 24, 700, "replicated"), (2, 32, 0, "partitioned")])
def test_row_partitioned_with_renumbered_blocks(hd, tmp_path, world, n, rep_rows, setup):
    """The solve-phase renumbering (hda_reorder.hip) on row blocks: only owned unknowns move, the halo
    send lists follow them, ghost slots and the level handed to the replicated tail keep their order.
    Forced down to 150 rows; same iterations and solution as one rank."""
    test_row_partitioned_solve_matches_single_rank(hd, tmp_path, world, n, "pcg", rep_rows, setup, reorder=150)


@pytest.mark.parametrize("world,n,P", [(2, 16, (2, 1, 1)), (4, 20, (2, 2, 1)), (4, 18, (2, 1, 2))])
def test_row_partitioned_x_and_y_splits(hd, tmp_path, world, n, P):
    """Rank grids that cut the x direction (the fastest index inside a block of the generator's
    numbering, laplacian.c:504-520), as the 2x2x2 grid of an 8-GPU run does."""
    test_row_partitioned_solve_matches_single_rank(hd, tmp_path, world, n, "pcg", 300, "partitioned", P=P)


@pytest.mark.parametrize("world,n,seed,rep_rows", [(3, 4000, 11, 0), (4, 6000, 12, 500)])
def test_row_partitioned_irregular_csr(hd, tmp_path, world, n, seed, rep_rows):
    """Irregular matrix (random couplings, a few long-range) handed over in row blocks with
    HYPREDRV_LinearSystemSetMatrixFromCSR on every rank (reference tests/test_setmatrix_from_csr_mpi.c):
    the partitioned AMG setup -- ghost layers of very different sizes, neighbours that are not a
    Cartesian stencil -- is compared level by level with the replicated one inside the library
    (HDA_DIST_CHECK=1), and the solve with the single-rank solve of the same matrix."""
    out = str(tmp_path / "res.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_REPLICATE_ROWS=str(rep_rows), HDA_DIST_CHECK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "csr", out, str(n), str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "dist check rank 0 level 0 P: identical pattern, max rel diff 0.00e+00" in r.stderr
    res = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import random_mmatrix
    A = random_mmatrix(seed, n)
    h = hd.Hypredrv("solver: pcg\npreconditioner: amg\n")
    h.set_matrix_csr(0, n - 1, A.indptr, A.indices, A.data)
    h.set_rhs_array(0, n - 1, np.ones(n))
    h.finish_system()
    ref = h.solve()
    assert res["converged"] and res["iters"] == ref["iters"]   # identical hierarchy: PMIS hashes global ids
    assert res["norm"] == pytest.approx(h.solution_norm("L2"), rel=1e-8)
    h.close()


def test_reference_laplacian_driver_unmodified(orc):
    """The reference's example driver (examples/src/C_laplacian/laplacian.c), compiled UNMODIFIED
    against include/HYPREDRV.h + libhypredrv_amd.so by __graft_entry__.build(): same table as
    examples/refOutput/laplacian.txt:34-38 (5 entries, r0 = 1.00e+01, LS build only on entry 0);
    iteration count = oracle with the hypre-GPU defaults this library ships."""
    exe = os.path.join(ROOT, "oracle", "_ref", "laplacian_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/laplacian_ref not built (needs /root/reference + MPICH at build time)")
    r = subprocess.run([exe, "-v", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^\|\s+(\d+) \|\s+([\d.]*) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    assert len(rows) == 5 and all(x[4] == "1.00e+01" for x in rows)
    Ao, b = orc.lap7(10, 10, 10)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    assert all(int(x[6]) == ref["iters"] and float(x[5]) < 1e-6 for x in rows)
    # 27-point stencil through the same unmodified driver (row-at-a-time IJ assembly)
    r = subprocess.run([exe, "-v", "1", "-s", "27", "-n", "12", "12", "12", "-ns", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    assert row and float(row.group(1)) < 1e-6


def test_reference_laplacian_driver_unmodified_at_benchmark_size():
    """The same unmodified binary on BASELINE config 2 (`-n 256 256 256`, reference scripts/node_scaling.sh's one-GPU point): six
    solves, r0 = 256 = sqrt(65 536 ones), 13 iterations each -- the oracle's count at 256^3 (test_parity_at_128_and_256_cubed,
    bench.py's cpu_baseline.iters_match) -- and the reference's own 'solve' column under 0.05 s."""
    exe = os.path.join(ROOT, "oracle", "_ref", "laplacian_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/laplacian_ref not built (needs /root/reference + MPICH at build time)")
    r = subprocess.run([exe, "-v", "1", "-n", "256", "256", "256", "-ns", "6"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = re.findall(r"^\|\s+(\d+) \|\s+([\d.]*) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    assert len(rows) == 6 and all(x[4] == "2.56e+02" and int(x[6]) == 13 and float(x[5]) < 1e-6 for x in rows), r.stdout[-2000:]
    assert all(float(x[3]) < 0.05 for x in rows[1:]) and all(float(x[2]) < 1.0 for x in rows[1:])


def _run_convdif(cfg):
    exe = os.path.join(ROOT, "oracle", "_ref", "convdif_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/convdif_ref not built (needs /root/reference + MPICH at build time)")
    r = subprocess.run([exe, "-i", cfg, "-v", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    steps = re.findall(r"^Time step:\s+(\d+) \|.*\| Lin:\s+(\d+) \| min\(c\)=\s*\S+ max\(c\)=\s*(\S+) mass=(\S+)", r.stdout, re.M)
    rows = re.findall(r"^\|\s+(\d+\.\d+) \|\s+[\d.]* \|\s+[\d.]+ \|\s+[\d.]+ \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    return r.stdout, steps, rows


def test_reference_convdif_driver_unmodified(pins):
    """The reference's second self-contained driver (examples/src/C_convdif/convdif.c: implicit
    convection-diffusion time stepping, nonsymmetric operator, GMRES(30) + BoomerAMG, library mode,
    level annotations), compiled UNMODIFIED against include/ + libhypredrv_amd.so, run with the
    reference's CPU-default AMG options.  Against examples/refOutput/convdif.txt: the initial
    residual norms of all ten systems and the printed physics (max c, total mass) agree to every
    printed digit, the table carries the same "timestep.system" paths and the aggregate summary."""
    out, steps, rows = _run_convdif("examples/convdif-cpudefaults.yml")
    ref = pins["convdif"]
    assert len(steps) == len(ref["steps"]) == 10
    for got, want in zip(steps, ref["steps"]):
        assert int(got[0]) == want["step"]
        assert float(got[2]) == pytest.approx(want["cmax"], rel=2e-3)   # printed with 3 digits
        assert float(got[3]) == pytest.approx(want["mass"], rel=2e-6)   # printed with 7 digits
        assert 1 <= int(got[1]) <= want["lin"]
    assert [x[0] for x in rows] == [p["path"] for p in ref["paths"]]        # "1.1" ... "10.10"
    assert [x[1] for x in rows] == [f"{p['r0']:.2e}" for p in ref["paths"]]  # same systems: same ||b - A x0||
    assert all(float(x[2]) < 1e-6 for x in rows)
    assert "Aggregate Summary:" in out and "Total number of Non-linear iterations: 10" in out


def test_reference_convdif_iteration_counts(pins):
    """Iteration counts of refOutput/convdif.txt (4 5 5 5 6 7 7 8 8 9, finals 1.8e-10..8.9e-9): the file
    behaves like relative_tol 1e-8 (the driver's preset gives 1e-6 today; the checked-in outputs are
    older than the code).  With 1e-8 this build takes 4 4 5 5 6 6 6 6 7 7 (round 3, HMIS = the Ruge first pass
    alone: 4 5 5 5 5 6 6 6 7 7; since round 4 HMIS ends with hypre's PMIS pass over what the first pass left
    undecided, which moves system 2 from 5 iterations at 1.5e-10 to 4 at 6.4e-09 -- per-iteration rates 0.011 /
    0.009 against the reference's 0.011 -- and system 1's rate from 0.0093 to 0.0069 against the reference's
    0.0051).  The bar written here: every count within [ref - 2, ref], every final residual below 1e-8."""
    _, steps, rows = _run_convdif("examples/convdif-cpudefaults-tol8.yml")
    ref = pins["convdif"]
    got = [int(x[3]) for x in rows]
    want = [p["iters"] for p in ref["paths"]]
    assert got[0] == want[0]
    assert all(w - 2 <= g <= w for g, w in zip(got, want)), (got, want)
    assert all(float(x[2]) < 1e-8 for x in rows)
    for g, w in zip(steps, ref["steps"]):
        assert float(g[3]) == pytest.approx(w["mass"], rel=2e-6)


def test_level_annotations_and_level_stats(hd):
    """HYPREDRV_AnnotateLevelBegin/End + StatsLevelGetCount/GetEntry (reference src/internal/stats.c:953-1122,
    1615-1690; behaviours of tests/test_stats.c:28-152,275-335): a level entry aggregates the solves
    between its begin and end; child ids restart under every parent; misuse is ERROR_INVALID_VAL;
    an End without a Begin is a no-op."""
    L = hd.lib()
    h = hd.Hypredrv("solver: pcg\npreconditioner: amg\n")
    h.set_laplacian7((10, 10, 10))
    its = []
    for t in range(2):
        assert L.HYPREDRV_AnnotateLevelBegin(h.h, 0, b"timestep", t) == 0
        for n in range(2):
            assert L.HYPREDRV_AnnotateLevelBegin(h.h, 1, b"newton", n) == 0
            its.append(h.solve()["iters"])
            assert L.HYPREDRV_AnnotateLevelEnd(h.h, 1, b"newton", n) == 0
        assert L.HYPREDRV_AnnotateLevelEnd(h.h, 0, b"timestep", t) == 0
    cnt = C.c_int()
    assert L.HYPREDRV_StatsLevelGetCount(h.h, 0, C.byref(cnt)) == 0 and cnt.value == 2
    assert L.HYPREDRV_StatsLevelGetCount(h.h, 1, C.byref(cnt)) == 0 and cnt.value == 4
    eid, ns, li = C.c_int(), C.c_int(), C.c_int()
    ts, tv = C.c_double(), C.c_double()
    assert L.HYPREDRV_StatsLevelGetEntry(h.h, 0, 1, C.byref(eid), C.byref(ns), C.byref(li), C.byref(ts), C.byref(tv)) == 0
    assert (eid.value, ns.value, li.value) == (2, 2, its[2] + its[3]) and ts.value > 0 and tv.value > 0
    assert L.HYPREDRV_StatsLevelGetEntry(h.h, 1, 3, C.byref(eid), C.byref(ns), C.byref(li), None, None) == 0
    assert (eid.value, ns.value, li.value) == (2, 1, its[3])     # second newton step of its timestep
    assert L.HYPREDRV_StatsLevelGetEntry(h.h, 0, 0, None, None, None, None, None) == 0
    assert L.HYPREDRV_StatsLevelGetEntry(h.h, 0, -1, C.byref(eid), None, None, None, None) != 0
    L.HYPREDRV_ErrorCodeClear()
    assert L.HYPREDRV_StatsLevelGetCount(h.h, 12, C.byref(cnt)) & hd.ERROR_INVALID_VAL
    L.HYPREDRV_ErrorCodeClear()
    # misuse
    assert L.HYPREDRV_AnnotateLevelEnd(h.h, 2, b"never_begun", 0) == 0
    assert L.HYPREDRV_AnnotateLevelBegin(h.h, 0, b"timestep", 7) == 0
    assert L.HYPREDRV_AnnotateLevelBegin(h.h, 0, b"timestep", 8) & hd.ERROR_INVALID_VAL
    L.HYPREDRV_ErrorCodeClear()
    assert L.HYPREDRV_AnnotateLevelEnd(h.h, 0, b"other", 7) & hd.ERROR_INVALID_VAL
    L.HYPREDRV_ErrorCodeClear()
    assert L.HYPREDRV_AnnotateLevelEnd(h.h, 0, b"timestep", 7) == 0
    assert L.HYPREDRV_AnnotateLevelBegin(h.h, 10, b"deep", 0) & hd.ERROR_INVALID_VAL
    L.HYPREDRV_ErrorCodeClear()
    h.close()


def test_reference_elasticity_driver_unmodified(pins):
    """Third self-contained driver of the reference (examples/src/C_elasticity/elasticity.c: Q1
    hexahedra, 3 unknowns per node, interleaved dofmap, presets pcg + elasticity_3D = systems AMG
    with num_functions 3 and strong_th 0.8), UNMODIFIED.  The driver leaves no way to pick
    relaxation or coarsening, so the CPU-build defaults of the reference are selected with
    HYPREDRV_AMD_DEFAULTS=cpu; examples/refOutput/elasticity.txt:37-41 then reads 21 iterations,
    r0 1.79e+01, 2.66e-07 -- this build: 21 iterations, 1.79e+01, 2.97e-07."""
    exe = os.path.join(ROOT, "oracle", "_ref", "elasticity_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/elasticity_ref not built (needs /root/reference + MPICH at build time)")
    r = subprocess.run([exe, "-v", "1"], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, HYPREDRV_AMD_DEFAULTS="cpu"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = re.findall(r"^\|\s+(\d+) \|\s+([\d.]*) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    ref = pins["elasticity"]["stats"]
    assert len(rows) == len(ref) == 5
    for got, want in zip(rows, ref):
        assert got[4] == f"{want['r0']:.2e}"
        assert int(got[6]) == want["iters"] == 21
        assert float(got[5]) == pytest.approx(want["rel"], rel=0.2) and float(got[5]) < 1e-6
    # the library's own (GPU) defaults on the same system: PMIS + l1-Jacobi need about twice the iterations
    r = subprocess.run([exe, "-v", "1", "-ns", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0
    row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    assert row and float(row.group(1)) < 1e-6 and 21 < int(row.group(2)) < 80


def test_reference_heatflow_driver_unmodified():
    """Fourth unmodified driver (examples/src/C_heatflow/heatflow.c: transient nonlinear heat conduction,
    Newton iterations inside time steps, GMRES + BoomerAMG, HYPREDRV_StateVector* for the time levels,
    two annotation levels).  No checked-in output exists for it; the checks are the driver's own:
    every Newton solve converges, the error against its manufactured solution stays at
    discretisation level, energy decays, and the table carries "timestep.newton.system" paths."""
    exe = os.path.join(ROOT, "oracle", "_ref", "heatflow_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/heatflow_ref not built (needs /root/reference + MPICH at build time)")
    r = subprocess.run([exe, "-v", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    steps = re.findall(r"^Time step:\s+(\d+) \|.*\| NL:\s+(\d+) \| Lin:\s+(\d+) \|.*L2\(Err\)=(\S+) \| E=(\S+)", r.stdout, re.M)
    assert len(steps) >= 5
    assert all(int(s[1]) >= 1 and 1 <= int(s[2]) <= 40 and float(s[3]) < 2e-2 for s in steps)
    E = [float(s[4]) for s in steps]
    assert all(b < a for a, b in zip(E, E[1:]))               # T = 0 on one face, insulated elsewhere: energy decays
    rows = re.findall(r"^\|\s+(\d+\.\d+\.\d+) \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    assert len(rows) >= len(steps) and rows[0][0] == "1.1.1" and all(float(x[1]) < 1e-6 for x in rows)


def test_reference_laplacian_driver_cpu_defaults(pins):
    """examples/refOutput/laplacian.txt:34-38 (5 iterations, 6.12e-07) through the unmodified driver
    with the reference's CPU-build defaults."""
    exe = os.path.join(ROOT, "oracle", "_ref", "laplacian_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/laplacian_ref not built")
    r = subprocess.run([exe, "-v", "1"], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, HYPREDRV_AMD_DEFAULTS="cpu"))
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^\|\s+(\d+) \|\s+([\d.]*) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    ref = pins["laplacian"]["stats"]
    assert len(rows) == len(ref) == 5
    for got, want in zip(rows, ref):
        assert got[4] == f"{want['r0']:.2e}" and int(got[6]) == want["iters"]
        assert float(got[5]) == pytest.approx(want["rel"], rel=0.02)


def test_rccl_transport_single_rank_selftest():
    """RCCL refuses two ranks on one GPU, so the builder cannot run it multi-rank; at least
    exercise the whole RCCL code path (dlopen, ncclGetUniqueId, ncclCommInitRank, all-reduce,
    grouped send/recv plumbing) with a 1-rank communicator, in a fresh process."""
    code = (
        "import ctypes as C, os\n"
        "os.environ['HDA_FORCE_RCCL'] = '1'\n"
        "from hypredrive_amd import hypredrv as hd\n"
        "L = hd.lib(); uid = (C.c_ubyte * 128)()\n"
        "hd.check(L.HYPREDRV_AMD_CommGetUniqueId(uid))\n"
        "hd.check(L.HYPREDRV_AMD_CommInit(0, 1, 0, uid))\n"
        "assert L.hda_comm_selftest() == 0, L.hda_last_error()\n"
        "h = hd.Hypredrv('solver: pcg\\npreconditioner: amg\\n'); h.set_laplacian7((12, 12, 12)); r = h.solve()\n"
        "assert r['converged']\n"
        "hd.check(L.HYPREDRV_AMD_CommFinalize()); print('rccl-ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT),
                       timeout=600)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ----------------------------------------------------------- on-disk containers

def _lap_coo(n=10):
    import scipy.sparse as sp
    I = sp.identity(n)
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    A = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr()
    A.sort_indices()
    return A


def _write_binary_parts(prefix_A, prefix_b, A, nparts, idx_bytes=8, val_bytes=8):
    """The reference's multipart containers (src/internal/matrix.c:142-300: 11 x u64 header, rows, cols,
    vals; src/internal/vector.c:92-380: 8 x u64 header, vals)."""
    N = A.shape[0]
    it = np.uint64 if idx_bytes == 8 else np.uint32
    vt = np.float64 if val_bytes == 8 else np.float32
    for p in range(nparts):
        lo, hi = p * N // nparts, (p + 1) * N // nparts
        blk = A[lo:hi].tocoo()
        hd = np.zeros(11, dtype=np.uint64)
        hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], hd[8], hd[9], hd[10] = idx_bytes, val_bytes, N, N, hi - lo, blk.nnz, lo, hi - 1, 0, N - 1
        with open(f"{prefix_A}.{p:05d}.bin", "wb") as f:
            hd.tofile(f)
            (blk.row + lo).astype(it).tofile(f)
            blk.col.astype(it).tofile(f)
            blk.data.astype(vt).tofile(f)
        hv = np.zeros(8, dtype=np.uint64)
        hv[1], hv[5] = val_bytes, hi - lo
        with open(f"{prefix_b}.{p:05d}.bin", "wb") as f:
            hv.tofile(f)
            np.ones(hi - lo, dtype=vt).tofile(f)


def _solve_files(hd, tmp_path, extra=""):
    h = hd.Hypredrv(f"linear_system:\n  matrix_filename: {tmp_path}/A\n  rhs_filename: {tmp_path}/b\n{extra}solver: pcg\npreconditioner: amg\n")
    hd.check(hd.lib().HYPREDRV_LinearSystemBuild(h.h))
    r = h.solve()
    nrm = h.solution_norm("L2")
    h.close()
    return r, nrm


@pytest.mark.parametrize("nparts,ib,vb", [(1, 8, 8), (3, 4, 8), (2, 8, 4)])
def test_multipart_binary_files_match_ascii(hd, orc, tmp_path, nparts, ib, vb):
    """linear_system files in hypredrive's multipart binary container (one rank reads all parts,
    32- or 64-bit indices, float or double coefficients) solve like the ASCII ps3d10pt7 data."""
    A = _lap_coo(10)
    _write_binary_parts(str(tmp_path / "A"), str(tmp_path / "b"), A, nparts, ib, vb)
    r, nrm = _solve_files(hd, tmp_path)
    Ao, b = orc.lap7(10, 10, 10, b_mode=1)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    assert r["converged"] and r["iters"] == ref["iters"]
    assert nrm == pytest.approx(np.linalg.norm(ref["x"]), rel=1e-6)


def test_multipart_binary_bad_files_are_errors(hd, tmp_path):
    """Truncated header / wrong index width -> ERROR_FILE_UNEXPECTED_ENTRY, as tests/test_vector.c:189-212
    and the validation in src/internal/matrix.c:36-131 demand; the handle stays usable."""
    (tmp_path / "A.00000.bin").write_bytes(np.zeros(4, dtype=np.uint64).tobytes())
    h = hd.Hypredrv(f"linear_system:\n  matrix_filename: {tmp_path}/A\n  rhs_filename: {tmp_path}/b\nsolver: pcg\npreconditioner: amg\n")
    code = hd.lib().HYPREDRV_LinearSystemReadMatrix(h.h)
    assert code & hd.ERROR_FILE_UNEXPECTED_ENTRY
    hdr = np.zeros(11, dtype=np.uint64)
    hdr[1], hdr[2], hdr[3], hdr[4], hdr[8] = 2, 8, 4, 4, 3
    (tmp_path / "A.00000.bin").write_bytes(hdr.tobytes())
    code = hd.lib().HYPREDRV_LinearSystemReadMatrix(h.h)
    assert code & hd.ERROR_FILE_UNEXPECTED_ENTRY
    h.close()


@pytest.mark.parametrize("symmetric", [False, True])
def test_matrix_market_file(hd, orc, tmp_path, symmetric):
    """linear_system.type mtx (reference src/internal/linsys.c:986 HYPRE_IJMatrixReadMM)."""
    import scipy.sparse as sp
    A = _lap_coo(10)
    M = sp.tril(A).tocoo() if symmetric else A.tocoo()
    with open(tmp_path / "A", "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real {'symmetric' if symmetric else 'general'}\n% 7-pt Laplacian\n")
        f.write(f"{A.shape[0]} {A.shape[1]} {M.nnz}\n")
        for i, j, v in zip(M.row, M.col, M.data):
            f.write(f"{i + 1} {j + 1} {v:.17g}\n")
    h = hd.Hypredrv(f"linear_system:\n  type: mtx\n  matrix_filename: {tmp_path}/A\n  rhs_mode: ones\nsolver: pcg\npreconditioner: amg\n")
    hd.check(hd.lib().HYPREDRV_LinearSystemBuild(h.h))
    r = h.solve()
    Ao, b = orc.lap7(10, 10, 10, b_mode=1)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    assert r["converged"] and r["iters"] == ref["iters"]
    h.close()


# ------------------------------------------------- ILU(0): preconditioner and AMG complex smoother through the YAML surface

ILU_YAML = "solver:\n  gmres:\n    relative_tol: 1.0e-8\npreconditioner:\n  ilu:\n    type: bj-iluk\n    fill_level: 0\n    max_iter: {mi}\n    tri_solve: {ts}\n"
AMG_ILU_YAML = ("solver: pcg\npreconditioner:\n  amg:\n    smoother:\n      type: ilu\n      num_levels: {nl}\n      num_sweeps: {ns}\n"
                "      ilu:\n        type: bj-iluk\n        tri_solve: {ts}\n")


@pytest.mark.parametrize("mi,ts", [(1, 1), (2, 0)])
def test_yaml_ilu_preconditioner_matches_oracle(hd, orc, mi, ts):
    """'preconditioner: ilu' (reference examples/ex1b.yml selects it the same way; keys of src/internal/ilu.c:15-28)
    under GMRES: iteration count and solution of the oracle's block-Jacobi ILU(0)."""
    Ao, b = orc.lap7(12, 12, 12)
    ref = orc.gmres(Ao, b, orc.IluPrecond(Ao, max_iter=mi, tri_solve=ts), orc.krylov_params(True, rtol=1e-8))
    h = hd.Hypredrv(ILU_YAML.format(mi=mi, ts=ts))
    h.set_laplacian7((12, 12, 12))
    r = h.solve()
    assert r["converged"] and r["iters"] == ref["iters"]
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-8
    h.close()


@pytest.mark.parametrize("nl,ns,ts", [(1, 1, 1), (2, 2, 0)])
def test_yaml_amg_ilu_smoother_matches_oracle(hd, orc, nl, ns, ts):
    """amg.smoother.type ilu / num_levels / num_sweeps (reference examples/ex8.yml variant 4, amg.c:899-921)."""
    Ao, b = orc.lap7(14, 14, 14)
    ao = orc.Amg(Ao, orc.amg_params(True))
    ao.set_ilu_smoother(num_levels=nl, num_sweeps=ns, tri_solve=ts)
    ref = orc.pcg(Ao, b, ao)
    h = hd.Hypredrv(AMG_ILU_YAML.format(nl=nl, ns=ns, ts=ts))
    h.set_laplacian7((14, 14, 14))
    r = h.solve()
    assert r["converged"] and r["iters"] == ref["iters"]
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-9
    h.close()


SPE10_YAML = ("solver:\n  gmres:\n    relative_tol: 1.0e-6\npreconditioner:\n  amg:\n    smoother:\n      type: ilu\n      num_levels: 1\n"
              "      ilu:\n        type: bj-iluk\n        tri_solve: 0\n")


@pytest.mark.parametrize("via", ["csr", "mtx"])
def test_spe10_like_gmres_amg_ilu_matches_oracle(hd, orc, tmp_path, via):
    """BASELINE config 5 in the small: a heterogeneous, anisotropic (k_v / k_h = 1e-3, three decades of log-normal
    permeability) 7-point reservoir operator -- hypredrive_amd/synthetic.py stands in for SPE10, which is unreachable offline --
    solved by GMRES(30) + BoomerAMG with the ILU(0) complex smoother on level 0 (Jacobi-iterative triangular solves).  Nothing in
    this hierarchy is constant-coefficient, so every operator runs through the plain-CSR kernels.  Handed over as CSR arrays
    (HYPREDRV_LinearSystemSetMatrixFromCSR) and as a Matrix Market file (linear_system.type mtx, reference linsys.c:986): the
    oracle's iteration count and solution."""
    import scipy.sparse as sp
    from hypredrive_amd.synthetic import spe10_like
    n = 40
    N = n ** 3
    ip, ix, v, b = spe10_like(n)
    S = sp.csr_matrix((v, ix, ip), shape=(N, N))
    Ao = orc.Csr.from_scipy(S)
    ao = orc.Amg(Ao, orc.amg_params(True))
    ao.set_ilu_smoother(num_levels=1, num_sweeps=1, tri_solve=0)
    ref = orc.gmres(Ao, b, ao, orc.krylov_params(True))
    plain = orc.gmres(Ao, b, orc.Amg(Ao, orc.amg_params(True)), orc.krylov_params(True))
    assert ref["converged"] and ref["iters"] < plain["iters"]  # the smoother earns its keep on this operator
    if via == "csr":
        h = hd.Hypredrv(SPE10_YAML)
        h.set_matrix_csr(0, N - 1, ip, ix, v)
        h.set_rhs_array(0, N - 1, b)
        h.finish_system()
    else:
        M = S.tocoo()
        with open(tmp_path / "A", "w") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n")
            f.write(f"{N} {N} {M.nnz}\n")
            for i, j, q in zip(M.row, M.col, M.data):
                f.write(f"{i + 1} {j + 1} {q:.17g}\n")
        h = hd.Hypredrv(f"linear_system:\n  type: mtx\n  matrix_filename: {tmp_path}/A\n  rhs_mode: ones\n" + SPE10_YAML)
        hd.check(hd.lib().HYPREDRV_LinearSystemBuild(h.h))
        ref = orc.gmres(Ao, np.ones(N), ao, orc.krylov_params(True))
    r = h.solve()
    assert r["converged"] and r["iters"] == ref["iters"], (r["iters"], ref["iters"])
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-7
    h.close()


def test_yaml_aggressive_coarsening_matches_oracle(hd, orc, tmp_path):
    """`preconditioner: amg: aggressive: {num_levels: 1}` (reference AMGagg_args, src/internal/amg.c:160-173 -> HYPRE_BoomerAMGSetAgg*
    at :938-944) through the HYPREDRV_* API: the oracle's iteration count and solution; `num_paths: 2` likewise; what is not built
    (a two-stage interpolation type) is refused by name; `max_nnz_row` / `trunc_factor` truncate the aggressive levels' interpolation as in
    the oracle; on row partitions the aggressive levels come from the replicated setup."""
    n = 20
    Ao, b = orc.lap7(n, n, n)
    for paths in (1, 2):
        ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True, agg_num_levels=1, agg_num_paths=paths)))
        h = hd.Hypredrv(f"solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: 1\n      num_paths: {paths}\n")
        h.set_laplacian7((n, n, n))
        r = h.solve()
        assert r["converged"] and r["iters"] == ref["iters"], (r["iters"], ref["iters"])
        assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-10
        h.close()
    plain = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    assert ref["iters"] > plain["iters"]  # the cheaper hierarchy costs iterations
    # truncation of the aggressive levels' interpolation (aggressive.max_nnz_row / trunc_factor -> SetAggPMaxElmts / SetAggTruncFactor)
    for extra, kw in (("max_nnz_row: 2", dict(agg_pmax=2)), ("trunc_factor: 0.3", dict(agg_trunc_factor=0.3))):
        ref2 = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True, agg_num_levels=1, **kw)))
        h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: 1\n      " + extra + "\n")
        h.set_laplacian7((n, n, n))
        r = h.solve()
        assert r["converged"] and r["iters"] == ref2["iters"], (extra, r["iters"], ref2["iters"])
        h.close()
    # the two-stage interpolation types are not built: refused by name
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: 1\n      prolongation_type: 2_stage_extended+i\n")
    h.set_laplacian7((8, 8, 8))
    with pytest.raises(hd.HypredrvError, match="multipass"):
        h.solve()
    hd.lib().HYPREDRV_ErrorCodeClear()
    h.close()
    # row partitions (three ranks, irregular matrix): the aggressive levels are built on the gathered operator (the replicated setup,
    # like HMIS and systems AMG) and cut into row blocks -- the single-rank hierarchy, so the single-rank iteration count
    nn, seed = 4000, 3
    yaml = "solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: 1\n"
    out = str(tmp_path / "agg3.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_REPLICATE_ROWS="300", HDA_TEST_YAML=yaml)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "csr", out, str(nn), str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import random_mmatrix
    M = random_mmatrix(seed, nn)
    h = hd.Hypredrv(yaml)
    h.set_matrix_csr(0, nn - 1, M.indptr, M.indices, M.data)
    h.set_rhs_array(0, nn - 1, np.ones(nn))
    h.finish_system()
    one = h.solve()
    assert res["converged"] and res["iters"] == one["iters"]
    assert res["norm"] == pytest.approx(h.solution_norm("L2"), rel=1e-8)
    h.close()


def test_yaml_unimplemented_ilu_variants_fail_loudly(hd):
    for extra in ("type: bj-ilut", "fill_level: 1", "reordering: 1"):
        h = hd.Hypredrv("solver: gmres\npreconditioner:\n  ilu:\n    " + extra + "\n")
        h.set_laplacian7((6, 6, 6))
        with pytest.raises(hd.HypredrvError, match="implemented"):
            h.solve()
        hd.lib().HYPREDRV_ErrorCodeClear()
        h.close()
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    smoother:\n      type: fsai\n      num_levels: 1\n")
    h.set_laplacian7((6, 6, 6))
    with pytest.raises(hd.HypredrvError, match="ILU"):
        h.solve()
    hd.lib().HYPREDRV_ErrorCodeClear()
    h.close()


@pytest.mark.parametrize("world,kind", [(3, "ilu"), (4, "amg-ilu")])
def test_row_partitioned_ilu_is_block_jacobi(hd, orc, tmp_path, world, kind):
    """Row blocks on several ranks: the ILU is the block-Jacobi one (every rank factorises its diagonal
    block, ghost couplings enter only through the residual) -- same iteration count as the oracle's ILU
    on the same partition, as preconditioner and as level-0 smoother of the partitioned AMG hierarchy."""
    n, seed = 3000, 21
    out = str(tmp_path / "res.json")
    yaml = ILU_YAML.format(mi=1, ts=1) if kind == "ilu" else AMG_ILU_YAML.format(nl=1, ns=1, ts=1)
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_REPLICATE_ROWS="400", HDA_TEST_YAML=yaml)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "csr", out, str(n), str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import random_mmatrix
    A = random_mmatrix(seed, n)
    Ao = orc.Csr.from_scipy(A)
    part = [k * n // world for k in range(world + 1)]
    if kind == "ilu":
        ref = orc.gmres(Ao, np.ones(n), orc.IluPrecond(Ao, part=part), orc.krylov_params(True, rtol=1e-8))
    else:
        ao = orc.Amg(Ao, orc.amg_params(True))
        ao.set_ilu_smoother(num_levels=1, num_sweeps=1, part=part)
        ref = orc.pcg(Ao, np.ones(n), ao)
    assert res["converged"] and res["iters"] == ref["iters"]
    assert res["norm"] == pytest.approx(np.linalg.norm(ref["x"]), rel=1e-7)


@pytest.mark.parametrize("solver,precon", [("bicgstab", "amg"), ("bicgstab", "ilu"), ("fgmres", "amg")])
def test_yaml_bicgstab_and_fgmres(hd, orc, solver, precon):
    """examples/ex1a.yml (bicgstab + amg) and ex1b.yml (bicgstab + ilu) of the reference select these by name."""
    Ao, b = orc.lap7(12, 12, 12)
    po = orc.Amg(Ao, orc.amg_params(True)) if precon == "amg" else orc.IluPrecond(Ao)
    ref = (orc.bicgstab if solver == "bicgstab" else orc.fgmres)(Ao, b, po)
    h = hd.Hypredrv(f"solver: {solver}\npreconditioner: {precon}\n")
    h.set_laplacian7((12, 12, 12))
    r = h.solve()
    assert r["converged"] and r["iters"] == ref["iters"]
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-7
    h.close()


# ------------------------------------------------- preconditioner.reuse (static policy; src/HYPREDRV.c:233-256, 3010-3020)

def _sequence(orc, k):
    """k systems of a 'time stepping' sequence on one sparsity pattern: lap7 + (1 + 0.5 s) I."""
    import scipy.sparse as sp
    A0, b = orc.lap7(12, 11, 10)
    S = A0.to_scipy()
    return [(S + (0.2 + 0.6 * s) * sp.identity(S.shape[0])).tocsr() for s in range(k)], b


@pytest.mark.parametrize("reuse,rebuild_on", [("\n    frequency: 1\n", {0, 2, 4}), (" always\n", {0}),
                                              ("\n    linear_system_ids: [0, 3]\n", {0, 3}), ("\n    enabled: off\n", {0, 1, 2, 3, 4})])
def test_precon_reuse_over_a_sequence_matches_oracle(hd, orc, reuse, rebuild_on):
    """Five systems through SetMatrixFromCSR + Create/Setup/Apply/Destroy each (library flow).  On the systems the policy
    does not rebuild, the hierarchy of the last rebuild is applied with the NEW matrix on level 0 (hypre's BoomerAMGSolve
    semantics): iteration counts and solutions equal the oracle doing exactly that, and the setup timer shows no setup."""
    mats, b = _sequence(orc, 5)
    n = mats[0].shape[0]
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    print_level: 0\n  reuse:" + reuse)
    amg_o = None
    for s, S in enumerate(mats):
        Ao = orc.Csr.from_scipy(S)
        if s in rebuild_on:
            amg_o = orc.Amg(Ao, orc.amg_params(True))
        else:
            amg_o.rebind_level0(Ao)
        ref = orc.pcg(Ao, b, amg_o)
        h.set_matrix_csr(0, n - 1, S.indptr, S.indices, S.data)
        h.set_rhs_array(0, n - 1, b)
        h.finish_system()
        r = h.solve()
        assert r["converged"] and r["iters"] == ref["iters"], (s, r["iters"], ref["iters"])
        assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-9
        if s in rebuild_on:
            assert r["setup_s"] > 1e-4
        else:
            assert r["setup_s"] < 1e-4, (s, r["setup_s"])
    h.close()


def test_precon_reuse_with_hybrid_gauss_seidel_on_row_blocks(hd, orc, monkeypatch):
    """The same sequence with the reference's CPU-build smoother (hybrid l1 Gauss-Seidel 13 / 14) on three row blocks, every level on the
    sweep-order kernels: a kept hierarchy sweeps the NEW level-0 matrix with the divisors of its setup, whose sweep-order copy is kept
    across cycles and solves (GsPlan::sd_src), the right-hand side's across the sweeps of one cycle (round 5)."""
    monkeypatch.setenv("HDA_BLOCKS", "3")
    monkeypatch.setenv("HDA_GS_SORTED_MIN", "0")
    monkeypatch.setenv("HDA_GS_FREE_CHECK", "1")
    mats, b = _sequence(orc, 4)
    n = mats[0].shape[0]
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    print_level: 0\n    relaxation:\n      down_type: 13\n      up_type: 14\n"
                    "  reuse: always\n")
    amg_o = None
    for s, S in enumerate(mats):
        Ao = orc.Csr.from_scipy(S)
        if s == 0:
            amg_o = orc.Amg(Ao, orc.amg_params(True, relax_down=13, relax_up=14, blocks=3))
        else:
            amg_o.rebind_level0(Ao)
        ref = orc.pcg(Ao, b, amg_o)
        h.set_matrix_csr(0, n - 1, S.indptr, S.indices, S.data)
        h.set_rhs_array(0, n - 1, b)
        h.finish_system()
        for rep in range(2):  # (twice: the second solve finds every sweep-order copy of the first)
            r = h.solve()
            assert r["converged"] and r["iters"] == ref["iters"], (s, rep, r["iters"], ref["iters"])
            assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-9
    h.close()


def test_precon_reuse_repeated_solves_of_one_system(hd):
    """The laplacian driver's loop (5 x Create/Setup/Apply/Destroy on the same system) with reuse: always builds once."""
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    print_level: 0\n  reuse: always\n")
    h.set_laplacian7((16, 16, 16))
    rs = [h.solve() for _ in range(5)]
    assert all(r["converged"] and r["iters"] == rs[0]["iters"] for r in rs)
    assert rs[0]["setup_s"] > 1e-4 and all(r["setup_s"] < 1e-4 for r in rs[1:])
    h.close()


def test_precon_reuse_rejects_a_different_size(hd, orc):
    import scipy.sparse as sp
    h = hd.Hypredrv("solver: pcg\npreconditioner:\n  amg:\n    print_level: 0\n  reuse: always\n")
    for n in (400, 500):
        S = (sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])).tocsr()
        h.set_matrix_csr(0, n - 1, S.indptr, S.indices, S.data)
        h.set_rhs_array(0, n - 1, np.ones(n))
        h.finish_system()
        if n == 400:
            assert h.solve()["converged"]
        else:
            with pytest.raises(hd.HypredrvError, match="same number of local rows"):
                h.solve()
            hd.lib().HYPREDRV_ErrorCodeClear()
    h.close()


# ------------------------------------------------- MGR through the HYPREDRV_* / YAML surface

EX3_MGR_YAML = ("solver:\n  gmres:\n    relative_tol: 1.0e-8\npreconditioner:\n  mgr:\n    level:\n      0:\n        f_dofs: [2]\n"
                "        prolongation_type: jacobi\n      1:\n        f_dofs: [1]\n        g_relaxation: l1-hsgs\n        restriction_type: columped\n"
                "    coarsest_level: amg\n")


def test_yaml_mgr_matches_oracle(hd, orc):
    """The mgr block of the reference's examples/ex3.yml (two reduction levels, jacobi prolongation, l1-hsgs global
    relaxation, column-lumped restriction, BoomerAMG on the coarsest system) on a 3-field model problem with an
    interleaved dofmap: GMRES iteration count and solution of the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pins import three_field_system
    S, labels = three_field_system(14, seed=5)
    n = S.shape[0]
    Ao = orc.Csr.from_scipy(S)
    lev = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")]
    ref = orc.gmres(Ao, np.ones(n), orc.MgrPrecond(Ao, labels, lev), orc.krylov_params(True, rtol=1e-8))
    h = hd.Hypredrv(EX3_MGR_YAML)
    h.set_matrix_csr(0, n - 1, S.indptr, S.indices, S.data)
    h.set_rhs_array(0, n - 1, np.ones(n))
    h.finish_system()
    hd.check(hd.lib().HYPREDRV_LinearSystemSetInterleavedDofmap(h.h, n // 3, 3))
    rs = [h.solve() for _ in range(2)]
    assert all(r["converged"] and r["iters"] == ref["iters"] for r in rs)
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-8
    h.close()


def test_cli_mgr_with_dofmap_file(tmp_path, orc):
    """hypredrive-cli on files laid out like the reference's compflow6k data set (IJ matrix / rhs / dofmap parts,
    examples/ex3.yml): the dofmap file is read (containers.c:443-620 format) and MGR runs from the YAML alone."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pins import three_field_system
    S, labels = three_field_system(10, seed=2)
    n = S.shape[0]
    d = tmp_path / "np1"
    d.mkdir()
    with open(d / "IJ.out.A.00000", "w") as f:
        f.write(f"0 {n - 1} 0 {n - 1}\n")
        C = S.tocoo()
        for i, j, v in sorted(zip(C.row, C.col, C.data)):
            f.write(f"{i} {j} {v:.17e}\n")
    with open(d / "IJ.out.b.00000", "w") as f:
        f.write(f"0 {n - 1}\n" + "".join(f"{i} 1.0\n" for i in range(n)))
    with open(d / "dofmap.out.00000", "w") as f:
        f.write(f"{n}\n" + "".join(f"{v}\n" for v in labels))
    cfg = tmp_path / "mgr.yml"
    cfg.write_text(f"linear_system:\n  rhs_filename: {d}/IJ.out.b\n  matrix_filename: {d}/IJ.out.A\n  dofmap_filename: {d}/dofmap.out\n" + EX3_MGR_YAML)
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", str(cfg)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    Ao = orc.Csr.from_scipy(S)
    lev = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")]
    ref = orc.gmres(Ao, np.ones(n), orc.MgrPrecond(Ao, labels, lev), orc.krylov_params(True, rtol=1e-8))
    assert row and int(row.group(2)) == ref["iters"] and float(row.group(1)) < 1e-8


def test_reference_darcy_driver_unmodified():
    """examples/src/C_darcy/darcy.c of the reference (mixed RT0 Darcy flow: face fluxes label 1, cell pressures label 0;
    its built-in configuration is GMRES(60) + two-level MGR, f_dofs [1], jacobi F-relaxation and prolongation, injection
    restriction, rap coarse grid, BoomerAMG coarsest), compiled UNMODIFIED against this library."""
    exe = os.path.join(ROOT, "oracle", "_ref", "darcy_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/darcy_ref not built (needs /root/reference + MPICH at build time)")
    for args in ([], ["-n", "12", "12", "6"]):
        r = subprocess.run([exe, "-v", "1"] + args, capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        assert "HYPREDRIVE Failure" not in r.stdout + r.stderr
        # the driver checks its own answer: pressure against the exact (linear) solution of the driven flow
        m = re.search(r"relative pressure L2 error\s*:\s*(\S+)", r.stdout)
        assert m and float(m.group(1)) < 1e-8, r.stdout[-2000:]
        row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
        assert row and float(row.group(1)) < 1e-9 and int(row.group(2)) <= 30


def test_cli_include_and_ex4_style_mgr(tmp_path, orc):
    """examples/ex5.yml of the reference keeps its solver and preconditioner blocks in files of their own
    ("include: ex5-gmres.yml", "include: ex5-mgr.yml"); its MGR block (= ex4.yml's) uses ILU as global relaxation on the
    second reduction level.  Same structure here on the generated 3-field system."""
    (tmp_path / "inc-gmres.yml").write_text("gmres:\n  max_iter: 100\n  relative_tol: 1.0e-8\n")
    (tmp_path / "inc-mgr.yml").write_text(
        "mgr:\n  tolerance: 0.0\n  max_iter: 1\n  print_level: 0\n  coarse_th: 0.0\n  level:\n    0:\n      f_dofs: [2]\n      f_relaxation: jacobi\n"
        "      g_relaxation: none\n      restriction_type: injection\n      prolongation_type: jacobi\n      coarse_level_type: rap\n\n    1:\n"
        "      f_dofs: [1]\n      f_relaxation: jacobi\n      g_relaxation: ilu\n      restriction_type: columped\n      prolongation_type: injection\n"
        "      coarse_level_type: rap\n\n  coarsest_level:\n    amg:\n      tolerance: 0.0\n      max_iter: 1\n      print_level: 0\n      coarsening:\n"
        "        type: pmis\n        strong_th: 0.3\n")
    d = os.path.join(ROOT, "data", "threefield", "np1")
    (tmp_path / "main.yml").write_text(f"linear_system:\n  rhs_filename: {d}/IJ.out.b\n  matrix_filename: {d}/IJ.out.A\n  dofmap_filename: {d}/dofmap.out\n"
                                       "solver:\n  include: inc-gmres.yml\n\npreconditioner:\n  include: inc-mgr.yml\n")
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", str(tmp_path / "main.yml")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_threefield import system
    S, labels = system(16)
    Ao = orc.Csr.from_scipy(S)
    lev = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="ilu", restriction_type="columped")]
    ref = orc.gmres(Ao, np.ones(S.shape[0]), orc.MgrPrecond(Ao, labels, lev, orc.amg_params(True, strong_th=0.3)), orc.krylov_params(True, rtol=1e-8, max_iter=100))
    assert row and int(row.group(2)) == ref["iters"] and float(row.group(1)) < 1e-8
    # a missing include is an error, a cycle too
    (tmp_path / "bad.yml").write_text("solver:\n  include: nowhere.yml\npreconditioner: amg\n")
    r = subprocess.run([cli, "-q", str(tmp_path / "bad.yml")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "nowhere.yml" in r.stdout + r.stderr
    (tmp_path / "loop.yml").write_text("solver:\n  include: loop.yml\npreconditioner: amg\n")
    r = subprocess.run([cli, "-q", str(tmp_path / "loop.yml")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "cycle" in r.stdout + r.stderr


MGR_DIST_CASES = {
    "ex3": (EX3_MGR_YAML, [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")], 2),
    "jacobi-columped": ("solver:\n  gmres:\n    relative_tol: 1.0e-8\npreconditioner:\n  mgr:\n    level:\n      0:\n        f_dofs: [2]\n"
                        "        prolongation_type: jacobi\n        restriction_type: jacobi\n      1:\n        f_dofs: [1]\n"
                        "        prolongation_type: l1-jacobi\n        restriction_type: columped\n    coarsest_level: amg\n",
                        [dict(f_dofs=[2], prolongation_type="jacobi", restriction_type="jacobi"),
                         dict(f_dofs=[1], prolongation_type="l1-jacobi", restriction_type="columped")], 1),
}


MGR_DIST_CASES["famg"] = ("solver:\n  gmres:\n    relative_tol: 1.0e-8\npreconditioner:\n  mgr:\n    level:\n      0:\n        f_dofs: [0]\n"
                          "        f_relaxation:\n          amg:\n            max_iter: 1\n            coarsening:\n              strong_th: 0.3\n"
                          "        prolongation_type: jacobi\n    coarsest_level: amg\n",
                          [dict(f_dofs=[0], prolongation_type="jacobi", f_relaxation="amg", f_amg_kw=dict(strong_th=0.3))], 1)


# nested Krylov components on row partitions: GMRES(3) + BoomerAMG on A_FF, GMRES(2) + BoomerAMG on the coarsest system, FlexGMRES outside
MGR_DIST_CASES["nested"] = ("solver:\n  fgmres:\n    relative_tol: 1.0e-8\npreconditioner:\n  mgr:\n    level:\n      0:\n        f_dofs: [0]\n"
                            "        f_relaxation:\n          gmres:\n            max_iter: 3\n            relative_tol: 0.0\n            preconditioner:\n"
                            "              amg:\n                coarsening:\n                  strong_th: 0.3\n"
                            "        prolongation_type: jacobi\n    coarsest_level:\n      gmres:\n        max_iter: 2\n        relative_tol: 0.0\n"
                            "        preconditioner: amg\n",
                            [dict(f_dofs=[0], prolongation_type="jacobi", f_relaxation="amg", f_amg_kw=dict(strong_th=0.3),
                                  f_krylov=dict(method="gmres", max_iter=3, rtol=0.0), coarsest_krylov=dict(method="gmres", max_iter=2, rtol=0.0))], 1)


@pytest.mark.parametrize("world,case,rep_rows", [(2, "jacobi-columped", 0), (3, "jacobi-columped", 100000), (4, "ex3", 0), (3, "ex3", 100000),
                                                 (3, "famg", 0), (2, "famg", 100000), (3, "nested", 0)])
def test_row_partitioned_mgr(hd, orc, tmp_path, world, case, rep_rows):
    """MGR on a row-partitioned matrix (rows cut inside cells too): ghost labels / C-F marks / coarse ids through the halo plan,
    reduced operators by two row-partitioned products, BoomerAMG on the partitioned coarsest system.  Without global relaxation
    the preconditioner is the single-rank one up to rounding (same iterations as the oracle, +-1 with the hybrid smoother)."""
    yaml, lev, slack = MGR_DIST_CASES[case]
    n = 14
    out = str(tmp_path / "res.json")
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_REPLICATE_ROWS=str(rep_rows), HDA_TEST_YAML=yaml)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "mgr", out, str(n)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pins import three_field_system
    S, labels = three_field_system(n, seed=4)
    Ao = orc.Csr.from_scipy(S)
    lev = [dict(l) for l in lev]
    for l in lev:
        if "f_amg_kw" in l:
            l["f_amg"] = orc.amg_params(True, **l.pop("f_amg_kw"))
    outer = orc.fgmres if case == "nested" else orc.gmres
    ref = outer(Ao, np.ones(S.shape[0]), orc.MgrPrecond(Ao, labels, lev), orc.krylov_params(True, rtol=1e-8))
    assert res["converged"] and abs(res["iters"] - ref["iters"]) <= slack, (res["iters"], ref["iters"])
    assert res["norm"] == pytest.approx(np.linalg.norm(ref["x"]), rel=1e-6)


def _solve_on_thread_ranks(hd, cuts, S, b, yaml, labels=None):
    """len(cuts) - 1 ranks as threads of this process (hypredrive_amd/_lib.py run_thread_ranks): rank r hands over rows cuts[r] .. cuts[r + 1] - 1
    of the global matrix S through the public API (HYPREDRV_LinearSystemSetMatrixFromCSR / SetRHSFromArray / SetDofmap), solves, and
    returns its block of the solution.  Returns (rank 0's result dict, the gathered solution, rank 0's partitioned levels)."""
    import ctypes as C
    from hypredrive_amd import _lib

    def body(rank, world):
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        blk = S[lo:hi]
        h = hd.Hypredrv(yaml)
        try:
            h.set_matrix_csr(lo, hi - 1, blk.indptr, blk.indices, blk.data)
            h.set_rhs_array(lo, hi - 1, b[lo:hi])
            h.finish_system()
            if labels is not None:
                lab = np.ascontiguousarray(labels[lo:hi], dtype=np.int32)
                hd.check(hd.lib().HYPREDRV_LinearSystemSetDofmap(h.h, hi - lo, lab.ctypes.data_as(C.POINTER(C.c_int))))
            L = hd.lib()
            hd.check(L.HYPREDRV_LinearSystemResetInitialGuess(h.h))
            hd.check(L.HYPREDRV_LinearSolverCreate(h.h))
            hd.check(L.HYPREDRV_LinearSolverSetup(h.h))
            part = _lib.load().hda_amd_partitioned_levels(h.h)
            hd.check(L.HYPREDRV_LinearSolverApply(h.h))
            r = h.last()
            hd.check(L.HYPREDRV_LinearSolverDestroy(h.h))
            return r, np.array(h.solution(), copy=True), part
        finally:
            h.close()

    outs = _lib.run_thread_ranks(len(cuts) - 1, body)
    assert len({o[0]["iters"] for o in outs}) == 1                      # every rank stopped at the same iteration
    return outs[0][0], np.concatenate([o[1] for o in outs]), outs[0][2]


@pytest.mark.parametrize("rep_rows,transport", [(0, "host"), (1500, "host"), (0, "device"), (1500, "device")])
def test_eight_thread_ranks_irregular_csr_match_oracle(hd, orc, monkeypatch, rep_rows, transport):
    """Eight ranks, an IRREGULAR matrix (tests/dist_worker.py random_mmatrix: mostly local couplings plus long-range ones, so a block's
    peers and ghost layers are whatever the matrix says), row blocks of very different sizes -- one of them 7 rows -- handed over
    through HYPREDRV_LinearSystemSetMatrixFromCSR like reference tests/test_setmatrix_from_csr_mpi.c does per rank.  The partitioned
    setup must build the ORACLE's hierarchy (same PMIS draw per global row): AMG-PCG iteration count within 1 of the oracle's, the gathered
    solution its solution to the stopping tolerance, the true residual of the gathered solution below it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import random_mmatrix
    n = 9000
    S = random_mmatrix(5, n)
    b = np.ones(n)
    cuts = np.array([0, 1400, 1407, 3000, 3900, 5200, 6100, 8000, n])
    monkeypatch.setenv("HDA_REPLICATE_ROWS", str(rep_rows))
    monkeypatch.setenv("HDA_THREAD_TRANSPORT", transport)  # device: exchanges and all-reduces are only enqueued, as with RCCL
    res, x, part = _solve_on_thread_ranks(hd, cuts, S, b, "solver: pcg\npreconditioner: amg\n")
    Ao = orc.Csr.from_scipy(S)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    assert part >= 2                                                      # the hierarchy really is cut into row blocks
    assert res["converged"] and abs(res["iters"] - ref["iters"]) <= 1, (res["iters"], ref["iters"])
    assert np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-5
    assert np.linalg.norm(b - S @ x) / np.linalg.norm(b) < 2e-6


@pytest.mark.parametrize("transport", ["host", "device"])
def test_config5_standin_on_eight_thread_ranks(hd, orc, monkeypatch, transport):
    """BASELINE config 5's shape -- GMRES(30) + BoomerAMG with the ILU(0) smoother on level 0, heterogeneous anisotropic reservoir
    operator (hypredrive_amd/synthetic.py) -- on EIGHT row blocks (slabs of the 40^3 grid, the last two uneven).  The level-0 ILU is
    block Jacobi by rank (`bj-iluk`, reference src/internal/ilu.c), so the preconditioner is not the one-rank one: the bar is convergence
    to the same tolerance within a few iterations of the oracle's one-rank count and the true residual of the gathered solution."""
    import scipy.sparse as sp
    from hypredrive_amd.synthetic import spe10_like
    n = 40
    N = n ** 3
    ip, ix, v, b = spe10_like(n)
    S = sp.csr_matrix((v, ix, ip), shape=(N, N))
    cuts = np.array([0, 8000, 16000, 24000, 32000, 40000, 48000, 59000, N])
    monkeypatch.setenv("HDA_REPLICATE_ROWS", "3000")
    monkeypatch.setenv("HDA_THREAD_TRANSPORT", transport)
    res, x, part = _solve_on_thread_ranks(hd, cuts, S, b, SPE10_YAML)
    Ao = orc.Csr.from_scipy(S)
    ao = orc.Amg(Ao, orc.amg_params(True))
    ao.set_ilu_smoother(num_levels=1, num_sweeps=1, tri_solve=0)
    ref = orc.gmres(Ao, b, ao, orc.krylov_params(True))
    assert part >= 2
    assert res["converged"] and abs(res["iters"] - ref["iters"]) <= 3, (res["iters"], ref["iters"])
    assert np.linalg.norm(b - S @ x) / np.linalg.norm(b) < 2e-6
    assert np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-4


@pytest.mark.parametrize("case,transport", [("ex3", "host"), ("jacobi-columped", "host"), ("famg", "host"), ("ex3", "device"), ("famg", "device")])
def test_mgr_on_eight_thread_ranks(hd, orc, monkeypatch, case, transport):
    """BASELINE config 4's shape (GMRES + MGR by dof labels, examples/ex3.yml on the three-field stand-in) on eight row blocks cut
    anywhere, also inside a cell: the oracle's one-rank iteration count (+- the slack of the hybrid smoother) and solution."""
    yaml, lev, slack = MGR_DIST_CASES[case]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pins import three_field_system
    S, labels = three_field_system(14, seed=4)
    N = S.shape[0]
    cuts = np.array([0, N // 9, N // 9 + 4] + [N * k // 8 for k in range(3, 8)] + [N])
    assert len(cuts) == 9 and np.all(np.diff(cuts) > 0)
    monkeypatch.setenv("HDA_REPLICATE_ROWS", "0")
    monkeypatch.setenv("HDA_THREAD_TRANSPORT", transport)
    b = np.ones(N)
    res, x, _ = _solve_on_thread_ranks(hd, cuts, S, b, yaml, labels=labels)
    Ao = orc.Csr.from_scipy(S)
    lev = [dict(l) for l in lev]
    for l in lev:
        if "f_amg_kw" in l:
            l["f_amg"] = orc.amg_params(True, **l.pop("f_amg_kw"))
    ref = orc.gmres(Ao, b, orc.MgrPrecond(Ao, labels, lev), orc.krylov_params(True, rtol=1e-8))
    assert res["converged"] and abs(res["iters"] - ref["iters"]) <= slack, (res["iters"], ref["iters"])
    assert np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-6


def test_random_partitions_of_random_matrices_match_oracle():
    """tests/fuzz_ranks.py, 32 cases: random irregular M-matrices (400-7000 rows) cut into 2-8 row blocks at random -- blocks of a few
    rows, EMPTY blocks -- on thread ranks, host-staged or asynchronous device transport, every level partitioned or a replicated tail at a
    random depth, HDA_DIST_CHECK on a third; PCG / GMRES with the default, Chebyshev, aggressive, truncated-interpolation hierarchies (the
    oracle's iteration count +-1 and its solution to 1e-5: observed 0 and 1e-15) and the rank-dependent hybrid GS / ILU smoothers
    (convergence, true residual).  (180 cases ran clean in tools/gpurun/r03_zc.sh.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_ranks.py"), "32", "9000"], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=ROOT), timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"cases": 32, "failed": 0}


def test_a_failing_thread_rank_releases_its_peers(hd):
    """A rank that raises must not leave the other ranks blocked in a collective: they get an error, and the first real error surfaces."""
    from hypredrive_amd import _lib

    def body(rank, world):
        if rank == 2:
            raise ValueError("rank 2 gives up")
        h = hd.Hypredrv("solver: pcg\npreconditioner: amg\n")
        try:
            h.set_laplacian7((12, 12, 12), (1, 1, 4))
            return h.solve()
        finally:
            hd.lib().HYPREDRV_ErrorCodeClear()
            h.close()

    with pytest.raises(ValueError, match="rank 2 gives up"):
        _lib.run_thread_ranks(4, body)


def test_precon_reuse_with_mgr(hd, orc):
    """preconditioner.reuse with MGR: a kept MGR is applied to the next system with level 0 taken from the new matrix."""
    import scipy.sparse as sp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pins import three_field_system
    S0, labels = three_field_system(12, seed=7)
    n = S0.shape[0]
    mats = [(S0 + 0.3 * s * sp.identity(n)).tocsr() for s in range(3)]
    lev = [dict(f_dofs=[2], prolongation_type="jacobi"), dict(f_dofs=[1], g_relaxation="l1-hsgs", restriction_type="columped")]
    h = hd.Hypredrv(EX3_MGR_YAML + "  reuse: always\n")
    hd.check(hd.lib().HYPREDRV_LinearSystemSetInterleavedDofmap(h.h, n // 3, 3))
    Mo = None
    for s, S in enumerate(mats):
        Ao = orc.Csr.from_scipy(S)
        if s == 0:
            Mo = orc.MgrPrecond(Ao, labels, lev)
        else:
            Mo.rebind_level0(Ao)
        ref = orc.gmres(Ao, np.ones(n), Mo, orc.krylov_params(True, rtol=1e-8))
        h.set_matrix_csr(0, n - 1, S.indptr, S.indices, S.data)
        h.set_rhs_array(0, n - 1, np.ones(n))
        h.finish_system()
        r = h.solve()
        assert r["converged"] and r["iters"] == ref["iters"], (s, r["iters"], ref["iters"])
        assert (r["setup_s"] > 1e-4) == (s == 0)
    h.close()


def test_reference_lidcavity_driver_unmodified():
    """examples/src/C_lidcavity/lidcavity.c of the reference (lid-driven cavity: Newton on stabilised Q1-Q1 Navier-Stokes,
    three unknowns per node, two annotation levels), UNMODIFIED, with its default solver block -- FGMRES(100) + systems
    BoomerAMG (num_functions 3, strong_th 0.6) + ILU smoother on five levels -- passed through -i with bj-iluk / fill 0 in
    place of bj-ilut (examples/lidcavity-ilu0.yml).  Newton converges in every time step; a handful of FGMRES iterations each."""
    exe = os.path.join(ROOT, "oracle", "_ref", "lidcavity_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/lidcavity_ref not built (needs /root/reference + MPICH at build time)")
    r = subprocess.run([exe, "-i", "examples/lidcavity-ilu0.yml", "-n", "16", "16", "-tf", "2", "-v", "1"], capture_output=True, text=True,
                       cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "HYPREDRIVE Failure" not in r.stdout + r.stderr
    rows = re.findall(r"^\|\s+(\d+)\.(\d+)\.\d+ \|.*\|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
    assert len(rows) >= 8
    assert all(int(it) <= 30 and float(rr) < 1e-6 for _, _, _, rr, it in rows)
    for step in sorted({ts for ts, _, _, _, _ in rows}):
        r0 = [float(x[2]) for x in rows if x[0] == step]
        assert r0[-1] < 1e-3 * r0[0]   # Newton residual (= initial residual of each linear solve) drops by orders of magnitude


def test_yaml_mgr_component_solvers(hd, orc):
    """MGR component solvers through the YAML surface: BoomerAMG on A_FF as F-relaxation, an ILU block (Jacobi-iterative
    solves) as global relaxation on the second level, ILU iterations on the coarsest system -- against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_pins import three_field_system
    S, labels = three_field_system(13, seed=9)
    n = S.shape[0]
    yaml = ("solver:\n  gmres:\n    relative_tol: 1.0e-8\npreconditioner:\n  mgr:\n    level:\n      0:\n        f_dofs: [0]\n        f_relaxation:\n          amg:\n"
            "            max_iter: 1\n            coarsening:\n              strong_th: 0.4\n        prolongation_type: jacobi\n      1:\n        f_dofs: [1]\n"
            "        g_relaxation:\n          ilu:\n            tri_solve: 0\n            lower_jac_iters: 4\n        restriction_type: columped\n"
            "    coarsest_level:\n      ilu:\n        max_iter: 3\n")
    Ao = orc.Csr.from_scipy(S)
    lev = [dict(f_dofs=[0], f_relaxation="amg", f_amg=orc.amg_params(True, strong_th=0.4), prolongation_type="jacobi"),
           dict(f_dofs=[1], g_relaxation="ilu", ilu=dict(tri_solve=0, lower_jac_iters=4), restriction_type="columped", coarsest_ilu=dict(max_iter=3))]
    ref = orc.gmres(Ao, np.ones(n), orc.MgrPrecond(Ao, labels, lev, coarsest="ilu"), orc.krylov_params(True, rtol=1e-8))
    h = hd.Hypredrv(yaml)
    h.set_matrix_csr(0, n - 1, S.indptr, S.indices, S.data)
    h.set_rhs_array(0, n - 1, np.ones(n))
    h.finish_system()
    hd.check(hd.lib().HYPREDRV_LinearSystemSetInterleavedDofmap(h.h, n // 3, 3))
    r = h.solve()
    assert r["converged"] and r["iters"] == ref["iters"], (r["iters"], ref["iters"])
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-8
    h.close()


def test_yaml_chebyshev_relaxation(hd, orc):
    """relaxation.down_type / up_type 16 with a chebyshev block (the first variants of the reference's examples/ex8.yml use
    this smoother): iteration count and solution of the oracle."""
    Ao, b = orc.lap7(12, 12, 12)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True, relax_down=16, relax_up=16, cheby_order=4, cheby_fraction=0.1)),
                  orc.krylov_params(False, rtol=1e-9, max_iter=500))
    h = hd.Hypredrv("solver:\n  pcg:\n    relative_tol: 1.0e-9\n    max_iter: 500\npreconditioner:\n  amg:\n    relaxation:\n      down_type: 16\n"
                    "      up_type: chebyshev\n      chebyshev:\n        order: 4\n        fraction: 0.1\n")
    h.set_laplacian7((12, 12, 12))
    r = h.solve()
    assert r["converged"] and r["iters"] == ref["iters"]
    assert np.linalg.norm(h.solution() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-8
    h.close()


def test_row_partitioned_chebyshev(hd, tmp_path):
    """Chebyshev smoother on row blocks (3 ranks, partitioned setup): the eigenvalue estimate runs its products and dot
    products across the ranks; same iteration count (+-1: every rank block draws its own start vector) as one rank."""
    n, seed, world = 4000, 31, 3
    out = str(tmp_path / "res.json")
    yaml = "solver: pcg\npreconditioner:\n  amg:\n    relaxation:\n      down_type: 16\n      up_type: 16\n"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_REPLICATE_ROWS="500", HDA_TEST_YAML=yaml)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "csr", out, str(n), str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import random_mmatrix
    A = random_mmatrix(seed, n)
    h = hd.Hypredrv(yaml)
    h.set_matrix_csr(0, n - 1, A.indptr, A.indices, A.data)
    h.set_rhs_array(0, n - 1, np.ones(n))
    h.finish_system()
    ref = h.solve()
    assert res["converged"] and abs(res["iters"] - ref["iters"]) <= 1
    assert res["norm"] == pytest.approx(h.solution_norm("L2"), rel=1e-6)
    h.close()


def test_cli_sequence_of_systems_with_reuse(tmp_path, orc):
    """A sequence of linear systems in the reference's directory layout -- <dirname>_<suffix %05d>/<filename>
    (src/internal/linsys.c:832-866), init_suffix .. last_suffix -- solved by hypredrive-cli with preconditioner.reuse
    frequency 1: systems 0 and 2 rebuild the hierarchy, system 1 reuses it with its own matrix on level 0."""
    import scipy.sparse as sp
    Ao, b = orc.lap7(9, 8, 7)
    S0 = Ao.to_scipy()
    n = S0.shape[0]
    mats = [(S0 + 0.4 * s * sp.identity(n)).tocsr() for s in range(3)]
    for s, S in enumerate(mats):
        d = tmp_path / f"ls_{s:05d}"
        d.mkdir()
        with open(d / "IJ.out.A.00000", "w") as f:
            f.write(f"0 {n - 1} 0 {n - 1}\n")
            C = S.tocoo()
            for i, j, v in sorted(zip(C.row, C.col, C.data)):
                f.write(f"{i} {j} {v:.17e}\n")
        with open(d / "IJ.out.b.00000", "w") as f:
            f.write(f"0 {n - 1}\n" + "".join(f"{i} {b[i]:.17e}\n" for i in range(n)))
    cfg = tmp_path / "seq.yml"
    cfg.write_text(f"general:\n  use_millisec: on\nlinear_system:\n  dirname: {tmp_path}/ls\n  init_suffix: 0\n  last_suffix: 2\n"
                   "  matrix_filename: IJ.out.A\n  rhs_filename: IJ.out.b\nsolver: pcg\npreconditioner:\n  amg:\n    print_level: 0\n  reuse:\n    frequency: 1\n")
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    r = subprocess.run([cli, "-q", str(cfg)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^\|\s+(\d+) \|\s+([\d.]*) \|\s+([\d.]+) \|\s+([\d.]+) \|\s+(\S+) \|\s+(\S+) \|\s+(\d+) \|", r.stdout, re.M)
    assert len(rows) == 3, r.stdout
    amg = None
    for s, S in enumerate(mats):
        A = orc.Csr.from_scipy(S)
        if s % 2 == 0:
            amg = orc.Amg(A, orc.amg_params(True))
        else:
            amg.rebind_level0(A)
        ref = orc.pcg(A, b, amg)
        assert int(rows[s][6]) == ref["iters"] and float(rows[s][5]) < 1e-6
    assert float(rows[1][2]) < 0.2 < float(rows[0][2])   # setup time [ms]: nothing to set up on the reused system


def test_cli_precmat_filename(tmp_path, orc):
    """linear_system.precmat_filename: the preconditioner is set up on a second matrix (here a shifted operator) while
    the Krylov method iterates on A (reference src/internal/linsys.c:2620-2660) -- not silently replaced by A."""
    import scipy.sparse as sp
    Ao, b = orc.lap7(9, 9, 9)
    S = Ao.to_scipy()
    n = S.shape[0]
    M = (S + 2.0 * sp.identity(n)).tocsr()
    for name, mat in (("A", S), ("M", M)):
        with open(tmp_path / f"IJ.out.{name}.00000", "w") as f:
            f.write(f"0 {n - 1} 0 {n - 1}\n")
            C = mat.tocoo()
            for i, j, v in sorted(zip(C.row, C.col, C.data)):
                f.write(f"{i} {j} {v:.17e}\n")
    with open(tmp_path / "IJ.out.b.00000", "w") as f:
        f.write(f"0 {n - 1}\n" + "".join(f"{i} {b[i]:.17e}\n" for i in range(n)))
    cli = os.path.join(ROOT, "hypredrive_amd", "bin", "hypredrive-cli")
    its = {}
    for tag, extra in (("A", ""), ("M", f"  precmat_filename: {tmp_path}/IJ.out.M\n")):
        cfg = tmp_path / f"{tag}.yml"
        cfg.write_text(f"linear_system:\n  matrix_filename: {tmp_path}/IJ.out.A\n  rhs_filename: {tmp_path}/IJ.out.b\n{extra}solver: pcg\npreconditioner: amg\n")
        r = subprocess.run([cli, "-q", str(cfg)], capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        row = re.search(r"^\|\s+0 \|.*\|\s+(\S+) \|\s+(\d+) \|$", r.stdout, re.M)
        assert row and float(row.group(1)) < 1e-6
        its[tag] = int(row.group(2))
    amg = orc.Amg(orc.Csr.from_scipy(M), orc.amg_params(True))   # hierarchy of M ...
    amg.rebind_level0(Ao)                                          # ... applied with the system matrix on level 0, as hypre's BoomerAMGSolve does
    ref = orc.pcg(Ao, b, amg)
    assert its["M"] == ref["iters"] and its["M"] > its["A"]


# ------------------------------------------------------------------ solver.scaling

def _scaled_oracle(orc, A, b, labels, kind, values, solver):
    """The reference's scaled solve restated on the host (src/internal/scaling.c:246-262, :788-868, :900-928, :1050-1075 and the
    scaled branch of src/HYPREDRV.c:3179-3259): transform the system, run the oracle's Krylov + AMG on it from x0 = 0, map the
    solution back.  Also returns the caller's matrix and right-hand side as the inverse transforms leave them (products with
    the reciprocal weights: equal to the originals only up to rounding, which is what the next solve starts from)."""
    if kind == "rhs_l2":
        s = 1.0 / np.sqrt(np.linalg.norm(b))
        s2 = s * s
        As, bs, back = (A * s2).tocsr(), s * b, lambda y: y * s
        A_after, b_after = (As * (1.0 / s2)).tocsr(), bs * (1.0 / s)
    else:
        d = np.asarray(values)[labels]
        inv = 1.0 / d
        D, Di = sp.diags(d), sp.diags(inv)
        if kind == "dofmap_custom":
            As, bs, back = (D @ A @ D).tocsr(), d * b, lambda y: d * y
            A_after, b_after = (Di @ As @ Di).tocsr(), bs / d
        elif kind == "dofmap_row_custom":
            As, bs, back = (D @ A).tocsr(), d * b, lambda y: y
            A_after, b_after = (Di @ As).tocsr(), bs / d
        elif kind == "dofmap_col_custom":
            As, bs, back = (A @ D).tocsr(), b, lambda y: d * y
            A_after, b_after = (As @ Di).tocsr(), b
        else:
            As, bs, back = (Di @ A @ D).tocsr(), b / d, lambda y: d * y
            A_after, b_after = (D @ As @ Di).tocsr(), d * bs
    As.sort_indices()
    A_after.sort_indices()
    Ao = orc.Csr.from_scipy(As)
    amg = orc.Amg(Ao, orc.amg_params(True))
    ref = orc.pcg(Ao, bs, amg) if solver == "pcg" else orc.gmres(Ao, bs, amg, orc.krylov_params(True))
    return ref["iters"], back(ref["x"]), A_after, b_after


@pytest.mark.parametrize("kind,solver", [("rhs_l2", "pcg"), ("dofmap_custom", "pcg"), ("dofmap_row_custom", "gmres"),
                                         ("dofmap_col_custom", "gmres"), ("dofmap_similarity_custom", "gmres")])
def test_solver_scaling_matches_scaled_oracle(hd, orc, kind, solver):
    """solver.scaling of the reference (src/internal/scaling.c): the system is scaled in place before the AMG setup, solved in
    the scaled variables, and handed back unscaled.  Iterations and solution of the oracle run on the explicitly scaled
    system.  A second solve starts from the system as the inverse transform left it -- the reference multiplies by
    reciprocals, so equal coefficients come back different in the last bit and the hierarchy built on them is another one;
    the oracle is fed the same arithmetic.  rhs_l2 uses a right-hand side whose norm makes s a power of two."""
    A = _lap_coo(16 if kind == "rhs_l2" else 12)
    n = A.shape[0]
    labels = np.arange(n) % 3
    values = [2.0, 0.5, 3.0]
    b = np.ones(n) if kind == "rhs_l2" else 7.0 * np.ones(n)
    extra = "" if kind == "rhs_l2" else f"    custom_values: [{', '.join(str(v) for v in values)}]\n"
    h = hd.Hypredrv(f"solver:\n  {solver}:\n    max_iter: 100\n  scaling:\n    enabled: on\n    type: {kind}\n{extra}preconditioner: amg\n")
    h.set_matrix_csr(0, n - 1, A.indptr, A.indices, A.data)
    h.set_rhs_array(0, n - 1, b)
    h.finish_system()
    if kind != "rhs_l2":
        hd.check(hd.lib().HYPREDRV_LinearSystemSetInterleavedDofmap(h.h, n // 3, 3))
    A0, b0 = A, b
    for _ in range(2):
        its, xref, A, b = _scaled_oracle(orc, A, b, labels, kind, values, solver)
        r = h.solve()
        x = h.solution()
        assert r["converged"] and r["iters"] == its
        assert np.linalg.norm(x - xref) / np.linalg.norm(xref) < 1e-12
        assert np.linalg.norm(b0 - A0 @ x) / np.linalg.norm(b0) < 1e-5
    h.close()


def test_solver_scaling_errors(hd):
    """src/internal/scaling.c:400-470: custom scaling without a dofmap -> ERROR_MISSING_DOFMAP; a zero weight ->
    ERROR_INVALID_VAL; a weight count that differs from the number of labels -> ERROR_UNKNOWN; the system is left as it was.
    dofmap_mag (hypre's tagged scaling, not in the reference sources) is refused at parse time."""
    A = _lap_coo(6)
    n = A.shape[0]

    def build(vals, dofmap=True):
        h = hd.Hypredrv(f"solver:\n  pcg:\n    max_iter: 50\n  scaling:\n    enabled: on\n    type: dofmap_custom\n    custom_values: {vals}\npreconditioner: amg\n")
        h.set_matrix_csr(0, n - 1, A.indptr, A.indices, A.data)
        h.set_rhs_array(0, n - 1, np.ones(n))
        h.finish_system()
        if dofmap:
            hd.check(hd.lib().HYPREDRV_LinearSystemSetInterleavedDofmap(h.h, n // 3, 3))
        hd.check(hd.lib().HYPREDRV_LinearSolverCreate(h.h))
        code = hd.lib().HYPREDRV_LinearSolverSetup(h.h)
        hd.lib().HYPREDRV_ErrorCodeClear()
        return h, code

    h, code = build("[1.0, 2.0, 3.0]", dofmap=False)
    assert code & hd.ERROR_MISSING_DOFMAP
    h.close()
    h, code = build("[1.0, 0.0, 3.0]")
    assert code & hd.ERROR_INVALID_VAL
    h.close()
    h, code = build("[1.0, 2.0]")
    assert code & hd.ERROR_UNKNOWN
    h.close()
    with pytest.raises(hd.HypredrvError):
        hd.Hypredrv("solver:\n  pcg:\n    max_iter: 50\n  scaling:\n    enabled: on\n    type: dofmap_mag\npreconditioner: amg\n")


def test_row_partitioned_scaling(hd, orc, tmp_path):
    """dofmap_custom on three row blocks: the column factor of ghost columns comes from their owners (halo exchange of
    the weight vector); same iterations and solution norm as the oracle on the explicitly scaled system."""
    n, seed, world = 3000, 33, 3
    out = str(tmp_path / "res.json")
    yaml = "solver:\n  pcg:\n    max_iter: 100\n  scaling:\n    enabled: on\n    type: dofmap_custom\n    custom_values: [2.0, 0.5, 3.0, 1.5]\npreconditioner: amg\n"
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", HDA_REPLICATE_ROWS="400", HDA_TEST_YAML=yaml, HDA_TEST_DOFMAP_MOD="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), "csr", out, str(n), str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    res = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import random_mmatrix
    A = random_mmatrix(seed, n)
    its, xref, _A, _b = _scaled_oracle(orc, A, np.ones(n), np.arange(n) % 4, "dofmap_custom", [2.0, 0.5, 3.0, 1.5], "pcg")
    assert res["converged"] and res["iters"] == its
    assert res["norm"] == pytest.approx(np.linalg.norm(xref), rel=1e-7)


def test_reference_solution_error_norms(hd, capfd):
    """HYPREDRV_LinearSystemSetReferenceSolution + the tail of LinearSolverApply (src/HYPREDRV.c:3310-3323): rank 0 prints the
    L2 norms of the error, the solution and the reference solution."""
    import ctypes as C
    import scipy.sparse.linalg as spla
    A = _lap_coo(8)
    n = A.shape[0]
    xs = spla.spsolve(A.tocsc(), np.ones(n))
    L = hd.lib()
    v = C.c_void_p()
    L.HYPRE_IJVectorCreate.argtypes = [C.c_int, C.c_longlong, C.c_longlong, C.POINTER(C.c_void_p)]
    L.HYPRE_IJVectorSetObjectType.argtypes = [C.c_void_p, C.c_int]
    L.HYPRE_IJVectorInitialize.argtypes = [C.c_void_p]
    L.HYPRE_IJVectorSetValues.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
    L.HYPRE_IJVectorAssemble.argtypes = [C.c_void_p]
    L.HYPRE_IJVectorDestroy.argtypes = [C.c_void_p]
    assert L.HYPRE_IJVectorCreate(hd.MPI_COMM_WORLD, 0, n - 1, C.byref(v)) == 0
    assert L.HYPRE_IJVectorSetObjectType(v, 5555) == 0 and L.HYPRE_IJVectorInitialize(v) == 0
    idx = (C.c_longlong * n)(*range(n))
    assert L.HYPRE_IJVectorSetValues(v, n, idx, xs.ctypes.data_as(C.POINTER(C.c_double))) == 0 and L.HYPRE_IJVectorAssemble(v) == 0
    h = hd.Hypredrv("solver:\n  pcg:\n    relative_tol: 1.0e-10\npreconditioner: amg\n")
    h.set_matrix_csr(0, n - 1, A.indptr, A.indices, A.data)
    h.set_rhs_array(0, n - 1, np.ones(n))
    h.finish_system()
    L.HYPREDRV_LinearSystemSetReferenceSolution.argtypes = [C.c_void_p, C.c_void_p]
    hd.check(L.HYPREDRV_LinearSystemSetReferenceSolution(h.h, v))
    capfd.readouterr()
    r = h.solve()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    out = capfd.readouterr().out
    assert r["converged"]
    err = float(re.search(r"L2 norm of error: (\S+)", out).group(1))
    sol = float(re.search(r"L2 norm of solution: (\S+)", out).group(1))
    ref = float(re.search(r"L2 norm of ref. solution: (\S+)", out).group(1))
    assert ref == pytest.approx(np.linalg.norm(xs), rel=1e-6) and sol == pytest.approx(ref, rel=1e-6)
    assert err < 1e-7 * ref
    hd.check(L.HYPREDRV_LinearSystemSetReferenceSolution(h.h, None))
    h.close()
    L.HYPRE_IJVectorDestroy(v)


@pytest.mark.parametrize("kind", ["rhs_l2", "dofmap_custom"])
def test_scaling_is_reapplied_when_setup_is_skipped(hd, orc, kind):
    """src/HYPREDRV.c:3161-3176: LinearSolverApply transforms the system itself when the preceding call left it unscaled
    (second Apply after one Setup, or a reused preconditioner).  Weights that are powers of two make every transform exact,
    so both applies must reproduce the oracle's solve of the scaled system and leave the caller's system untouched."""
    A = _lap_coo(16 if kind == "rhs_l2" else 12)
    n = A.shape[0]
    values = [2.0, 0.5, 4.0]
    b = np.ones(n)
    if kind == "dofmap_custom":
        b = b * 3.0
    its, xref, A_after, b_after = _scaled_oracle(orc, A, b, np.arange(n) % 3, kind, values, "pcg")
    assert (A_after != A).nnz == 0 and np.array_equal(b_after, b)  # exact round trip: the premise of this test
    extra = "" if kind == "rhs_l2" else "    custom_values: [2.0, 0.5, 4.0]\n"
    h = hd.Hypredrv(f"solver:\n  pcg:\n    max_iter: 100\n  scaling:\n    enabled: on\n    type: {kind}\n{extra}preconditioner: amg\n")
    h.set_matrix_csr(0, n - 1, A.indptr, A.indices, A.data)
    h.set_rhs_array(0, n - 1, b)
    h.finish_system()
    if kind != "rhs_l2":
        hd.check(hd.lib().HYPREDRV_LinearSystemSetInterleavedDofmap(h.h, n // 3, 3))
    h.create_and_setup()
    for _ in range(3):
        r = h.apply()
        x = h.solution()
        assert r["converged"] and r["iters"] == its
        assert np.linalg.norm(x - xref) / np.linalg.norm(xref) < 1e-12
    h.destroy_solver()
    h.close()


def test_library_state_is_process_global_across_threads(hd, orc):
    """The reference's contract (include/HYPREDRV.h:66-70) is one thread AT A TIME, not one thread: a caller may initialise and
    build on one thread, set up and solve on another and release on a third (a finalizer thread).  Every piece of library state --
    HYPREDRV_Initialize's flag, the context and its stream, the allocator, the solver registry, the error state -- must be the same
    on all of them."""
    import threading
    Ao, b = orc.lap7(12, 12, 12)
    ref = orc.pcg(Ao, b, orc.Amg(Ao, orc.amg_params(True)))
    h = hd.Hypredrv()  # Initialize + Create on the main thread
    h.presets("pcg", "poisson")
    h.set_laplacian7((12, 12, 12))
    box = {}

    def worker():
        try:
            box["r"] = h.solve()          # LinearSolverCreate / Setup / Apply on another thread
            box["x"] = h.solution().copy()
        except Exception as e:  # noqa: BLE001
            box["err"] = e

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert "err" not in box, box.get("err")
    assert box["r"]["converged"] and box["r"]["iters"] == ref["iters"]
    assert np.linalg.norm(box["x"] - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-10
    r2 = h.solve()  # and again on the main thread: same objects, same stream
    assert r2["iters"] == ref["iters"]
    import hypredrive_amd as hh
    before = hh.memory_stats()[0]
    t2 = threading.Thread(target=h.close)  # released on a third thread: the blocks go back to the allocator they came from
    t2.start()
    t2.join()
    assert hh.memory_stats()[0] < before


@pytest.mark.parametrize("name", ["test_init_guess", "test_setmatrix_from_csr"])
def test_reference_unit_tests_unmodified(name):
    """The reference's OWN unit tests of this path -- /root/reference/tests/test_init_guess.c:170-270 (initial-guess modes `previous` /
    `ones`, asserted through GMRES iteration counts) and tests/test_setmatrix_from_csr.c:168-199,397-417 (CSR ingestion: known-answer
    solves, ownership transitions, empty and single rows, every negative case with its error bits) -- compiled UNMODIFIED against
    include/ + libhypredrv_amd.so (oracle/Makefile target ref_tests) and run as they are: every assertion of theirs is an assertion
    on this boundary.  Exit code 0 = all of them held."""
    exe = os.path.join(ROOT, "oracle", "_ref", name + "_ref")
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/{name}_ref not built (needs /root/reference + MPICH at build time)")
    r = subprocess.run([exe], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])


def _bench_two_ranks(extra_env, timeout=300):
    env = dict(os.environ, HDA_TRANSPORT="staged", HDA_BENCH_SERIAL_FIRST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "40", "--steps", "2", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    return r, lines


def test_bench_measures_serial_exchanges_before_overlapped_ones():
    """bench.py on N > 1 ranks over an asynchronous transport (forced here on the staged one: two ranks share the GPU) runs the W + K
    solves with every exchange ahead of its product first (hda_set_overlap(0)) and reports them as `serial_exchange` beside the headline."""
    r, lines = _bench_two_ranks({})
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["converged"] and "overlap_error" not in d
    se = d["serial_exchange"]
    assert se["converged"] and se["iters"] == d["iters"] and se["ms_per_step"] > 0 and abs(se["final_rel"] - d["final_rel"]) < 1e-9


def test_bench_reports_the_serial_measurement_when_the_overlapped_phase_hangs():
    """...and if the overlapped phase never returns (test hook), the guard writes the line from the serial measurement, labelled, and
    every rank leaves with a NON-ZERO status (round-4 ADVICE: the designed path failed -- "a result line, and a failure", as for the staged
    fallback): a first contact with RCCL still ends with a measurement of the job, and nobody reads it as a success of the overlap."""
    r, lines = _bench_two_ranks({"HDA_BENCH_TEST_OVERLAP_HANG": "1", "HDA_BENCH_OVERLAP_TIMEOUT": "4"})
    assert r.returncode != 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = lines[0]
    assert "did not finish" in d["overlap_error"] and d["value"] == d["serial_exchange"]["value"] and d["ms_per_step"] == d["serial_exchange"]["ms_per_step"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["converged"] and d["scaling"] == "weak"


@pytest.mark.parametrize("what", [0, 1])
def test_thread_rank_seam_selftests(what):
    """include/hypredrv_amd_testranks.h hda_testranks_selftest, in a child process (what = 1 asks the driver for nearly all of HBM).
    0: two thread ranks in DIFFERENT collectives get an error that names the disagreement -- the full-size config-3 run once crashed
       there, reading a stale pointer of a rank that had left the setup with an out-of-memory error.
    1: an allocation larger than the driver's free memory is served by returning ANOTHER rank thread's cached blocks -- the cause of
       that out-of-memory error: eight ranks' caches (up to twice each rank's peak) filled the device."""
    code = ("import sys; from hypredrive_amd import _lib; ok, msg = _lib.testranks_selftest(%d, 8.0); print(msg); sys.exit(0 if ok else 1)" % what)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    if what == 0:
        assert "different collectives" in r.stdout
    else:
        assert "served" in r.stdout and "cached 8." in r.stdout


def test_bench_side_configs_on_small_sizes():
    """BASELINE.json configs 4 and 5 as bench.py carries them (tools/side_configs.py, stand-in data): GMRES + MGR with examples/ex3.yml's
    block on the three-field system, GMRES + BoomerAMG with the ILU(0) smoother on the anisotropic reservoir operator -- through
    HYPREDRV_LinearSolverSetup / Apply, with the oracle's iteration count on a size it finishes in seconds."""
    import importlib.util
    import hypredrive_amd as hh
    spec = importlib.util.spec_from_file_location("side_configs", os.path.join(ROOT, "tools", "side_configs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m = mod.gmres_mgr(hh, cells=64, steps=2, warmup=1, oracle_cells=16)
    assert m["converged"] and m["iters_match"] and m["rows"] == 3 * 64 * 64 and m["final_rel"] < 1e-6
    assert m["dominant_kernel"]["launches"] >= m["iters"] and m["dominant_kernel"]["avg_ms"] > 0
    a = mod.gmres_amg_ilu0(hh, n=32, steps=2, warmup=1, oracle_n=14)
    assert a["converged"] and a["iters_match"] and a["rows"] == 32 ** 3 and a["final_rel"] < 1e-6
    assert a["dominant_kernel"]["launches"] > 0 and a["level0_residual"]["launches"] > 0
    assert "STAND-IN DATA" in m["what"] and "STAND-IN DATA" in a["what"]
