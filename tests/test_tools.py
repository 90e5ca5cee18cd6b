"""CPU tests of the measurement tooling that bench.py runs on the GPU box (tools/pmc_traffic.py: the reduction of two rocprofv3 counter
passes to HBM bytes per launch behind roofline.traffic)."""
import csv
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write(d, counter, rows):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "run_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, (k, v) in enumerate(rows):
            w.writerow([i + 1, k, counter, v])


def test_pmc_traffic_calibrates_on_cg_dir_and_reports_the_quoted_launches(tmp_path):
    """FETCH_SIZE on gfx950 counts half of a streamed 8-byte read: the factor is re-derived from k_cg_dir (reads z, p; writes p on 256^3)
    and applied to every kernel; the dominant kernel's traffic is the median of the launches within 5 % of the largest FETCH_SIZE."""
    mod = _load()
    n = 256 ** 3
    dir_k = "void hda::k_cg_dir<false>(int, double*, int, int, double const*, double*, double const*)"
    win = "void hda::k_spmv_win<2, false, false, false, false>(int, int const*)"
    row = "void hda::k_spmv_rowclass<0, true, false>(int, unsigned char const*)"
    kernels = [dir_k, win, win, win, row, dir_k]
    fetch = {dir_k: 8.0 * n / 1024, win: 950000.0, row: 77000.0}   # k_cg_dir: exactly half of its 2 x 8 B per lane
    write = {dir_k: 8.0 * n / 1024, win: 44000.0, row: 131000.0}
    frows = [(k, fetch[k]) for k in kernels]
    frows[2] = (win, 300000.0)                                       # a smaller level: outside the top cluster
    _write(tmp_path / "f", "FETCH_SIZE", frows)
    _write(tmp_path / "w", "WRITE_SIZE", [(k, write[k]) for k in kernels])
    out = mod.compute(str(tmp_path / "f"), str(tmp_path / "w"), "unit", str(tmp_path / "summary.csv"))
    assert out["calibration"]["fetch_factor"] == pytest.approx(2.0) and out["calibration"]["write_factor"] == pytest.approx(1.0)
    assert out["k_spmv_stream_jacobi_level1_bytes_per_launch"] == pytest.approx(2.0 * 950000.0 * 1024 + 44000.0 * 1024)
    assert out["k_spmv_level0_bytes_per_launch"] == pytest.approx(2.0 * 77000.0 * 1024 + 131000.0 * 1024)
    assert out["round"] == "unit" and os.path.exists(tmp_path / "summary.csv")


def test_pmc_traffic_sums_the_solve_phase_between_the_markers(tmp_path):
    """solve_phase_traffic_bytes = calibrated FETCH + WRITE of EVERY dispatch between the two hda::k_marker launches bench.py puts around
    its timed solve; what runs before the first marker (setup) and after the last one (probe read-outs) does not count."""
    mod = _load()
    n = 256 ** 3
    dir_k = "void hda::k_cg_dir<false>(int, double*, int, int, double const*, double*, double const*)"
    mark = "hda::k_marker(int, int*)"
    setup = "void hda::k_spgemm_esc<8, 256>(int)"
    a, b = "void hda::k_spmv_win<2, false, false, false, false>(int, int const*)", "void hda::k_cg_update<true, true>(int)"
    seq = [setup, setup, mark, a, dir_k, b, a, mark, setup]
    fetch = {setup: 5e6, mark: 0.0, a: 900000.0, b: 200000.0, dir_k: 8.0 * n / 1024}
    write = {setup: 3e6, mark: 0.0, a: 40000.0, b: 260000.0, dir_k: 8.0 * n / 1024}
    _write(tmp_path / "f", "FETCH_SIZE", [(k, fetch[k]) for k in seq])
    _write(tmp_path / "w", "WRITE_SIZE", [(k, write[k]) for k in seq])
    out = mod.compute(str(tmp_path / "f"), str(tmp_path / "w"), "unit")
    inside = [a, dir_k, b, a]
    want = sum(2.0 * fetch[k] + write[k] for k in inside) * 1024
    assert out["solve_phase_traffic_bytes"] == pytest.approx(want)
    assert out["solve_phase"]["dispatches"] == 4
    assert out["solve_phase"]["top_kernels"][0]["kernel"].endswith("k_spmv_win<2, false, false, false, false>")
    # without markers the whole-solve figure is absent rather than guessed
    _write(tmp_path / "f2", "FETCH_SIZE", [(k, fetch[k]) for k in seq if k != mark])
    _write(tmp_path / "w2", "WRITE_SIZE", [(k, write[k]) for k in seq if k != mark])
    assert "solve_phase_traffic_bytes" not in mod.compute(str(tmp_path / "f2"), str(tmp_path / "w2"), "unit")


def test_bench_skips_its_counter_passes_when_it_is_being_profiled(monkeypatch):
    """bench.py's own rocprofv3 children must not start under a profiler (a profiler inside a profiler): the committed profile is quoted."""
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.setattr("shutil.which", lambda name: "/usr/bin/true")

    class A:
        n = 256
    t, why = bench.measure_traffic(A())
    assert t is None and "profiled" in why


def test_series_b_rank_block_numbering_is_the_generators():
    """tools/series_b.py --rank-grid p hands the library the n^3 Laplacian in the numbering the reference's generator produces at
    np = p^3 (`-P p p p`, examples/src/C_laplacian/laplacian.c:504-520 grid2idx; rhs :898-905): the oracle's restatement of that
    generator (pinned by laplacian.txt) must give the same matrix and right-hand side, entry for entry."""
    import importlib.util
    import numpy as np
    import scipy.sparse as sp
    from oracle import oracle_ffi as orc
    spec = importlib.util.spec_from_file_location("series_b", os.path.join(ROOT, "tools", "series_b.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for n, p in ((8, 2), (12, 3), (8, 1)):
        ip, cj, v, b = mod.lap7_rank_blocks(n, p)
        A, bo = orc.lap7(n, n, n, P=(p, p, p))
        M = sp.csr_matrix((v, cj, ip), shape=(n ** 3, n ** 3))
        Mo = sp.csr_matrix((np.array(A.val), np.array(A.col), np.array(A.rowptr)), shape=(n ** 3, n ** 3))
        assert M.nnz == Mo.nnz and abs(M - Mo).max() == 0.0
        assert np.array_equal(b, bo)
        # block q of n^3 / p^3 consecutive rows is a sub-cube: its rows couple to at most 6 (n/p)^2 rows outside
        nl3 = (n // p) ** 3
        q = np.repeat(np.arange(n ** 3) // nl3, np.diff(ip))
        assert (q != cj // nl3).sum() == 6 * (n // p) ** 2 * p ** 3 - 6 * n * n
