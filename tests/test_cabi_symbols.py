"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol
the headers declare, and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hda_[a-z0-9_]+|HYPREDRV_[A-Za-z0-9_]+|HYPRE_[A-Za-z0-9]+)\s*\(", txt)))


def test_kernel_abi_exports_every_declared_symbol():
    import hypredrive_amd as h
    L = h.load()
    names = [n for n in _declared("hypredrv_amd.h") if n.startswith("hda_")]
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hypredrv_amd.h but not exported"
    assert sorted(names) == sorted(h._lib.SYMBOLS)


def test_test_seam_is_a_library_of_its_own():
    """The test transport "ranks as threads of one process" is not in the product library: libhypredrv_amd_testranks.so exports what
    include/hypredrv_amd_testranks.h declares, libhypredrv_amd.so exports none of it and does not depend on it."""
    import subprocess
    import hypredrive_amd as h
    L, T = h.load(), h._lib.load_testranks()
    names = [n for n in _declared("hypredrv_amd_testranks.h") if n.startswith("hda_")]
    assert sorted(names) == sorted(h._lib.TESTRANKS_SYMBOLS) and len(names) == 6
    for n in names:
        assert hasattr(T, n), f"{n} declared in include/hypredrv_amd_testranks.h but not exported by the test library"
        assert not hasattr(L, n), f"{n} (test seam) is exported by the product library"
    lib = os.path.join(ROOT, "hypredrive_amd", "lib", "libhypredrv_amd.so")
    needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "testranks" not in needed
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    assert "ThreadComm" not in syms and "thread_world" not in syms


def test_param_struct_layout_matches_oracle(orc):
    """hda_amg_params starts with the oracle's orc_amg_params (same fields, order and GPU defaults of
    src/internal/amg.c:120-238); the complex-smoother fields follow (the oracle takes them through
    orc_amg_set_ilu_smoother)."""
    import hypredrive_amd as h
    hp = h.AmgParams.default()
    op = orc.amg_params(True)
    every = [n for (n, _t) in orc.AmgParams._fields_]
    shared = every[:every.index("cheby_fraction") + 1]
    assert [n for (n, _t) in h.AmgParams._fields_][:len(shared)] == shared
    for name in shared:
        assert getattr(hp, name) == getattr(op, name), name
        assert getattr(h.AmgParams, name).offset == getattr(orc.AmgParams, name).offset, name
    # the aggressive-coarsening fields close both structs (AMGagg_args defaults of src/internal/amg.c:164-171: 0 levels, 1 path, multipass)
    # ... then the row blocks (the reference at np = V); the product's struct ends with its own size, which hda_amg_create checks
    tail = ["agg_num_levels", "agg_num_paths", "agg_interp_type", "agg_pmax", "agg_trunc_factor", "blocks", "block_part"]
    # (the oracle alone ends with pmis_rng: hypre's own tie-break stream, an oracle-side experiment -- tests/test_oracle_pins.py)
    assert every[len(shared):] == tail + ["pmis_rng"] and tail == [n for (n, _t) in h.AmgParams._fields_][-8:-1]
    assert op.pmis_rng == 0
    assert h.AmgParams._fields_[-1][0] == "struct_size" and hp.struct_size == C.sizeof(h.AmgParams)
    for name in tail[:-1]:
        assert getattr(hp, name) == getattr(op, name), name
    assert hp.blocks == 1 and not hp.block_part and not op.block_part
    assert (hp.agg_num_levels, hp.agg_num_paths, hp.agg_interp_type, hp.agg_pmax, hp.agg_trunc_factor) == (0, 1, 4, 0, 0.0)
    # ILU_args defaults of src/internal/ilu.c:21-23 and smoother off (amg.c:237)
    assert (hp.smooth_num_levels, hp.smooth_num_sweeps, hp.ilu_tri_solve, hp.ilu_lower_it, hp.ilu_upper_it) == (0, 1, 1, 5, 5)
    assert (hp.coarsen_type, hp.relax_down, hp.relax_up, hp.relax_coarse) == (8, 18, 18, 9)
    assert (hp.pmax, hp.strong_th, hp.max_row_sum, hp.max_coarse_size, hp.max_levels) == (4, 0.25, 0.9, 64, 25)
    kp = h.KrylovParams.default(False)
    assert (kp.max_iter, kp.rtol, kp.two_norm) == (100, 1e-6, 1)
    kg = h.KrylovParams.default(True)
    assert (kg.max_iter, kg.krylov_dim) == (300, 30)


def test_no_silent_cpu_fallback():
    import hypredrive_amd as h
    if h.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(h.LibraryError, match="no HIP device"):
        h.lap7(4, 4, 4)
    with pytest.raises(h.LibraryError, match="no HIP device"):
        h.Csr.from_arrays(1, 1, [0, 1], [0], [1.0])


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under hypredrive_amd/ or include/ may
    reference it."""
    bad = []
    for base in ("hypredrive_amd", "include"):
        for dp, _dn, fn in os.walk(os.path.join(ROOT, base)):
            for f in fn:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c", "Makefile")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_ffi|amg_oracle|liboracle|from oracle|import oracle", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_thread_transport_reports_ranks_in_different_collectives():
    """Host-only part of the thread-rank seam (no GPU call): two thread ranks that enter DIFFERENT collectives both come back with an
    error naming the disagreement instead of reading each other's stale pointers (include/hypredrv_amd_testranks.h
    hda_testranks_selftest, what = 0)."""
    import subprocess
    import sys
    code = "import sys; from hypredrive_amd import _lib; ok, msg = _lib.testranks_selftest(%d); print(msg); sys.exit(0 if ok else 1)"
    r = subprocess.run([sys.executable, "-c", code % 0], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("different collectives") == 2
    # what = 2: the same collective, but one rank expects 16 bytes where 8 are sent
    r = subprocess.run([sys.executable, "-c", code % 2], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sends 8 bytes" in r.stdout and "which expects 16" in r.stdout
