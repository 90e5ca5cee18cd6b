"""ctypes binding of the kernel-level C ABI (include/hypredrv_amd.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "lib", "libhypredrv_amd.so")
_L = None


class LibraryError(RuntimeError):
    pass


class AmgParams(C.Structure):
    _fields_ = [("coarsen_type", C.c_int), ("interp_type", C.c_int), ("pmax", C.c_int),
                ("trunc_factor", C.c_double), ("strong_th", C.c_double),
                ("max_row_sum", C.c_double), ("max_coarse_size", C.c_int),
                ("min_coarse_size", C.c_int), ("max_levels", C.c_int),
                ("relax_down", C.c_int), ("relax_up", C.c_int), ("relax_coarse", C.c_int),
                ("sweeps_down", C.c_int), ("sweeps_up", C.c_int), ("sweeps_coarse", C.c_int),
                ("relax_weight", C.c_double), ("outer_weight", C.c_double),
                ("seed", C.c_uint64), ("num_functions", C.c_int),
                ("cheby_order", C.c_int), ("cheby_eig_est", C.c_int), ("cheby_variant", C.c_int), ("cheby_scale", C.c_int),
                ("cheby_fraction", C.c_double),
                ("smooth_num_levels", C.c_int), ("smooth_num_sweeps", C.c_int),
                ("ilu_tri_solve", C.c_int), ("ilu_lower_it", C.c_int), ("ilu_upper_it", C.c_int),
                ("agg_num_levels", C.c_int), ("agg_num_paths", C.c_int), ("agg_interp_type", C.c_int),
                ("agg_pmax", C.c_int), ("agg_trunc_factor", C.c_double),
                ("blocks", C.c_int), ("block_part", C.POINTER(C.c_int64)), ("struct_size", C.c_int)]

    @staticmethod
    def default(**kw):
        p = AmgParams()
        load().hda_amg_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise KeyError(k)
            if k == "block_part":
                if v is None:
                    continue
                keep = np.ascontiguousarray(v, dtype=np.int64)
                p._block_part_keep = keep  # (borrowed until the setup has read it)
                p.block_part = keep.ctypes.data_as(C.POINTER(C.c_int64))
                continue
            setattr(p, k, v)
        return p


class KrylovParams(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("rtol", C.c_double), ("atol", C.c_double),
                ("two_norm", C.c_int), ("krylov_dim", C.c_int)]

    @staticmethod
    def default(gmres=False, **kw):
        p = KrylovParams()
        load().hda_krylov_default_params(C.byref(p), 1 if gmres else 0)
        for k, v in kw.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
        return p


class MgrLevelParams(C.Structure):
    _fields_ = [("n_f_labels", C.c_int), ("f_labels", C.POINTER(C.c_int)), ("interp_type", C.c_int), ("restrict_type", C.c_int),
                ("frelax_type", C.c_int), ("frelax_sweeps", C.c_int), ("grelax_type", C.c_int), ("grelax_sweeps", C.c_int),
                ("frelax_amg", C.POINTER(AmgParams)),
                ("ilu_tri_solve", C.c_int), ("ilu_lower_it", C.c_int), ("ilu_upper_it", C.c_int),
                ("coarse_ilu_max_iter", C.c_int), ("coarse_ilu_tri_solve", C.c_int), ("coarse_ilu_lower_it", C.c_int), ("coarse_ilu_upper_it", C.c_int),
                ("frelax_krylov", C.c_int), ("frelax_krylov_precond", C.c_int), ("frelax_kp", KrylovParams),
                ("coarse_krylov", C.c_int), ("coarse_krylov_precond", C.c_int), ("coarse_kp", KrylovParams),
                ("mgr_cycle", C.c_int), ("mgr_frelax_pos", C.c_int), ("mgr_gsmooth_pos", C.c_int), ("grelax_blocks", C.c_int)]


# every symbol include/hypredrv_amd.h declares (checked by tests/test_cabi_symbols.py)
SYMBOLS = [
    "hda_last_error", "hda_device_count", "hda_device_name", "hda_device_pci_bus_id", "hda_device_sync",
    "hda_amg_default_params", "hda_krylov_default_params", "hda_csr_create", "hda_csr_destroy",
    "hda_csr_dims", "hda_csr_download", "hda_lap7_create", "hda_spmv", "hda_relax", "hda_dot",
    "hda_l1_norms", "hda_strength", "hda_pmis", "hda_interp_extpi", "hda_interp_direct", "hda_rap", "hda_transpose",
    "hda_spgemm", "hda_amg_create", "hda_amg_destroy", "hda_amg_num_levels",
    "hda_last_precond_calls", "hda_amg_create_dof", "hda_format_bytes", "hda_probe_spmv", "hda_probe_read", "hda_amg_level_matrix", "hda_amg_level_cf", "hda_amg_complexities", "hda_amg_vcycle_bytes",
    "hda_amg_vcycle", "hda_pcg", "hda_gmres", "hda_time_kernel", "hda_solve_device",
    "hda_pcg_iteration_bytes", "hda_memory_stats", "hda_memory_cached", "hda_memory_driver_stats", "hda_memory_trim", "hda_comm_selftest", "hda_check_row_total", "hda_ilu_create", "hda_ilu_create_blocks", "hda_ilu_blocks", "hda_ilu_factors", "hda_fgmres", "hda_bicgstab", "hda_mgr_create", "hda_mgr_matrix",
    "hda_probe_add", "hda_probe_read_id", "hda_borrow_hypredrv", "hda_comm_stats", "hda_comm_name", "hda_comm_size", "hda_halo_plan_host",
    "hda_amd_partitioned_levels", "hda_amd_hierarchy_levels",
    "hda_second_strength", "hda_coarsen_second_pass", "hda_interp_multipass", "hda_truncate_rows",
    "hda_interp_mm_extpi", "hda_interp_standard", "hda_set_overlap", "hda_marker", "hda_relax_blocks", "hda_l1_norms_blocks", "hda_hmis_blocks", "hda_amg_blocks", "hda_amg_level_blocks",
]


TESTRANKS_SYMBOLS = ["hda_thread_ranks_lap7", "hda_thread_world_create", "hda_thread_world_join", "hda_thread_world_leave",
                     "hda_thread_world_destroy", "hda_testranks_selftest"]
_TESTRANKS_PATH = os.path.join(os.path.dirname(_LIBPATH), "libhypredrv_amd_testranks.so")
_T = None


def load_testranks():
    """The test seam "ranks as threads of one process" (include/hypredrv_amd_testranks.h): its own small library on top of the
    product library, which is loaded first so that both share one instance of it."""
    global _T
    if _T is None:
        load()
        if not os.path.exists(_TESTRANKS_PATH):
            raise LibraryError(f"{_TESTRANKS_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _T = C.CDLL(_TESTRANKS_PATH)
    return _T


def testranks_selftest(what, cache_gb=8.0):
    """include/hypredrv_amd_testranks.h hda_testranks_selftest: (ok, the ranks' messages)"""
    T = load_testranks()
    T.hda_testranks_selftest.argtypes = [C.c_int, C.c_double, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(4096)
    rc = T.hda_testranks_selftest(int(what), float(cache_gb), buf, 4096)
    return rc == 0, buf.value.decode(errors="replace")


def load():
    """Load libhypredrv_amd.so; raises (never falls back) when it is missing."""
    global _L
    if _L is not None:
        return _L
    if not os.path.exists(_LIBPATH):
        raise LibraryError(
            f"{_LIBPATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MI355X solve path has no CPU fallback)")
    L = C.CDLL(_LIBPATH)
    P = C.POINTER
    dp, ip, vp = P(C.c_double), P(C.c_int), C.c_void_p
    L.hda_last_error.restype = C.c_char_p
    L.hda_device_name.argtypes = [C.c_char_p, C.c_int]
    L.hda_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.hda_amg_default_params.argtypes = [P(AmgParams)]
    L.hda_amg_default_params.restype = None
    L.hda_krylov_default_params.argtypes = [P(KrylovParams), C.c_int]
    L.hda_krylov_default_params.restype = None
    L.hda_csr_create.argtypes = [C.c_int, C.c_int, P(C.c_int64), P(C.c_int64), dp, P(vp)]
    L.hda_csr_destroy.argtypes = [vp]
    L.hda_csr_dims.argtypes = [vp, ip, ip, ip]
    L.hda_csr_download.argtypes = [vp, ip, ip, dp]
    L.hda_lap7_create.argtypes = [ip, ip, ip, dp, P(vp), dp]
    L.hda_spmv.argtypes = [vp, C.c_double, dp, C.c_double, dp]
    L.hda_relax.argtypes = [vp, C.c_int, C.c_double, C.c_int, dp, dp]
    L.hda_dot.argtypes = [C.c_int, dp, dp, dp]
    L.hda_l1_norms.argtypes = [vp, C.c_int, dp]
    L.hda_relax_blocks.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.c_int, P(C.c_int64), dp, dp]
    L.hda_l1_norms_blocks.argtypes = [vp, C.c_int, C.c_int, P(C.c_int64), dp]
    L.hda_hmis_blocks.argtypes = [vp, P(C.c_ubyte), C.c_int, P(C.c_int64), C.c_uint64, C.c_int, ip]
    L.hda_amg_blocks.argtypes = [vp]
    L.hda_marker.argtypes = [C.c_int]
    L.hda_amg_level_blocks.argtypes = [vp, C.c_int, P(C.c_int64)]
    L.hda_strength.argtypes = [vp, C.c_double, C.c_double, P(C.c_ubyte)]
    L.hda_pmis.argtypes = [vp, P(C.c_ubyte), C.c_uint64, C.c_int, C.c_int64, ip]
    L.hda_interp_extpi.argtypes = [vp, P(C.c_ubyte), ip, C.c_int, C.c_double, P(vp)]
    L.hda_interp_direct.argtypes = [vp, P(C.c_ubyte), ip, C.c_int, C.c_double, P(vp)]
    L.hda_interp_mm_extpi.argtypes = [vp, P(C.c_ubyte), ip, C.c_int, C.c_double, P(vp)]
    L.hda_interp_standard.argtypes = [vp, P(C.c_ubyte), ip, C.c_int, C.c_double, P(vp)]
    L.hda_rap.argtypes = [vp, vp, P(vp)]
    L.hda_second_strength.argtypes = [vp, P(C.c_ubyte), ip, C.c_int, P(vp)]
    L.hda_coarsen_second_pass.argtypes = [vp, P(C.c_ubyte), C.c_int, C.c_uint64, C.c_int, ip]
    L.hda_interp_multipass.argtypes = [vp, P(C.c_ubyte), ip, P(vp)]
    L.hda_truncate_rows.argtypes = [vp, C.c_int, C.c_double]
    L.hda_transpose.argtypes = [vp, P(vp)]
    L.hda_spgemm.argtypes = [vp, vp, P(vp)]
    L.hda_amg_create.argtypes = [P(AmgParams), vp, P(vp)]
    L.hda_amg_destroy.argtypes = [vp]
    L.hda_mgr_create.argtypes = [vp, ip, C.c_int, P(MgrLevelParams), P(AmgParams), C.c_int, P(vp)]
    L.hda_mgr_matrix.argtypes = [vp, C.c_int, C.c_int, P(vp)]
    L.hda_ilu_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, P(vp)]
    L.hda_ilu_factors.argtypes = [vp, C.c_int, P(vp)]
    L.hda_ilu_create_blocks.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_int64), P(vp)]
    L.hda_ilu_blocks.argtypes = [vp, C.c_int]
    L.hda_set_overlap.argtypes = [C.c_int]
    L.hda_set_overlap.restype = None
    L.hda_amg_num_levels.argtypes = [vp]
    L.hda_amg_level_matrix.argtypes = [vp, C.c_int, C.c_int, P(vp)]
    L.hda_amg_level_cf.argtypes = [vp, C.c_int, ip]
    L.hda_amg_complexities.argtypes = [vp, dp, dp]
    L.hda_amg_vcycle_bytes.argtypes = [vp]
    L.hda_amg_vcycle_bytes.restype = C.c_double
    L.hda_amg_vcycle.argtypes = [vp, dp, dp]
    for f in (L.hda_pcg, L.hda_gmres, L.hda_fgmres, L.hda_bicgstab):
        f.argtypes = [vp, vp, P(KrylovParams), dp, dp, dp, ip, ip, dp]
    L.hda_time_kernel.argtypes = [C.c_int, vp, vp, C.c_int, dp, dp]
    L.hda_solve_device.argtypes = [vp, vp, P(KrylovParams), C.c_int, dp, C.c_int, dp, ip, dp, dp, dp, dp]
    L.hda_pcg_iteration_bytes.argtypes = [vp]
    L.hda_pcg_iteration_bytes.restype = C.c_double
    L.hda_memory_stats.argtypes = [dp, dp]
    L.hda_memory_cached.restype = C.c_double
    L.hda_memory_driver_stats.argtypes = [C.POINTER(C.c_double), C.c_int]
    L.hda_check_row_total.argtypes = [C.c_longlong, C.c_int]
    L.hda_format_bytes.argtypes = [vp, vp, dp, dp, dp, ip]
    L.hda_probe_spmv.argtypes = [vp, C.c_int]
    L.hda_probe_read.argtypes = [dp, ip]
    L.hda_probe_add.argtypes = [vp, C.c_int, ip]
    L.hda_probe_read_id.argtypes = [C.c_int, dp, ip]
    L.hda_borrow_hypredrv.argtypes = [vp, P(vp), P(vp)]
    L.hda_comm_stats.argtypes = [dp, C.c_int]
    L.hda_comm_name.restype = C.c_char_p
    _L = L
    return L


def _check(rc):
    if rc != 0:
        raise LibraryError(f"libhypredrv_amd error {rc}: {load().hda_last_error().decode()}")


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def device_count():
    return load().hda_device_count()


def device_name():
    buf = C.create_string_buffer(256)
    _check(load().hda_device_name(buf, 256))
    return buf.value.decode()


class Csr:
    """CSR block resident in HBM."""

    def __init__(self, handle, owned=True, keep=None):
        self.h = handle
        self.owned = owned
        self._keep = keep
        self.rhs = None

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            load().hda_csr_destroy(self.h)
            self.h = None

    @staticmethod
    def from_arrays(nrows, ncols, rowptr, cols, vals):
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        cj = np.ascontiguousarray(cols, dtype=np.int64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        out = C.c_void_p()
        _check(load().hda_csr_create(nrows, ncols, rp.ctypes.data_as(C.POINTER(C.c_int64)),
                                     cj.ctypes.data_as(C.POINTER(C.c_int64)), _dp(v), C.byref(out)))
        return Csr(out)

    @staticmethod
    def from_scipy(m):
        m = m.tocsr()
        return Csr.from_arrays(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)

    @property
    def dims(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(load().hda_csr_dims(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    nrows = property(lambda s: s.dims[0])
    ncols = property(lambda s: s.dims[1])
    nnz = property(lambda s: s.dims[2])

    def download(self):
        n, m, nnz = self.dims
        rp = np.zeros(n + 1, dtype=np.int32)
        cj = np.zeros(max(nnz, 1), dtype=np.int32)
        v = np.zeros(max(nnz, 1), dtype=np.float64)
        _check(load().hda_csr_download(self.h, _ip(rp), _ip(cj), _dp(v)))
        return rp, cj[:nnz], v[:nnz]

    def to_scipy(self):
        import scipy.sparse as sp
        rp, cj, v = self.download()
        n, m, _ = self.dims
        return sp.csr_matrix((v, cj, rp), shape=(n, m))

    def spmv(self, x, alpha=1.0, beta=0.0, y=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.nrows) if y is None else np.ascontiguousarray(y, dtype=np.float64).copy()
        _check(load().hda_spmv(self.h, alpha, _dp(x), beta, _dp(y)))
        return y

    def relax(self, b, x, relax_type=18, weight=1.0, sweeps=1):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        _check(load().hda_relax(self.h, relax_type, weight, sweeps, _dp(b), _dp(x)))
        return x

    def l1_norms(self, option=1):
        out = np.zeros(self.nrows)
        _check(load().hda_l1_norms(self.h, option, _dp(out)))
        return out

    # the row-block forms (part: V + 1 row starts) = the reference at np = V
    def relax_blocks(self, b, x, part, relax_type=13, weight=1.0, sweeps=1):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        pt = np.ascontiguousarray(part, dtype=np.int64)
        _check(load().hda_relax_blocks(self.h, relax_type, weight, sweeps, len(pt) - 1, pt.ctypes.data_as(C.POINTER(C.c_int64)), _dp(b), _dp(x)))
        return x

    def l1_norms_blocks(self, option, part):
        out = np.zeros(max(self.nrows, 1))
        pt = np.ascontiguousarray(part, dtype=np.int64)
        _check(load().hda_l1_norms_blocks(self.h, option, len(pt) - 1, pt.ctypes.data_as(C.POINTER(C.c_int64)), _dp(out)))
        return out[:self.nrows]

    def hmis_blocks(self, smask, part, seed=2747, level=0):
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cf = np.zeros(max(self.nrows, 1), dtype=np.int32)
        pt = np.ascontiguousarray(part, dtype=np.int64)
        _check(load().hda_hmis_blocks(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), len(pt) - 1, pt.ctypes.data_as(C.POINTER(C.c_int64)),
                                      seed, level, _ip(cf)))
        return cf[:self.nrows]

    def strength(self, theta=0.25, max_row_sum=0.9):
        sm = np.zeros(max(self.nnz, 1), dtype=np.uint8)
        _check(load().hda_strength(self.h, theta, max_row_sum, sm.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return sm[:self.nnz]

    def pmis(self, smask, seed=2747, level=0, row_offset=0):
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cf = np.zeros(max(self.nrows, 1), dtype=np.int32)
        _check(load().hda_pmis(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), seed, level, row_offset, _ip(cf)))
        return cf[:self.nrows]

    def interp_extpi(self, smask, cf, pmax=4, trunc_factor=0.0):
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cfa = np.ascontiguousarray(np.concatenate([cf, np.zeros(1, np.int32)]), dtype=np.int32)
        out = C.c_void_p()
        _check(load().hda_interp_extpi(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cfa), pmax,
                                       trunc_factor, C.byref(out)))
        return Csr(out)

    def interp_mm_extpi(self, smask, cf, pmax=4, trunc_factor=0.0):
        """interpolation type 17 (mm-ext+i): the matrix-matrix form of extended+i"""
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cfa = np.ascontiguousarray(np.concatenate([cf, np.zeros(1, np.int32)]), dtype=np.int32)
        out = C.c_void_p()
        _check(load().hda_interp_mm_extpi(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cfa), pmax, trunc_factor, C.byref(out)))
        return Csr(out)

    def interp_standard(self, smask, cf, pmax=4, trunc_factor=0.0):
        """interpolation type 8 (standard): strong F neighbours eliminated through their own rows, direct interpolation on the result"""
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cfa = np.ascontiguousarray(np.concatenate([cf, np.zeros(1, np.int32)]), dtype=np.int32)
        out = C.c_void_p()
        _check(load().hda_interp_standard(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cfa), pmax, trunc_factor, C.byref(out)))
        return Csr(out)

    def interp_direct(self, smask, cf, pmax=4, trunc_factor=0.0):
        """interpolation type 3 (direct_sep_weights)"""
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cfa = np.ascontiguousarray(np.concatenate([cf, np.zeros(1, np.int32)]), dtype=np.int32)
        out = C.c_void_p()
        _check(load().hda_interp_direct(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cfa), pmax,
                                        trunc_factor, C.byref(out)))
        return Csr(out)

    def second_strength(self, smask, cf, num_paths=1):
        """aggressive coarsening: strong connections of distance <= 2 among the C points of cf (values = number of paths)"""
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cfa = np.ascontiguousarray(np.concatenate([cf, np.zeros(1, np.int32)]), dtype=np.int32)
        out = C.c_void_p()
        _check(load().hda_second_strength(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cfa), num_paths, C.byref(out)))
        return Csr(out)

    def coarsen_second_pass(self, smask, cf, num_paths=1, seed=2747, level=0):
        """aggressive coarsening: the second PMIS pass; returns the updated C/F marker"""
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cfa = np.ascontiguousarray(np.concatenate([cf, np.zeros(1, np.int32)]), dtype=np.int32)
        _check(load().hda_coarsen_second_pass(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), num_paths, seed, level, _ip(cfa)))
        return cfa[:-1].copy()

    def interp_multipass(self, smask, cf):
        """aggressive coarsening: multipass interpolation (aggressive.prolongation_type 4)"""
        sm = np.ascontiguousarray(np.concatenate([smask, np.zeros(1, np.uint8)]), dtype=np.uint8)
        cfa = np.ascontiguousarray(np.concatenate([cf, np.zeros(1, np.int32)]), dtype=np.int32)
        out = C.c_void_p()
        _check(load().hda_interp_multipass(self.h, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), _ip(cfa), C.byref(out)))
        return Csr(out)

    def truncate_rows(self, pmax=0, trunc_factor=0.0):
        """hypre_BoomerAMGInterpTruncation on the finished rows of this interpolation matrix, in place"""
        _check(load().hda_truncate_rows(self.h, pmax, trunc_factor))
        return self

    def rap(self, P):
        out = C.c_void_p()
        _check(load().hda_rap(self.h, P.h, C.byref(out)))
        return Csr(out)

    def transpose(self):
        out = C.c_void_p()
        _check(load().hda_transpose(self.h, C.byref(out)))
        return Csr(out)

    def matmul(self, Y):
        out = C.c_void_p()
        _check(load().hda_spgemm(self.h, Y.h, C.byref(out)))
        return Csr(out)


def lap7(nx, ny, nz, c=(1.0, 1.0, 1.0), want_rhs=True):
    """7-pt Laplacian generated in HBM (examples/src/C_laplacian/laplacian.c:719-921)."""
    n = (C.c_int * 3)(nx, ny, nz)
    P = (C.c_int * 3)(1, 1, 1)
    pc = (C.c_int * 3)(0, 0, 0)
    cc = (C.c_double * 3)(*c)
    out = C.c_void_p()
    rhs = np.zeros(nx * ny * nz) if want_rhs else None
    _check(load().hda_lap7_create(n, P, pc, cc, C.byref(out), _dp(rhs) if want_rhs else None))
    A = Csr(out)
    A.rhs = rhs
    return A


class Amg:
    def __init__(self, A, params=None, dof=None):
        self.A = A
        self.params = params if params is not None else AmgParams.default()
        self.h = C.c_void_p()
        if dof is None:
            _check(load().hda_amg_create(C.byref(self.params), A.h, C.byref(self.h)))
        else:
            d = np.ascontiguousarray(dof, dtype=np.int32)
            _check(load().hda_amg_create_dof(C.byref(self.params), A.h, d.ctypes.data_as(C.POINTER(C.c_int)), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            load().hda_amg_destroy(self.h)
            self.h = None

    @property
    def num_levels(self):
        return load().hda_amg_num_levels(self.h)

    def level_matrix(self, level, which=0):
        out = C.c_void_p()
        _check(load().hda_amg_level_matrix(self.h, level, which, C.byref(out)))
        return Csr(out, owned=False, keep=self)

    def ilu_factors(self, level):
        """Factors of the complex smoother (amg.smoother.type ilu) of a level."""
        out = C.c_void_p()
        _check(load().hda_ilu_factors(self.h, level, C.byref(out)))
        return Csr(out, owned=False, keep=self)

    def level_cf(self, level):
        n = self.level_matrix(level, 0).nrows
        cf = np.zeros(n, dtype=np.int32)
        _check(load().hda_amg_level_cf(self.h, level, _ip(cf)))
        return cf

    @property
    def blocks(self):
        """row blocks the setup worked with (AmgParams.blocks resolved; 1 = none)"""
        return load().hda_amg_blocks(self.h)

    def level_blocks(self, level):
        out = np.zeros(max(self.blocks, 1) + 1, dtype=np.int64)
        _check(load().hda_amg_level_blocks(self.h, level, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    @property
    def complexities(self):
        g, o = C.c_double(), C.c_double()
        _check(load().hda_amg_complexities(self.h, C.byref(g), C.byref(o)))
        return g.value, o.value

    @property
    def vcycle_bytes(self):
        return load().hda_amg_vcycle_bytes(self.h)

    def vcycle(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        _check(load().hda_amg_vcycle(self.h, _dp(b), _dp(x)))
        return x


MGR_INTERP = {"injection": 0, "l1-jacobi": 1, "jacobi": 2}
MGR_RESTRICT = {"injection": 0, "jacobi": 2, "columped": 14}
MGR_FRELAX = {"jacobi": 7, "single": 7, "l1-jacobi": 18, "amg": 2, "ilu": 32}
MGR_GRELAX = {"none": -1, "h-fgs": 3, "h-bgs": 4, "h-ssor": 6, "l1-hfgs": 13, "l1-hbgs": 14, "l1-hsgs": 88, "ilu": 16}

MGR_KRYLOV = {"pcg": 1, "gmres": 2, "fgmres": 3, "bicgstab": 4}


def _mgr_nested(entry, lv):
    """nested Krylov components of a level dict: f_krylov / coarsest_krylov = dict(method=..., precond=True, **krylov params)"""
    for key, pre in (("f_krylov", "frelax"), ("coarsest_krylov", "coarse")):
        nk = lv.get(key)
        if not nk:
            continue
        kw = {k: v for k, v in nk.items() if k not in ("method", "precond")}
        kp = getattr(entry, pre + "_kp")
        for k, v in {**dict(max_iter=100, rtol=1e-6, atol=0.0, two_norm=1, krylov_dim=30), **kw}.items():
            setattr(kp, k, v)
        setattr(entry, pre + "_krylov", MGR_KRYLOV[nk.get("method", "gmres")])
        setattr(entry, pre + "_krylov_precond", 1 if nk.get("precond", True) else 0)
    cyc = lv.get("cycle")  # on the last level: "v(1,0)" (default), "v(0,1)", "v(1,1)", "w", "w(0,1)", "w(1,1)"
    if cyc:
        entry.mgr_cycle = 2 if cyc.startswith("w") else 1
        pos = {"": 1, "(1,0)": 1, "(0,1)": 2, "(1,1)": 3}[cyc[1:]]
        entry.mgr_frelax_pos = entry.mgr_gsmooth_pos = pos


class Mgr:
    """'preconditioner: mgr': multigrid reduction by dof labels; levels = list of dicts with the YAML keys of
    mgr.level.N (f_dofs, prolongation_type, restriction_type, f_relaxation, g_relaxation [, f_sweeps, g_sweeps]).
    Usable as amg= in pcg()/gmres()/fgmres()/bicgstab()."""

    def __init__(self, A, labels, levels, coarse_params=None, max_iter=1, coarsest="amg"):
        self.A = A
        self.labels = np.ascontiguousarray(labels, dtype=np.int32)
        self.params = coarse_params if coarse_params is not None else AmgParams.default()
        arr = (MgrLevelParams * max(len(levels), 1))()
        self._keep = []
        for k, lv in enumerate(levels):
            f = np.ascontiguousarray(lv["f_dofs"], dtype=np.int32)
            self._keep.append(f)
            arr[k].n_f_labels = len(f)
            arr[k].f_labels = f.ctypes.data_as(C.POINTER(C.c_int))
            arr[k].interp_type = MGR_INTERP[lv.get("prolongation_type", "injection")]
            arr[k].restrict_type = MGR_RESTRICT[lv.get("restriction_type", "injection")]
            arr[k].frelax_type = MGR_FRELAX[lv.get("f_relaxation", "jacobi")]
            arr[k].frelax_sweeps = lv.get("f_sweeps", 1)
            arr[k].grelax_type = MGR_GRELAX[lv.get("g_relaxation", "none")]
            arr[k].grelax_sweeps = lv.get("g_sweeps", 1)
            arr[k].grelax_blocks = lv.get("g_blocks", 1)   # row blocks of the hybrid Gauss-Seidel global relaxation (the reference at np = V)
            if lv.get("f_amg") is not None:   # AmgParams of 'f_relaxation: {amg: {...}}'
                self._keep.append(lv["f_amg"])
                arr[k].frelax_amg = C.pointer(lv["f_amg"])
            ilu = lv.get("ilu", {})
            arr[k].ilu_tri_solve, arr[k].ilu_lower_it, arr[k].ilu_upper_it = ilu.get("tri_solve", 1), ilu.get("lower_jac_iters", 5), ilu.get("upper_jac_iters", 5)
            cil = lv.get("coarsest_ilu", {})
            arr[k].coarse_ilu_max_iter, arr[k].coarse_ilu_tri_solve = cil.get("max_iter", 1), cil.get("tri_solve", 1)
            arr[k].coarse_ilu_lower_it, arr[k].coarse_ilu_upper_it = cil.get("lower_jac_iters", 5), cil.get("upper_jac_iters", 5)
            _mgr_nested(arr[k], lv)
        self.nlevels = len(levels)
        self.h = C.c_void_p()
        _check(load().hda_mgr_create(A.h, _ip(self.labels), len(levels), arr, C.byref(self.params) if coarsest == "amg" else None,
                                     max_iter, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            load().hda_amg_destroy(self.h)
            self.h = None

    def matrix(self, level, which=0):
        out = C.c_void_p()
        _check(load().hda_mgr_matrix(self.h, level, which, C.byref(out)))
        return Csr(out, owned=False, keep=self)

    def vcycle(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        _check(load().hda_amg_vcycle(self.h, _dp(b), _dp(x)))
        return x


class Ilu:
    """'preconditioner: ilu': block-Jacobi ILU(0) of A's diagonal block; usable as amg= in pcg()/gmres()."""

    def __init__(self, A, max_iter=1, tri_solve=1, lower_it=5, upper_it=5, blocks=1, block_part=None):
        """blocks: V contiguous row blocks = bj-iluk at np = V (1 one block, 0 the setup's choice); block_part: V + 1 row starts"""
        self.A = A
        self.h = C.c_void_p()
        if blocks == 1 and block_part is None:
            _check(load().hda_ilu_create(A.h, max_iter, tri_solve, lower_it, upper_it, C.byref(self.h)))
        else:
            bp = None
            if block_part is not None:
                bp = np.ascontiguousarray(block_part, dtype=np.int64)
                blocks = len(bp) - 1
            _check(load().hda_ilu_create_blocks(A.h, max_iter, tri_solve, lower_it, upper_it, blocks,
                                                None if bp is None else bp.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(self.h)))

    @property
    def blocks(self):
        return load().hda_ilu_blocks(self.h, -1)

    def __del__(self):
        if getattr(self, "h", None):
            load().hda_amg_destroy(self.h)
            self.h = None

    @property
    def factors(self):
        out = C.c_void_p()
        _check(load().hda_ilu_factors(self.h, -1, C.byref(out)))
        return Csr(out, owned=False, keep=self)

    def apply(self, r):
        """One application from a zero guess (max_iter iterations x += M^-1 (r - A x))."""
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        _check(load().hda_amg_vcycle(self.h, _dp(r), _dp(z)))
        return z


def _krylov(fn, A, b, amg, kp, x0):
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
    hist = np.zeros(kp.max_iter + 2)
    it, conv, frel = C.c_int(), C.c_int(), C.c_double()
    _check(fn(A.h, amg.h if amg is not None else None, C.byref(kp), _dp(b), _dp(x), _dp(hist),
              C.byref(it), C.byref(conv), C.byref(frel)))
    return dict(x=x, iters=it.value, converged=bool(conv.value), final_rel=frel.value,
                hist=hist[:it.value + 1].copy())


def pcg(A, b, amg=None, kp=None, x0=None):
    return _krylov(load().hda_pcg, A, b, amg, kp or KrylovParams.default(False), x0)


def gmres(A, b, amg=None, kp=None, x0=None):
    return _krylov(load().hda_gmres, A, b, amg, kp or KrylovParams.default(True), x0)


def fgmres(A, b, amg=None, kp=None, x0=None):
    return _krylov(load().hda_fgmres, A, b, amg, kp or KrylovParams.default(True), x0)


def bicgstab(A, b, amg=None, kp=None, x0=None):
    return _krylov(load().hda_bicgstab, A, b, amg, kp or KrylovParams.default(False), x0)


def time_kernel(kind, A, amg=None, reps=20):
    ms, by = C.c_double(), C.c_double()
    _check(load().hda_time_kernel(kind, A.h, amg.h if amg is not None else None, reps, C.byref(ms), C.byref(by)))
    return ms.value, by.value


def solve_device(A, amg=None, kp=None, b=None, solver=0, nsolves=1, profile_k1=True):
    """nsolves device-resident solves from x0 = 0 against an existing hierarchy."""
    kp = kp or KrylovParams.default(bool(solver))
    times = np.zeros(max(nsolves, 1))
    it = C.c_int()
    d = [C.c_double() for _ in range(4)]
    bb = None if b is None else _dp(np.ascontiguousarray(b, dtype=np.float64))
    _check(load().hda_solve_device(A.h, amg.h if amg is not None else None, C.byref(kp), solver, bb, nsolves,
                                   _dp(times), C.byref(it), C.byref(d[0]), C.byref(d[1]), C.byref(d[2]),
                                   C.byref(d[3]) if profile_k1 else None))
    return dict(solve_ms=times, iters=it.value, final_rel=d[0].value, r0=d[1].value, true_rel=d[2].value,
                k1_avg_ms=d[3].value, precond_calls=load().hda_last_precond_calls())


def pcg_iteration_bytes(A):
    return load().hda_pcg_iteration_bytes(A.h)


def format_bytes(A, amg=None):
    """Bytes really streamed when operators are stencil-coded: dict(pcg_iteration, vcycle, spmv, coded)."""
    a, b, c, d = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    _check(load().hda_format_bytes(A.h, amg.h if amg is not None else None, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
    return {"pcg_iteration": a.value, "vcycle": b.value, "spmv": c.value, "coded": d.value in (1, 2), "row_coded": d.value == 2,
            "windowed": d.value in (3, 5), "value_coded": d.value in (4, 5)}


def probe_spmv(A, mode):
    """Arm the launch-timing probe on matrix A (None disarms); mode 0 y=Ax, 1 residual, 2 Jacobi sweep."""
    _check(load().hda_probe_spmv(A.h if A is not None else None, mode))


def probe_read():
    ms, n = C.c_double(), C.c_int()
    _check(load().hda_probe_read(C.byref(ms), C.byref(n)))
    return ms.value, n.value


def probe_add(A, mode):
    """Arm one more probe (several may be armed at once); returns its id for probe_read_id."""
    k = C.c_int()
    _check(load().hda_probe_add(A.h, mode, C.byref(k)))
    return k.value


def probe_read_id(k):
    ms, n = C.c_double(), C.c_int()
    _check(load().hda_probe_read_id(k, C.byref(ms), C.byref(n)))
    return ms.value, n.value


def borrow(hypredrv_obj):
    """(Csr, Amg) views of what a hypredrv.Hypredrv object built (level-0 operator + hierarchy); borrowed, not copied."""
    a, g = C.c_void_p(), C.c_void_p()
    _check(load().hda_borrow_hypredrv(hypredrv_obj.h, C.byref(a), C.byref(g)))
    A = Csr(a, owned=True, keep=hypredrv_obj)  # destroying the view does not touch the operator
    amg = Amg.__new__(Amg)
    amg.A, amg.params, amg.h = A, None, g
    return A, amg


def borrow_matrix(hypredrv_obj):
    """Csr view of the level-0 operator of a hypredrv.Hypredrv object whatever its preconditioner is (MGR, ILU, ...)."""
    a = C.c_void_p()
    _check(load().hda_borrow_hypredrv(hypredrv_obj.h, C.byref(a), None))
    return Csr(a, owned=True, keep=hypredrv_obj)


def comm_stats(reset=False):
    v = (C.c_double * 5)()
    load().hda_comm_stats(v, 1 if reset else 0)
    return dict(allreduce=v[0], exchange=v[1], allreduce_doubles=v[2], exchange_doubles=v[3], overlapped=v[4])


def thread_ranks_lap7(nranks, n, P, yaml, nsolves=1, want_x=False):
    """`nranks` ranks of a row partition as threads of this process (hda_thread_ranks.hip; test seam): AMG-Krylov on the
    generator's 7-pt Laplacian, global grid n, rank grid P.  Returns the dict of one rank-0 result (+ "x" in block numbering)."""
    L = load_testranks()
    out = (C.c_double * 16)()
    err = C.create_string_buffer(4096)
    N = int(n[0]) * int(n[1]) * int(n[2])
    x = np.zeros(N) if want_x else None
    L.hda_thread_ranks_lap7.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.c_char_p, C.c_int]
    rc = L.hda_thread_ranks_lap7(int(nranks), (C.c_int * 3)(*n), (C.c_int * 3)(*P), yaml.encode(), int(nsolves), out,
                                 x.ctypes.data_as(C.POINTER(C.c_double)) if want_x else None, err, len(err))
    if rc != 0:
        raise LibraryError(f"hda_thread_ranks_lap7 failed ({rc}): {err.value.decode()}")
    keys = ("iters", "converged", "final_rel", "norm", "l1", "linf", "allreduce", "exchange", "overlapped", "allreduce_doubles",
            "exchange_doubles", "vcycles", "partitioned_levels", "iters_spread", "world")
    res = {k: out[i] for i, k in enumerate(keys)}
    for k in ("iters", "vcycles", "partitioned_levels", "iters_spread", "world"):
        res[k] = int(res[k])
    res["converged"] = bool(res["converged"])
    if want_x:
        res["x"] = x
    return res


def run_thread_ranks(nranks, body):
    """Test seam: `nranks` Python threads, each joined to one in-process world as its rank (hda_thread_world_*), run
    body(rank, nranks) -- which drives the public API like one rank of an MPI job would -- and return the list of results in rank
    order.  ctypes releases the interpreter lock inside library calls, so ranks blocked in a collective do not stall the others.
    A rank that raises releases its peers with an error; the first exception is re-raised here."""
    import threading
    L = load_testranks()
    L.hda_thread_world_create.restype = C.c_void_p
    L.hda_thread_world_join.argtypes = [C.c_void_p, C.c_int]
    L.hda_thread_world_leave.argtypes = [C.c_void_p, C.c_int]
    L.hda_thread_world_destroy.argtypes = [C.c_void_p]
    L.hda_thread_world_destroy.restype = None
    world = L.hda_thread_world_create(int(nranks))
    if not world:
        raise LibraryError("hda_thread_world_create failed")
    res, errs = [None] * nranks, [None] * nranks

    def run(rank):
        failed = 0
        try:
            if L.hda_thread_world_join(world, rank) != 0:
                raise LibraryError(f"rank {rank} could not join the thread world")
            res[rank] = body(rank, nranks)
        except BaseException as e:  # noqa: BLE001 - reported by the caller's thread below
            errs[rank], failed = e, 1
        finally:
            L.hda_thread_world_leave(world, failed)

    th = [threading.Thread(target=run, args=(r,), name=f"hda-rank-{r}") for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    L.hda_thread_world_destroy(world)
    first = [e for e in errs if e is not None and "another rank failed" not in str(e)] or [e for e in errs if e is not None]
    if first:
        raise first[0]
    return res


def comm_name():
    load().hda_comm_name.restype = C.c_char_p
    return load().hda_comm_name().decode()


def check_row_total(nrows, row_len):
    """Run the int32 size guard of the setup stages on nrows rows of row_len entries (raises LibraryError past 2^31-1)."""
    _check(load().hda_check_row_total(int(nrows), int(row_len)))


def memory_stats():
    a, b = C.c_double(), C.c_double()
    load().hda_memory_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def memory_cached():
    return load().hda_memory_cached()


def memory_driver_stats(reset=False):
    """hipMalloc calls that reached the driver since the last reset, host ms spent in them, bytes they returned."""
    v = (C.c_double * 3)()
    load().hda_memory_driver_stats(v, 1 if reset else 0)
    return {"hipmalloc_calls": int(v[0]), "hipmalloc_ms": v[1], "hipmalloc_gb": v[2] / 1e9}


def memory_trim():
    """cached device blocks of this thread's allocator back to the driver (another process is about to need the memory)"""
    _check(load().hda_memory_trim())


def sync():
    _check(load().hda_device_sync())
