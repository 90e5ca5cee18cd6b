"""ctypes binding of the CPU ORACLE (oracle/amg_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Product code (hypredrive_amd/) must never import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CsrStruct(C.Structure):
    _fields_ = [("nrows", C.c_int), ("ncols", C.c_int), ("rowptr", C.POINTER(C.c_int)),
                ("col", C.POINTER(C.c_int)), ("val", C.POINTER(C.c_double))]


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
                ("agg_num_levels", C.c_int), ("agg_num_paths", C.c_int), ("agg_interp_type", C.c_int),
                ("agg_pmax", C.c_int), ("agg_trunc_factor", C.c_double),
                ("blocks", C.c_int), ("block_part", C.POINTER(C.c_int64)), ("pmis_rng", C.c_int)]


class KrylovParams(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("rtol", C.c_double), ("atol", C.c_double),
                ("two_norm", C.c_int), ("krylov_dim", C.c_int)]


class MgrLevelParams(C.Structure):
    _fields_ = [("n_f_labels", C.c_int), ("f_labels", C.POINTER(C.c_int)), ("interp_type", C.c_int), ("restrict_type", C.c_int),
                ("frelax_type", C.c_int), ("frelax_sweeps", C.c_int), ("grelax_type", C.c_int), ("grelax_sweeps", C.c_int),
                ("frelax_amg", C.POINTER(AmgParams)),
                ("ilu_tri_solve", C.c_int), ("ilu_lower_it", C.c_int), ("ilu_upper_it", C.c_int),
                ("coarse_ilu_max_iter", C.c_int), ("coarse_ilu_tri_solve", C.c_int), ("coarse_ilu_lower_it", C.c_int), ("coarse_ilu_upper_it", C.c_int),
                ("frelax_krylov", C.c_int), ("frelax_krylov_precond", C.c_int), ("frelax_kp", KrylovParams),
                ("coarse_krylov", C.c_int), ("coarse_krylov_precond", C.c_int), ("coarse_kp", KrylovParams),
                ("mgr_cycle", C.c_int), ("mgr_frelax_pos", C.c_int), ("mgr_gsmooth_pos", C.c_int), ("grelax_blocks", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "amg_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # libgomp sizes its team when it is loaded: a GPU box exposes hundreds of logical CPUs but
    # gives this process a small share, and an oversubscribed, spinning team turns the small
    # parity problems into minutes.  Default to the CPUs we may run on (at most 16), passive waits.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(avail, 16))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    L = C.CDLL(build())
    P = C.POINTER
    cp = P(CsrStruct)
    dp = P(C.c_double)
    ip = P(C.c_int)
    L.orc_amg_default_params.argtypes = [P(AmgParams), C.c_int]
    L.orc_krylov_default_params.argtypes = [P(KrylovParams), C.c_int]
    L.orc_csr_free.argtypes = [cp]
    L.orc_csr_from_arrays.restype = cp
    L.orc_csr_from_arrays.argtypes = [C.c_int, C.c_int, P(C.c_int64), P(C.c_int64), dp]
    L.orc_csr_transpose.restype = cp
    L.orc_csr_transpose.argtypes = [cp]
    L.orc_lap7.restype = cp
    L.orc_lap7.argtypes = [C.c_int] * 6 + [C.c_double] * 3 + [C.c_int, dp]
    L.orc_lap7_partition.argtypes = [C.c_int] * 7 + [P(C.c_int64), P(C.c_int64)]
    L.orc_spmv.argtypes = [cp, C.c_double, dp, C.c_double, dp]
    L.orc_dot.restype = C.c_double
    L.orc_dot.argtypes = [C.c_int, dp, dp]
    L.orc_l1_norms.argtypes = [cp, C.c_int, dp]
    L.orc_relax.argtypes = [cp, dp, C.c_int, C.c_double, dp, dp, dp]
    L.orc_strength.argtypes = [cp, C.c_double, C.c_double, P(C.c_ubyte)]
    L.orc_pmis.argtypes = [cp, P(C.c_ubyte), C.c_uint64, C.c_int, C.c_int64, ip]
    L.orc_rs_first_pass.argtypes = [cp, P(C.c_ubyte), ip]
    L.orc_hmis_blocks.argtypes = [cp, P(C.c_ubyte), C.c_int, P(C.c_int64), C.c_uint64, C.c_int, ip]
    L.orc_l1_norms_blocks.argtypes = [cp, C.c_int, C.c_int, P(C.c_int64), dp]
    L.orc_relax_blocks.argtypes = [cp, dp, C.c_int, C.c_double, dp, dp, dp, C.c_int, P(C.c_int64)]
    L.orc_amg_block_part.restype = P(C.c_int64)
    L.orc_amg_block_part.argtypes = [C.c_void_p, C.c_int]
    L.orc_interp_extpi.restype = cp
    L.orc_interp_extpi.argtypes = [cp, P(C.c_ubyte), ip, C.c_int, C.c_double]
    L.orc_rap.restype = cp
    L.orc_rap.argtypes = [cp, cp]
    L.orc_second_strength.restype = cp
    L.orc_second_strength.argtypes = [cp, P(C.c_ubyte), ip, C.c_int]
    L.orc_coarsen_second_pass.argtypes = [cp, P(C.c_ubyte), C.c_int, C.c_uint64, C.c_int, ip]
    L.orc_interp_multipass.restype = cp
    L.orc_interp_multipass.argtypes = [cp, P(C.c_ubyte), ip]
    L.orc_truncate_rows.argtypes = [cp, C.c_int, C.c_double]
    L.orc_amg_setup.restype = C.c_void_p
    L.orc_amg_setup.argtypes = [cp, P(AmgParams)]
    L.orc_amg_setup_dof.restype = C.c_void_p
    L.orc_amg_setup_dof.argtypes = [cp, P(AmgParams), ip]
    L.orc_strength_dof.argtypes = [cp, C.c_double, C.c_double, ip, P(C.c_ubyte)]
    L.orc_interp_extpi_dof.restype = cp
    L.orc_interp_extpi_dof.argtypes = [cp, P(C.c_ubyte), ip, C.c_int, C.c_double, ip]
    L.orc_interp_mm_extpi_dof.restype = cp
    L.orc_interp_mm_extpi_dof.argtypes = [cp, P(C.c_ubyte), ip, C.c_int, C.c_double, ip]
    L.orc_interp_standard_dof.restype = cp
    L.orc_interp_standard_dof.argtypes = [cp, P(C.c_ubyte), ip, C.c_int, C.c_double, ip]
    L.orc_interp_direct_dof.restype = cp
    L.orc_interp_direct_dof.argtypes = [cp, P(C.c_ubyte), ip, C.c_int, C.c_double, ip]
    L.orc_amg_free.argtypes = [C.c_void_p]
    L.orc_amg_num_levels.argtypes = [C.c_void_p]
    L.orc_amg_A.restype = cp
    L.orc_amg_A.argtypes = [C.c_void_p, C.c_int]
    L.orc_amg_P.restype = cp
    L.orc_amg_P.argtypes = [C.c_void_p, C.c_int]
    L.orc_amg_cf.restype = ip
    L.orc_amg_cf.argtypes = [C.c_void_p, C.c_int]
    L.orc_amg_l1.restype = dp
    L.orc_amg_l1.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_amg_operator_complexity.restype = C.c_double
    L.orc_amg_operator_complexity.argtypes = [C.c_void_p]
    L.orc_amg_grid_complexity.restype = C.c_double
    L.orc_amg_grid_complexity.argtypes = [C.c_void_p]
    L.orc_amg_vcycle.argtypes = [C.c_void_p, dp, dp]
    for f in (L.orc_pcg, L.orc_gmres, L.orc_fgmres, L.orc_bicgstab):
        f.restype = C.c_int
        f.argtypes = [cp, C.c_void_p, P(KrylovParams), dp, dp, dp, ip, dp]
    L.orc_gselim.argtypes = [dp, dp, C.c_int]
    L.orc_cheby_setup.argtypes = [cp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_int, dp, dp, dp, dp]
    L.orc_cheby_apply.argtypes = [cp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp]
    L.orc_precond_mgr.restype = C.c_void_p
    L.orc_precond_mgr.argtypes = [cp, ip, C.c_int, P(MgrLevelParams), P(AmgParams), C.c_int]
    L.orc_mgr_matrix.restype = cp
    L.orc_mgr_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_amg_rebind_level0.restype = C.c_int
    L.orc_amg_rebind_level0.argtypes = [C.c_void_p, cp]
    i64p = P(C.c_int64)
    L.orc_ilu0_setup.restype = C.c_void_p
    L.orc_ilu0_setup.argtypes = [cp, C.c_int, i64p, C.c_int, C.c_int, C.c_int]
    L.orc_ilu_free.argtypes = [C.c_void_p]
    L.orc_ilu_factors.restype = cp
    L.orc_ilu_factors.argtypes = [C.c_void_p]
    L.orc_ilu_apply.argtypes = [C.c_void_p, dp, dp]
    L.orc_amg_set_ilu_smoother.restype = C.c_int
    L.orc_amg_set_ilu_smoother.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i64p, C.c_int, C.c_int, C.c_int]
    L.orc_precond_ilu.restype = C.c_void_p
    L.orc_precond_ilu.argtypes = [cp, C.c_int, C.c_int, i64p, C.c_int, C.c_int, C.c_int]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Csr:
    """Owns (or borrows) an orc_csr*; exposes numpy views."""

    def __init__(self, ptr, owned=True):
        self.ptr = ptr
        self.owned = owned

    def __del__(self):
        if getattr(self, "owned", False) and self.ptr:
            lib().orc_csr_free(self.ptr)
            self.ptr = None

    @property
    def nrows(self):
        return self.ptr.contents.nrows

    @property
    def ncols(self):
        return self.ptr.contents.ncols

    @property
    def rowptr(self):
        return np.ctypeslib.as_array(self.ptr.contents.rowptr, shape=(self.nrows + 1,))

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    @property
    def col(self):
        return np.ctypeslib.as_array(self.ptr.contents.col, shape=(max(self.nnz, 1),))[:self.nnz]

    @property
    def val(self):
        return np.ctypeslib.as_array(self.ptr.contents.val, shape=(max(self.nnz, 1),))[:self.nnz]

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val.copy(), self.col.copy(), self.rowptr.copy()),
                             shape=(self.nrows, self.ncols))

    @staticmethod
    def from_arrays(nrows, ncols, rowptr, cols, vals):
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        cj = np.ascontiguousarray(cols, dtype=np.int64)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        p = lib().orc_csr_from_arrays(nrows, ncols, rp.ctypes.data_as(C.POINTER(C.c_int64)),
                                      cj.ctypes.data_as(C.POINTER(C.c_int64)), _dp(v))
        return Csr(p)

    @staticmethod
    def from_scipy(m):
        m = m.tocsr()
        return Csr.from_arrays(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)


def lap7(nx, ny, nz, P=(1, 1, 1), c=(1.0, 1.0, 1.0), b_mode=0):
    n = nx * ny * nz
    b = np.zeros(n)
    p = lib().orc_lap7(nx, ny, nz, P[0], P[1], P[2], c[0], c[1], c[2], b_mode, _dp(b))
    return Csr(p), b


def lap7_partition(nx, ny, nz, P, rank):
    lo = C.c_int64()
    hi = C.c_int64()
    lib().orc_lap7_partition(nx, ny, nz, P[0], P[1], P[2], rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def _bpart(part):
    a = np.ascontiguousarray(part, dtype=np.int64)
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def even_blocks(n, blocks):
    """hypre's even split of n rows into `blocks` contiguous blocks (start q = floor(q * n / blocks))."""
    return np.array([(q * n) // blocks for q in range(blocks + 1)], dtype=np.int64)


def amg_params(gpu_defaults=True, **kw):
    p = AmgParams()
    lib().orc_amg_default_params(C.byref(p), 1 if gpu_defaults else 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        if k == "block_part":
            if v is None:
                continue
            keep, ptr = _bpart(v)
            p._block_part_keep = keep  # (the struct borrows the array until the setup has read it)
            p.block_part = ptr
            continue
        setattr(p, k, v)
    return p


def krylov_params(gmres=False, **kw):
    p = KrylovParams()
    lib().orc_krylov_default_params(C.byref(p), 1 if gmres else 0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def spmv(A, x, alpha=1.0, beta=0.0, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(A.nrows) if y is None else np.ascontiguousarray(y, dtype=np.float64).copy()
    lib().orc_spmv(A.ptr, alpha, _dp(x), beta, _dp(y))
    return y


def l1_norms(A, option=1):
    out = np.zeros(A.nrows)
    lib().orc_l1_norms(A.ptr, option, _dp(out))
    return out


def relax(A, l1, rtype, weight, b, x):
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    b = np.ascontiguousarray(b, dtype=np.float64)
    tmp = np.zeros(A.nrows)
    l1 = np.ascontiguousarray(l1, dtype=np.float64)
    lib().orc_relax(A.ptr, _dp(l1), rtype, weight, _dp(b), _dp(x), _dp(tmp))
    return x


def l1_norms_blocks(A, option, part):
    out = np.zeros(A.nrows)
    keep, ptr = _bpart(part)
    lib().orc_l1_norms_blocks(A.ptr, option, len(keep) - 1, ptr, _dp(out))
    return out


def relax_blocks(A, l1, rtype, weight, b, x, part):
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    b = np.ascontiguousarray(b, dtype=np.float64)
    tmp = np.zeros(A.nrows)
    l1 = np.ascontiguousarray(l1, dtype=np.float64)
    keep, ptr = _bpart(part)
    lib().orc_relax_blocks(A.ptr, _dp(l1), rtype, weight, _dp(b), _dp(x), _dp(tmp), len(keep) - 1, ptr)
    return x


def hmis_blocks(A, smask, part, seed=2747, level=0):
    cf = np.zeros(A.nrows, dtype=np.int32)
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    keep, ptr = _bpart(part)
    lib().orc_hmis_blocks(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), len(keep) - 1, ptr, seed, level,
                          cf.ctypes.data_as(C.POINTER(C.c_int)))
    return cf


def strength(A, theta=0.25, max_row_sum=0.9, dof=None):
    sm = np.zeros(max(A.nnz, 1), dtype=np.uint8)
    if dof is None:
        lib().orc_strength(A.ptr, theta, max_row_sum, sm.ctypes.data_as(C.POINTER(C.c_ubyte)))
    else:
        d = np.ascontiguousarray(dof, dtype=np.int32)
        lib().orc_strength_dof(A.ptr, theta, max_row_sum, d.ctypes.data_as(C.POINTER(C.c_int)), sm.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return sm[:A.nnz]


def pmis(A, smask, seed=2747, level=0, row_offset=0):
    cf = np.zeros(A.nrows, dtype=np.int32)
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    lib().orc_pmis(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), seed, level, row_offset,
                   cf.ctypes.data_as(C.POINTER(C.c_int)))
    return cf


def pmis_weights(A, smask, rnd):
    """PMIS with the caller's tie-break values in [0, 1)."""
    cf = np.zeros(A.nrows, dtype=np.int32)
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    r = np.ascontiguousarray(rnd, dtype=np.float64)
    lib().orc_pmis_weights.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib().orc_pmis_weights(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(r), cf.ctypes.data_as(C.POINTER(C.c_int)))
    return cf


def pmis_hypre_stream(n, part=None):
    """hypre_Rand seeded 2747 + rank, one draw per row of every row block (orc_pmis_hypre_stream)."""
    out = np.zeros(max(n, 1))
    lib().orc_pmis_hypre_stream.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    if part is None:
        lib().orc_pmis_hypre_stream(n, 1, None, _dp(out))
    else:
        keep, ptr = _bpart(part)
        lib().orc_pmis_hypre_stream(n, len(keep) - 1, ptr, _dp(out))
    return out[:n]


def rs_first_pass(A, smask):
    cf = np.zeros(A.nrows, dtype=np.int32)
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    lib().orc_rs_first_pass(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)),
                            cf.ctypes.data_as(C.POINTER(C.c_int)))
    return cf


def interp_extpi(A, smask, cf, pmax=4, trunc_factor=0.0, dof=None):
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    cfa = np.ascontiguousarray(cf, dtype=np.int32)
    if dof is not None:
        d = np.ascontiguousarray(dof, dtype=np.int32)
        return Csr(lib().orc_interp_extpi_dof(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), cfa.ctypes.data_as(C.POINTER(C.c_int)),
                                              pmax, trunc_factor, d.ctypes.data_as(C.POINTER(C.c_int))))
    return Csr(lib().orc_interp_extpi(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                      cfa.ctypes.data_as(C.POINTER(C.c_int)), pmax,
                                      trunc_factor))


def interp_mm_extpi(A, smask, cf, pmax=4, trunc_factor=0.0, dof=None):
    """interp type 17 (mm-ext+i): W = -D^-1 (I + B) A^s_FC, truncated"""
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    cfa = np.ascontiguousarray(cf, dtype=np.int32)
    d = None if dof is None else np.ascontiguousarray(dof, dtype=np.int32)
    return Csr(lib().orc_interp_mm_extpi_dof(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), cfa.ctypes.data_as(C.POINTER(C.c_int)), pmax,
                                             trunc_factor, None if d is None else d.ctypes.data_as(C.POINTER(C.c_int))))


def interp_direct(A, smask, cf, pmax=4, trunc_factor=0.0, dof=None):
    """interp type 3 (direct_sep_weights)"""
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    cfa = np.ascontiguousarray(cf, dtype=np.int32)
    d = None if dof is None else np.ascontiguousarray(dof, dtype=np.int32)
    return Csr(lib().orc_interp_direct_dof(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), cfa.ctypes.data_as(C.POINTER(C.c_int)), pmax,
                                           trunc_factor, None if d is None else d.ctypes.data_as(C.POINTER(C.c_int))))


def interp_standard(A, smask, cf, pmax=4, trunc_factor=0.0, dof=None):
    """interp type 8 (standard)"""
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    cfa = np.ascontiguousarray(cf, dtype=np.int32)
    d = None if dof is None else np.ascontiguousarray(dof, dtype=np.int32)
    return Csr(lib().orc_interp_standard_dof(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), cfa.ctypes.data_as(C.POINTER(C.c_int)), pmax,
                                             trunc_factor, None if d is None else d.ctypes.data_as(C.POINTER(C.c_int))))


def second_strength(A, smask, cf, num_paths=1):
    """aggressive coarsening: strong connections of distance <= 2 among the C points of cf (values = number of paths)"""
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    cfa = np.ascontiguousarray(cf, dtype=np.int32)
    return Csr(lib().orc_second_strength(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), cfa.ctypes.data_as(C.POINTER(C.c_int)), num_paths))


def coarsen_second_pass(A, smask, cf, num_paths=1, seed=2747, level=0):
    """aggressive coarsening: the second PMIS pass; returns the updated C/F marker"""
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    cfa = np.ascontiguousarray(cf, dtype=np.int32).copy()
    lib().orc_coarsen_second_pass(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), num_paths, seed, level, cfa.ctypes.data_as(C.POINTER(C.c_int)))
    return cfa


def interp_multipass(A, smask, cf):
    """aggressive coarsening: multipass interpolation (aggressive.prolongation_type 4)"""
    sm = np.ascontiguousarray(smask, dtype=np.uint8)
    cfa = np.ascontiguousarray(cf, dtype=np.int32)
    return Csr(lib().orc_interp_multipass(A.ptr, sm.ctypes.data_as(C.POINTER(C.c_ubyte)), cfa.ctypes.data_as(C.POINTER(C.c_int))))


def truncate_rows(P, pmax=0, trunc_factor=0.0):
    """hypre_BoomerAMGInterpTruncation on finished rows, in place (the aggressive levels' multipass interpolation)"""
    lib().orc_truncate_rows(P.ptr, pmax, trunc_factor)
    return P


def rap(A, P):
    return Csr(lib().orc_rap(A.ptr, P.ptr))


class Amg:
    def __init__(self, A, params=None, dof=None):
        self.A = A
        self.params = params if params is not None else amg_params(True)
        if dof is None:
            self.h = lib().orc_amg_setup(A.ptr, C.byref(self.params))
        else:
            d = np.ascontiguousarray(dof, dtype=np.int32)
            self.h = lib().orc_amg_setup_dof(A.ptr, C.byref(self.params), d.ctypes.data_as(C.POINTER(C.c_int)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_amg_free(self.h)
            self.h = None

    @property
    def num_levels(self):
        return lib().orc_amg_num_levels(self.h)

    def level_A(self, l):
        return Csr(lib().orc_amg_A(self.h, l), owned=False)

    def level_P(self, l):
        p = lib().orc_amg_P(self.h, l)
        return Csr(p, owned=False) if p else None

    def level_cf(self, l):
        n = self.level_A(l).nrows
        p = lib().orc_amg_cf(self.h, l)
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if p else None

    def level_block_part(self, l):
        p = lib().orc_amg_block_part(self.h, l)
        return np.ctypeslib.as_array(p, shape=(self.params.blocks + 1,)).copy() if p else None

    def level_l1(self, l, which=0):
        n = self.level_A(l).nrows
        return np.ctypeslib.as_array(lib().orc_amg_l1(self.h, l, which), shape=(n,)).copy()

    @property
    def operator_complexity(self):
        return lib().orc_amg_operator_complexity(self.h)

    @property
    def grid_complexity(self):
        return lib().orc_amg_grid_complexity(self.h)

    def vcycle(self, b, x0=None):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        lib().orc_amg_vcycle(self.h, _dp(b), _dp(x))
        return x

    def rebind_level0(self, A):
        """Keep the hierarchy, take level 0 from A (a later system of a sequence: preconditioner.reuse)."""
        if lib().orc_amg_rebind_level0(self.h, A.ptr):
            raise ValueError("rebind: size mismatch")
        self.A = A

    def set_ilu_smoother(self, num_levels=1, num_sweeps=1, part=None, tri_solve=1, lower_it=5, upper_it=5):
        """amg.smoother.type ilu: ILU(0) replaces the relaxation sweeps on the first num_levels levels."""
        n, pp, keep = _part(part)
        rc = lib().orc_amg_set_ilu_smoother(self.h, num_levels, num_sweeps, n, pp, tri_solve, lower_it, upper_it)
        if rc:
            raise ValueError(f"ILU smoother setup failed ({rc})")


def _part(part):
    if part is None:
        return 0, None, None
    a = np.ascontiguousarray(part, dtype=np.int64)
    return len(a) - 1, a.ctypes.data_as(C.POINTER(C.c_int64)), a


class Cheby:
    """Chebyshev smoother of relax type 16: setup (eigenvalue estimate + coefficients) and u += p(A)(f - A u)."""

    def __init__(self, A, order=2, eig_est=10, variant=0, scale=1, fraction=0.3, seed=2747, level=0):
        self.A, self.order, self.scale = A, order, scale
        n = A.nrows
        self.ds, self.coefs = np.zeros(n), np.zeros(5)
        mx, mn = C.c_double(), C.c_double()
        lib().orc_cheby_setup(A.ptr, order, eig_est, variant, scale, fraction, seed, level, _dp(self.ds), _dp(self.coefs), C.byref(mx), C.byref(mn))
        self.max_eig, self.min_eig = mx.value, mn.value

    def apply(self, f, u=None):
        f = np.ascontiguousarray(f, dtype=np.float64)
        u = np.zeros_like(f) if u is None else np.ascontiguousarray(u, dtype=np.float64).copy()
        r, v, w = np.zeros_like(f), np.zeros_like(f), np.zeros_like(f)
        lib().orc_cheby_apply(self.A.ptr, self.order, self.scale, _dp(self.ds), _dp(self.coefs), _dp(f), _dp(u), _dp(r), _dp(v), _dp(w))
        return u


class Ilu:
    """ILU(0) of the diagonal blocks of a row partition (part = row starts, None = one block)."""

    def __init__(self, A, part=None, tri_solve=1, lower_it=5, upper_it=5):
        self.A = A
        n, pp, self._keep = _part(part)
        self.h = lib().orc_ilu0_setup(A.ptr, n, pp, tri_solve, lower_it, upper_it)
        if not self.h:
            raise ValueError("ILU(0): missing diagonal or zero pivot")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ilu_free(self.h)
            self.h = None

    @property
    def factors(self):
        return Csr(lib().orc_ilu_factors(self.h), owned=False)

    def apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        lib().orc_ilu_apply(self.h, _dp(r), _dp(z))
        return z


class IluPrecond:
    """'preconditioner: ilu' for pcg()/gmres(): max_iter iterations of x += M^-1 (b - A x)."""

    def __init__(self, A, max_iter=1, part=None, tri_solve=1, lower_it=5, upper_it=5):
        self.A = A
        n, pp, self._keep = _part(part)
        self.h = lib().orc_precond_ilu(A.ptr, max_iter, n, pp, tri_solve, lower_it, upper_it)
        if not self.h:
            raise ValueError("ILU(0): missing diagonal or zero pivot")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_amg_free(self.h)
            self.h = None

    def vcycle(self, b, x0=None):
        return Amg.vcycle(self, b, x0)


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


def mgr_level_list(levels):
    """levels: list of dicts with the YAML keys of mgr.level.N (f_dofs, prolongation_type, restriction_type,
    f_relaxation, g_relaxation [, f_sweeps, g_sweeps]) -> ctypes array + keep-alive list"""
    arr = (MgrLevelParams * max(len(levels), 1))()
    keep = []
    for k, lv in enumerate(levels):
        f = np.ascontiguousarray(lv["f_dofs"], dtype=np.int32)
        keep.append(f)
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
            keep.append(lv["f_amg"])
            arr[k].frelax_amg = C.pointer(lv["f_amg"])
        ilu = lv.get("ilu", {})           # ILU arguments of this level's ILU components
        arr[k].ilu_tri_solve, arr[k].ilu_lower_it, arr[k].ilu_upper_it = ilu.get("tri_solve", 1), ilu.get("lower_jac_iters", 5), ilu.get("upper_jac_iters", 5)
        cil = lv.get("coarsest_ilu", {})  # on the last level: 'coarsest_level: ilu'
        arr[k].coarse_ilu_max_iter, arr[k].coarse_ilu_tri_solve = cil.get("max_iter", 1), cil.get("tri_solve", 1)
        arr[k].coarse_ilu_lower_it, arr[k].coarse_ilu_upper_it = cil.get("lower_jac_iters", 5), cil.get("upper_jac_iters", 5)
        _mgr_nested(arr[k], lv)
    return arr, keep


class MgrPrecond:
    """'preconditioner: mgr' for pcg()/gmres()/...: multigrid reduction by dof labels, BoomerAMG on the coarsest system."""

    def __init__(self, A, labels, levels, coarse_params=None, max_iter=1, coarsest="amg"):
        self.A = A
        self.labels = np.ascontiguousarray(labels, dtype=np.int32)
        self.params = coarse_params if coarse_params is not None else amg_params(True)
        arr, self._keep = mgr_level_list(levels)
        self.nlevels = len(levels)
        self.h = lib().orc_precond_mgr(A.ptr, self.labels.ctypes.data_as(C.POINTER(C.c_int)), len(levels), arr,
                                       C.byref(self.params) if coarsest == "amg" else None, max_iter)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_amg_free(self.h)
            self.h = None

    def rebind_level0(self, A):
        return Amg.rebind_level0(self, A)

    def matrix(self, level, which=0):
        """which: 0 operator of the level (level == number of reduction levels: the coarsest system), 1 P, 2 R"""
        return Csr(lib().orc_mgr_matrix(self.h, level, which), owned=False)

    def vcycle(self, b, x0=None):
        return Amg.vcycle(self, b, x0)


def _krylov(fn, A, b, amg, kp, x0):
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
    hist = np.zeros(kp.max_iter + 2)
    conv = C.c_int()
    frel = C.c_double()
    it = fn(A.ptr, amg.h if amg is not None else None, C.byref(kp), _dp(b), _dp(x), _dp(hist),
            C.byref(conv), C.byref(frel))
    return dict(x=x, iters=it, converged=bool(conv.value), final_rel=frel.value,
                hist=hist[:it + 1].copy())


def pcg(A, b, amg=None, kp=None, x0=None):
    return _krylov(lib().orc_pcg, A, b, amg, kp or krylov_params(False), x0)


def gmres(A, b, amg=None, kp=None, x0=None):
    return _krylov(lib().orc_gmres, A, b, amg, kp or krylov_params(True), x0)


def fgmres(A, b, amg=None, kp=None, x0=None):
    return _krylov(lib().orc_fgmres, A, b, amg, kp or krylov_params(True), x0)


def bicgstab(A, b, amg=None, kp=None, x0=None):
    return _krylov(lib().orc_bicgstab, A, b, amg, kp or krylov_params(False), x0)
