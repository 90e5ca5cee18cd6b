"""ctypes view of the reference-facing boundary (include/HYPREDRV.h): the same calls, in the
same order, that a C driver written against hypredrive makes."""
import ctypes as C

import numpy as np

from ._lib import load, LibraryError

_configured = False

# error bits (reference include/internal/error.h:16-48)
ERROR_INVALID_SOLVER = 0x00020000
ERROR_INVALID_PRECON = 0x00040000
ERROR_FILE_NOT_FOUND = 0x00080000
ERROR_FILE_UNEXPECTED_ENTRY = 0x00100000
ERROR_UNKNOWN_HYPREDRV_OBJ = 0x00200000
ERROR_NOT_INITIALIZED = 0x00400000
ERROR_UNKNOWN_TIMING = 0x00800000
ERROR_HYPRE_INTERNAL = 0x01000000
ERROR_UNSUPPORTED_AMD = 0x02000000
ERROR_MISSING_PRECON = 0x00008000
ERROR_MISSING_DOFMAP = 0x00010000
ERROR_MISSING_KEY = 0x00001000
ERROR_INVALID_KEY = 0x00000100
ERROR_INVALID_VAL = 0x00000200
ERROR_UNKNOWN = 0x80000000
MPI_COMM_WORLD = 0x44000000

# staged-transport callbacks (include/HYPREDRV.h): return 0 on success, non-zero makes the library raise on this rank
ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int)
ALLTOALLV_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.POINTER(C.c_long))


def lib():
    global _configured
    L = load()
    if _configured:
        return L
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    u32 = C.c_uint32
    sig = {
        "HYPREDRV_Initialize": [], "HYPREDRV_Finalize": [],
        "HYPREDRV_Create": [C.c_int, C.POINTER(vp)], "HYPREDRV_Destroy": [C.POINTER(vp)],
        "HYPREDRV_SetLibraryMode": [vp], "HYPREDRV_InputArgsParse": [C.c_int, C.POINTER(C.c_char_p), vp],
        "HYPREDRV_InputArgsSetPreconPreset": [vp, C.c_char_p], "HYPREDRV_InputArgsSetSolverPreset": [vp, C.c_char_p],
        "HYPREDRV_InputArgsGetNumRepetitions": [vp, ip], "HYPREDRV_InputArgsGetNumLinearSystems": [vp, ip],
        "HYPREDRV_InputArgsGetNumPreconVariants": [vp, ip], "HYPREDRV_InputArgsSetPreconVariant": [vp, C.c_int],
        "HYPREDRV_LinearSystemBuild": [vp],
        "HYPREDRV_LinearSystemSetMatrixFromCSR": [vp, C.c_longlong, C.c_longlong, C.POINTER(C.c_longlong),
                                                  C.POINTER(C.c_longlong), dp],
        "HYPREDRV_LinearSystemSetRHSFromArray": [vp, C.c_longlong, C.c_longlong, dp],
        "HYPREDRV_LinearSystemSetInitialGuess": [vp, vp], "HYPREDRV_LinearSystemSetPrecMatrix": [vp, vp],
        "HYPREDRV_LinearSystemResetInitialGuess": [vp],
        "HYPREDRV_LinearSystemGetSolutionValues": [vp, C.POINTER(dp)],
        "HYPREDRV_LinearSystemGetSolutionLength": [vp, C.POINTER(C.c_longlong)],
        "HYPREDRV_LinearSystemGetSolutionNorm": [vp, C.c_char_p, dp],
        "HYPREDRV_PreconCreate": [vp], "HYPREDRV_LinearSolverCreate": [vp], "HYPREDRV_PreconSetup": [vp],
        "HYPREDRV_LinearSolverSetup": [vp], "HYPREDRV_LinearSolverApply": [vp], "HYPREDRV_PreconDestroy": [vp],
        "HYPREDRV_LinearSolverDestroy": [vp], "HYPREDRV_StatsPrint": [vp],
        "HYPREDRV_AnnotateBegin": [vp, C.c_char_p, C.c_int], "HYPREDRV_AnnotateEnd": [vp, C.c_char_p, C.c_int],
        "HYPREDRV_LinearSolverGetNumIter": [vp, ip], "HYPREDRV_LinearSolverGetConverged": [vp, ip],
        "HYPREDRV_LinearSolverGetFinalRelativeResidualNorm": [vp, dp],
        "HYPREDRV_LinearSolverGetSetupTime": [vp, dp], "HYPREDRV_LinearSolverGetSolveTime": [vp, dp],
        "HYPREDRV_AMD_CommGetUniqueId": [vp], "HYPREDRV_AMD_CommInit": [C.c_int, C.c_int, C.c_int, vp],
        "HYPREDRV_AMD_CommInitCallbacks": [C.c_int, C.c_int, C.c_int, ALLREDUCE_CB, ALLTOALLV_CB],
        "HYPREDRV_AMD_CommFinalize": [],
        "HYPREDRV_AMD_LinearSystemSetLaplacian7pt": [vp, ip, ip, dp],
        "HYPREDRV_AMD_LinearSystemSetEmptyBlock": [vp, C.c_longlong],
        "HYPREDRV_LinearSystemSetDofmap": [vp, C.c_int, ip],
        "HYPREDRV_LinearSystemSetInterleavedDofmap": [vp, C.c_int, C.c_int],
        "HYPREDRV_LinearSystemSetContiguousDofmap": [vp, C.c_int, C.c_int],
        "HYPREDRV_LinearSystemSetNullSpace": [vp, C.c_int, C.c_int, dp],
        "HYPREDRV_LinearSystemReadMatrix": [vp],
        "HYPREDRV_AnnotateLevelBegin": [vp, C.c_int, C.c_char_p, C.c_int],
        "HYPREDRV_AnnotateLevelEnd": [vp, C.c_int, C.c_char_p, C.c_int],
        "HYPREDRV_StatsLevelGetCount": [vp, C.c_int, ip],
        "HYPREDRV_StatsLevelGetEntry": [vp, C.c_int, C.c_int, ip, ip, ip, dp, dp],
        "HYPREDRV_StatsLevelPrint": [vp, C.c_int],
        "HYPREDRV_AMD_SolvePhaseBytes": [vp, dp, dp],
        "HYPREDRV_AMD_ProbeDominant": [vp, C.c_int, ip, dp, dp, ip],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = u32
    L.HYPREDRV_ErrorCodeDescribe.argtypes = [u32]
    L.HYPREDRV_ErrorCodeDescribe.restype = None
    L.HYPREDRV_ErrorCodeClear.restype = None
    L.HYPREDRV_AMD_LastErrorMessage.restype = C.c_char_p
    _configured = True
    return L


class HypredrvError(LibraryError):
    def __init__(self, code, msg):
        super().__init__(f"HYPREDRV error 0x{code:08x}: {msg}")
        self.code = code


def check(code):
    if code:
        msg = lib().HYPREDRV_AMD_LastErrorMessage().decode()
        lib().HYPREDRV_ErrorCodeClear()
        raise HypredrvError(code, msg)


class Hypredrv:
    """Library-mode object, used like examples/src/C_laplacian/laplacian.c:331-468 uses HYPREDRV_t."""

    def __init__(self, yaml_text=None, library_mode=True, overrides=()):
        L = lib()
        check(L.HYPREDRV_Initialize())
        self.h = C.c_void_p()
        check(L.HYPREDRV_Create(MPI_COMM_WORLD, C.byref(self.h)))
        if library_mode:
            check(L.HYPREDRV_SetLibraryMode(self.h))
        if yaml_text is not None:
            self.parse(yaml_text, overrides)

    def parse(self, yaml_text, overrides=()):
        args = [yaml_text.encode()] + [o.encode() for o in overrides]
        argv = (C.c_char_p * len(args))(*args)
        check(lib().HYPREDRV_InputArgsParse(len(args), argv, self.h))

    def presets(self, solver="pcg", precon="poisson"):
        check(lib().HYPREDRV_InputArgsSetSolverPreset(self.h, solver.encode()))
        check(lib().HYPREDRV_InputArgsSetPreconPreset(self.h, precon.encode()))

    def set_matrix_csr(self, row_start, row_end, indptr, cols, data):
        if row_end == row_start - 1:  # a rank that owns nothing: the reference's entry refuses the range, the extension sets matrix AND rhs
            self._empty_block = True
            check(lib().HYPREDRV_AMD_LinearSystemSetEmptyBlock(self.h, row_start))
            return
        self._empty_block = False
        ip = np.ascontiguousarray(indptr, dtype=np.int64)
        cj = np.ascontiguousarray(cols, dtype=np.int64)
        v = np.ascontiguousarray(data, dtype=np.float64)
        ll = C.POINTER(C.c_longlong)
        check(lib().HYPREDRV_LinearSystemSetMatrixFromCSR(self.h, row_start, row_end, ip.ctypes.data_as(ll),
                                                          cj.ctypes.data_as(ll), v.ctypes.data_as(C.POINTER(C.c_double))))

    def set_rhs_array(self, row_start, row_end, values):
        if row_end == row_start - 1 and getattr(self, "_empty_block", False):
            return  # (set together with the empty matrix block)
        v = np.ascontiguousarray(values, dtype=np.float64)
        check(lib().HYPREDRV_LinearSystemSetRHSFromArray(self.h, row_start, row_end, v.ctypes.data_as(C.POINTER(C.c_double))))

    def set_laplacian7(self, n, P=(1, 1, 1), c=(1.0, 1.0, 1.0)):
        check(lib().HYPREDRV_AMD_LinearSystemSetLaplacian7pt(self.h, (C.c_int * 3)(*n), (C.c_int * 3)(*P), (C.c_double * 3)(*c)))

    def finish_system(self):
        check(lib().HYPREDRV_LinearSystemSetInitialGuess(self.h, None))
        check(lib().HYPREDRV_LinearSystemSetPrecMatrix(self.h, None))

    def solve(self):
        """ResetInitialGuess + Create + Setup + Apply + Destroy (laplacian.c:445-463)."""
        L = lib()
        check(L.HYPREDRV_LinearSystemResetInitialGuess(self.h))
        check(L.HYPREDRV_LinearSolverCreate(self.h))
        check(L.HYPREDRV_LinearSolverSetup(self.h))
        check(L.HYPREDRV_LinearSolverApply(self.h))
        out = self.last()
        check(L.HYPREDRV_LinearSolverDestroy(self.h))
        return out

    def create_and_setup(self):
        check(lib().HYPREDRV_LinearSolverCreate(self.h))
        check(lib().HYPREDRV_LinearSolverSetup(self.h))

    def apply(self, reset=True):
        if reset:
            check(lib().HYPREDRV_LinearSystemResetInitialGuess(self.h))
        check(lib().HYPREDRV_LinearSolverApply(self.h))
        return self.last()

    def destroy_solver(self):
        check(lib().HYPREDRV_LinearSolverDestroy(self.h))

    def last(self):
        L = lib()
        it, cv = C.c_int(), C.c_int()
        rel, ts, tv = C.c_double(), C.c_double(), C.c_double()
        check(L.HYPREDRV_LinearSolverGetNumIter(self.h, C.byref(it)))
        check(L.HYPREDRV_LinearSolverGetConverged(self.h, C.byref(cv)))
        check(L.HYPREDRV_LinearSolverGetFinalRelativeResidualNorm(self.h, C.byref(rel)))
        check(L.HYPREDRV_LinearSolverGetSetupTime(self.h, C.byref(ts)))
        check(L.HYPREDRV_LinearSolverGetSolveTime(self.h, C.byref(tv)))
        return dict(iters=it.value, converged=bool(cv.value), final_rel=rel.value, setup_s=ts.value, solve_s=tv.value)

    def solution(self):
        p = C.POINTER(C.c_double)()
        n = C.c_longlong()
        check(lib().HYPREDRV_LinearSystemGetSolutionValues(self.h, C.byref(p)))
        check(lib().HYPREDRV_LinearSystemGetSolutionLength(self.h, C.byref(n)))
        return np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[:n.value].copy()

    def solution_norm(self, kind="L2"):
        v = C.c_double()
        check(lib().HYPREDRV_LinearSystemGetSolutionNorm(self.h, kind.encode(), C.byref(v)))
        return v.value

    def solve_phase_bytes(self):
        """(iteration, vcycle) bytes of this rank: each a pair (CSR figure, bytes in the formats read)."""
        it, vc = (C.c_double * 2)(), (C.c_double * 2)()
        check(lib().HYPREDRV_AMD_SolvePhaseBytes(self.h, it, vc))
        return (it[0], it[1]), (vc[0], vc[1])

    def probe_dominant_arm(self):
        """Time every Jacobi-sweep launch on this rank's largest plain-CSR operator; returns (level, rows, cols, nnz)."""
        lvl, dims = C.c_int(), (C.c_double * 3)()
        check(lib().HYPREDRV_AMD_ProbeDominant(self.h, 1, C.byref(lvl), dims, None, None))
        return lvl.value, int(dims[0]), int(dims[1]), int(dims[2])

    def probe_dominant_read(self):
        ms, n = C.c_double(), C.c_int()
        check(lib().HYPREDRV_AMD_ProbeDominant(self.h, 0, None, None, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stats_print(self):
        check(lib().HYPREDRV_StatsPrint(self.h))

    def close(self):
        if self.h:
            lib().HYPREDRV_Destroy(C.byref(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
