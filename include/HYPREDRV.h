/*
 * HYPREDRV.h -- the hypredrive public C API (HYPREDRV_*), kept prototype-for-prototype so
 * drivers written against the reference (include/HYPREDRV.h:112-2263 there) compile and
 * link against libhypredrv_amd.so unchanged.  Return value: OR of the error bits of the
 * reference's include/internal/error.h:16-48; HYPREDRV_SUCCESS == 0.
 *
 * Scope: the AMG-Krylov solve path.  Entry points outside that path
 * (MGR dofmaps, state vectors, eigenspectrum, Maxwell operators, ...) are exported so
 * that programs link, and report HYPREDRV_ERROR_UNSUPPORTED_AMD via the normal error
 * channel instead of silently doing nothing.
 *
 * Threading contract as in the reference (include/HYPREDRV.h:66-70): single caller thread.
 */
#ifndef HYPREDRV_HEADER
#define HYPREDRV_HEADER

#include <stdint.h>

#include "HYPRE.h"
#include "HYPREDRV_config.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HYPREDRV_EXPORT_SYMBOL __attribute__((visibility("default")))
#define HYPREDRV_SUCCESS ((uint32_t)0u)
/* feature present in the reference but outside this build's hot path (uses the
 * reference's ERROR_MISSING_LIB bit, include/internal/error.h:44) */
#define HYPREDRV_ERROR_UNSUPPORTED_AMD ((uint32_t)0x02000000u)

struct hypredrv_struct;
typedef struct hypredrv_struct *HYPREDRV_t;

/* ---- library lifecycle (ref :112,:138) ---- */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_Initialize(void);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_Finalize(void);
/* ---- error channel (ref :187-221, :1456) ---- */
HYPREDRV_EXPORT_SYMBOL void     HYPREDRV_ErrorCodeDescribe(uint32_t error_code);
HYPREDRV_EXPORT_SYMBOL void     HYPREDRV_ErrorCodeClear(void);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_ErrorInvalidValue(const char *message);
HYPREDRV_EXPORT_SYMBOL void     HYPREDRV_SafeCallHandleError(uint32_t error_code, MPI_Comm comm, const char *file,
                                                             int line, const char *func);
/* ---- object lifecycle / info (ref :257-447, :895) ---- */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_Create(MPI_Comm comm, HYPREDRV_t *hypredrv_ptr);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_Destroy(HYPREDRV_t *hypredrv_ptr);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PrintLibInfo(MPI_Comm comm, int print_datetime);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PrintSystemInfo(MPI_Comm comm);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PrintExitInfo(MPI_Comm comm, const char *argv0);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsParse(int argc, char **argv, HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_SetLibraryMode(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_ObjectSetName(HYPREDRV_t hypredrv, const char *name);
/* ---- parsed-input queries and presets (ref :465-639) ---- */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsGetWarmup(HYPREDRV_t hypredrv, int *warmup);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsGetNumRepetitions(HYPREDRV_t hypredrv, int *num_reps);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsGetNumLinearSystems(HYPREDRV_t hypredrv, int *num_ls);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsGetNumPreconVariants(HYPREDRV_t hypredrv, int *num_variants);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsSetPreconVariant(HYPREDRV_t hypredrv, int variant_idx);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsSetPreconPreset(HYPREDRV_t hypredrv, const char *preset);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_InputArgsSetSolverPreset(HYPREDRV_t hypredrv, const char *preset);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_SolverPresetRegister(const char *name, const char *yaml_text, const char *help);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PreconPresetRegister(const char *name, const char *yaml_text, const char *help);
/* ---- linear system (ref :669-1518) ---- */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemBuild(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemReadMatrix(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetMatrix(HYPREDRV_t hypredrv, HYPRE_Matrix mat_A);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetDiscreteGradient(HYPREDRV_t hypredrv, HYPRE_Matrix G);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetDiscreteCurl(HYPREDRV_t hypredrv, HYPRE_Matrix C);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetCoordinates(HYPREDRV_t hypredrv, HYPRE_Vector x, HYPRE_Vector y,
                                                                    HYPRE_Vector z);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetRHS(HYPREDRV_t hypredrv, HYPRE_Vector vec);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetMatrixFromCSR(HYPREDRV_t hypredrv, HYPRE_BigInt row_start,
                                                                      HYPRE_BigInt row_end, const HYPRE_BigInt *indptr,
                                                                      const HYPRE_BigInt *col_indices,
                                                                      const HYPRE_Real *data);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetRHSFromArray(HYPREDRV_t hypredrv, HYPRE_BigInt row_start,
                                                                     HYPRE_BigInt row_end, const HYPRE_Real *values);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetInitialGuess(HYPREDRV_t hypredrv, HYPRE_Vector vec);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetSolution(HYPREDRV_t hypredrv, HYPRE_Vector vec);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetReferenceSolution(HYPREDRV_t hypredrv, HYPRE_Vector vec);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemResetInitialGuess(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetPrecMatrix(HYPREDRV_t hypredrv, HYPRE_Matrix mat);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetDofmap(HYPREDRV_t hypredrv, int size, const int *dofmap);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetInterleavedDofmap(HYPREDRV_t hypredrv, int num_local_blocks,
                                                                          int num_dof_types);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetContiguousDofmap(HYPREDRV_t hypredrv, int num_local_blocks,
                                                                         int num_dof_types);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemReadDofmap(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemPrintDofmap(HYPREDRV_t hypredrv, const char *filename);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemPrint(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetNearNullSpace(HYPREDRV_t hypredrv, int num_entries,
                                                                      int num_components, const HYPRE_Complex *values);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemSetNullSpace(HYPREDRV_t hypredrv, int num_entries, int num_components,
                                                                  const HYPRE_Complex *values);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemGetSolutionValues(HYPREDRV_t hypredrv, HYPRE_Complex **sol_data);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemGetSolutionLength(HYPREDRV_t hypredrv, HYPRE_BigInt *length);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemGetSolutionNorm(HYPREDRV_t hypredrv, const char *norm_type,
                                                                     double *norm);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemGetSolution(HYPREDRV_t hypredrv, HYPRE_Vector *vec);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemGetRHSValues(HYPREDRV_t hypredrv, HYPRE_Complex **rhs_data);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemGetRHS(HYPREDRV_t hypredrv, HYPRE_Vector *vec);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemGetMatrix(HYPREDRV_t hypredrv, HYPRE_Matrix *mat);
/* ---- state vectors (ref :1554-1694; outside the hot path) ---- */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StateVectorSet(HYPREDRV_t hypredrv, int nstates, HYPRE_IJVector *vecs);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StateVectorGetValues(HYPREDRV_t hypredrv, int index, HYPRE_Complex **data_ptr);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StateVectorCopy(HYPREDRV_t hypredrv, int index_in, int index_out);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StateVectorUpdateAll(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StateVectorApplyCorrection(HYPREDRV_t hypredrv, int state_idx);
/* ---- THE HOT PATH: preconditioner / Krylov lifecycle (ref :1719-1905) ---- */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PreconCreate(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverCreate(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PreconSetup(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverSetup(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverApply(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PreconApply(HYPREDRV_t hypredrv, HYPRE_Vector vec_b, HYPRE_Vector vec_x);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_PreconDestroy(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverDestroy(HYPREDRV_t hypredrv);
/* ---- statistics / annotations / getters (ref :1932-2262) ---- */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StatsPrint(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AnnotateBegin(HYPREDRV_t hypredrv, const char *name, int id);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AnnotateEnd(HYPREDRV_t hypredrv, const char *name, int id);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AnnotateLevelBegin(HYPREDRV_t hypredrv, int level, const char *name, int id);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AnnotateLevelEnd(HYPREDRV_t hypredrv, int level, const char *name, int id);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSystemComputeEigenspectrum(HYPREDRV_t hypredrv);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverGetNumIter(HYPREDRV_t hypredrv, int *iters);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverGetConverged(HYPREDRV_t hypredrv, int *converged);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverGetFinalRelativeResidualNorm(HYPREDRV_t hypredrv, double *norm);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverGetSetupTime(HYPREDRV_t hypredrv, double *seconds);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_LinearSolverGetSolveTime(HYPREDRV_t hypredrv, double *seconds);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StatsLevelGetCount(HYPREDRV_t hypredrv, int level, int *count);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StatsLevelGetEntry(HYPREDRV_t hypredrv, int level, int index, int *entry_id,
                                                            int *num_solves, int *linear_iters, double *setup_time,
                                                            double *solve_time);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_StatsLevelPrint(HYPREDRV_t hypredrv, int level);

/* ---- MI355X additions (not in the reference) -------------------------------------------
 * One process per GPU; ranks are connected by RCCL instead of MPI.  Call before
 * HYPREDRV_Initialize when world_size > 1: unique_id is the 128-byte ncclUniqueId created
 * by rank 0 (HYPREDRV_AMD_CommGetUniqueId) and distributed by the launcher
 * (torch.distributed store, MPI_Bcast, a file ...).  Without it the library runs 1 rank. */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_CommGetUniqueId(void *unique_id_128);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_CommInit(int rank, int world_size, int local_device,
                                                      const void *unique_id_128);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_CommFinalize(void);
/* Second transport for launchers without one GPU per rank (several ranks sharing a device in tests, gloo-only
 * hosts): the same messages staged through two host callbacks the launcher implements on whatever it has
 * (torch.distributed gloo in hypredrive_amd/dist.py, MPI in a C driver).  Both return 0 on success; any other
 * value raises an error on the calling rank.  allreduce: in place, count elements, dtype 0 = double / 1 = int64,
 * op 0 = sum / 1 = max.  alltoallv: buffers packed by ascending peer rank, counts in BYTES, arrays of world_size. */
typedef int (*HYPREDRV_AMD_AllreduceFn)(void *buf, long count, int dtype, int op);
typedef int (*HYPREDRV_AMD_AlltoallvFn)(const void *send, const long *send_bytes, void *recv, const long *recv_bytes);
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_CommInitCallbacks(int rank, int world_size, int local_device,
                                                               HYPREDRV_AMD_AllreduceFn allreduce,
                                                               HYPREDRV_AMD_AlltoallvFn alltoallv);
/* Build the benchmark system of examples/src/C_laplacian/laplacian.c:719-921 directly in HBM
 * (block pc of a P0 x P1 x P2 partition of an n0 x n1 x n2 grid) and attach it as
 * matrix + rhs of the object, as LinearSystemSetMatrix/SetRHS would. */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_LinearSystemSetLaplacian7pt(HYPREDRV_t hypredrv, const int n[3],
                                                                         const int P[3], const double c[3]);
/* Last error text of the MI355X backend (the reference prints through ErrorCodeDescribe). */
/* A rank that owns no rows of a row-partitioned system: empty matrix block and right-hand side at row_start.  (hypre's IJ layer
 * represents such a rank; HYPREDRV_LinearSystemSetMatrixFromCSR refuses row_end < row_start like the reference,
 * src/internal/linsys.c:1220-1226.) */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_LinearSystemSetEmptyBlock(HYPREDRV_t hypredrv, HYPRE_BigInt row_start);
HYPREDRV_EXPORT_SYMBOL const char *HYPREDRV_AMD_LastErrorMessage(void);
/* measurement hook of bench.py: bytes this rank streams per Krylov iteration and per V-cycle,
 * [0] as CSR, [1] in the formats actually read */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_SolvePhaseBytes(HYPREDRV_t hypredrv, double iteration[2], double vcycle[2]);
/* measurement hook of bench.py (N > 1): arm != 0 brackets every Jacobi-sweep launch on this rank's largest
 * plain-CSR operator with HIP events (returns its level and local rows/cols/nnz); arm == 0 returns the
 * average launch duration and disarms */
HYPREDRV_EXPORT_SYMBOL uint32_t HYPREDRV_AMD_ProbeDominant(HYPREDRV_t hypredrv, int arm, int *level, double dims[3], double *avg_ms, int *count);

#ifdef __cplusplus
}
#endif
#endif
