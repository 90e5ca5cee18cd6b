/*
 * HYPRE.h -- the slice of hypre's public C API that hypredrive's AMG-Krylov hot path binds
 * (SURVEY.md 8(b) "lower seam"), re-implemented on MI355X by libhypredrv_amd.so.
 *
 * Each group cites where the reference calls it.  Types follow a mixed-int hypre build:
 * HYPRE_Int = int32 (local indices), HYPRE_BigInt = int64 (global indices), HYPRE_Real =
 * HYPRE_Complex = double.  Handles are opaque; matrices and vectors live in HBM after
 * Assemble regardless of the memory location asked for (there is no host solve path).
 */
#ifndef HYPRE_AMD_HEADER
#define HYPRE_AMD_HEADER

#include "hda_mpi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HYPRE_RELEASE_NAME "hypre-subset (hypredrive_amd, MI355X)"
#define HYPRE_RELEASE_VERSION "3.0.0"
#define HYPRE_RELEASE_NUMBER 30000
#define HYPRE_DEVELOP_NUMBER 0
#define HYPRE_USING_GPU 1
#define HYPRE_USING_HIP 1
#define HYPRE_MIXEDINT 1

typedef int       HYPRE_Int;
typedef long long HYPRE_BigInt;
typedef double    HYPRE_Real;
typedef double    HYPRE_Complex;

typedef enum { HYPRE_MEMORY_UNDEFINED = -1, HYPRE_MEMORY_HOST = 0, HYPRE_MEMORY_DEVICE = 1 } HYPRE_MemoryLocation;
typedef enum { HYPRE_EXEC_UNDEFINED = -1, HYPRE_EXEC_HOST = 0, HYPRE_EXEC_DEVICE = 1 } HYPRE_ExecutionPolicy;

#define HYPRE_PARCSR 5555
#define HYPRE_ERROR_GENERIC 1
#define HYPRE_ERROR_MEMORY 2
#define HYPRE_ERROR_ARG 4
#define HYPRE_ERROR_CONV 256

struct hypre_IJMatrix_struct;
struct hypre_IJVector_struct;
struct hypre_Solver_struct;
typedef struct hypre_IJMatrix_struct *HYPRE_IJMatrix;
typedef struct hypre_IJVector_struct *HYPRE_IJVector;
typedef struct hypre_IJMatrix_struct *HYPRE_ParCSRMatrix; /* same object: IJ is a view */
typedef struct hypre_IJVector_struct *HYPRE_ParVector;
typedef struct hypre_IJMatrix_struct *HYPRE_Matrix;
typedef struct hypre_IJVector_struct *HYPRE_Vector;
typedef struct hypre_Solver_struct   *HYPRE_Solver;

typedef HYPRE_Int (*HYPRE_PtrToSolverFcn)(HYPRE_Solver, HYPRE_Matrix, HYPRE_Vector, HYPRE_Vector);
typedef HYPRE_Int (*HYPRE_PtrToParSolverFcn)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector);

/* ---- utilities (src/internal/runtime.c:101-120, src/HYPREDRV.c:316-346) ---- */
HYPRE_Int HYPRE_Initialize(void);
HYPRE_Int HYPRE_Finalize(void);
HYPRE_Int HYPRE_SetMemoryLocation(HYPRE_MemoryLocation loc);
HYPRE_Int HYPRE_SetExecutionPolicy(HYPRE_ExecutionPolicy pol);
HYPRE_Int HYPRE_GetError(void);
HYPRE_Int HYPRE_ClearAllErrors(void);
HYPRE_Int HYPRE_CheckError(HYPRE_Int ierr, HYPRE_Int code);

/* ---- IJ matrix (examples/src/C_laplacian/laplacian.c:734-914, src/internal/linsys.c:1281-1380) ---- */
HYPRE_Int HYPRE_IJMatrixCreate(MPI_Comm comm, HYPRE_BigInt ilower, HYPRE_BigInt iupper, HYPRE_BigInt jlower,
                               HYPRE_BigInt jupper, HYPRE_IJMatrix *matrix);
HYPRE_Int HYPRE_IJMatrixDestroy(HYPRE_IJMatrix matrix);
HYPRE_Int HYPRE_IJMatrixSetObjectType(HYPRE_IJMatrix matrix, HYPRE_Int type);
HYPRE_Int HYPRE_IJMatrixSetRowSizes(HYPRE_IJMatrix matrix, const HYPRE_Int *sizes);
HYPRE_Int HYPRE_IJMatrixSetDiagOffdSizes(HYPRE_IJMatrix matrix, const HYPRE_Int *diag, const HYPRE_Int *offd);
HYPRE_Int HYPRE_IJMatrixInitialize(HYPRE_IJMatrix matrix);
HYPRE_Int HYPRE_IJMatrixInitialize_v2(HYPRE_IJMatrix matrix, HYPRE_MemoryLocation loc);
HYPRE_Int HYPRE_IJMatrixSetValues(HYPRE_IJMatrix matrix, HYPRE_Int nrows, HYPRE_Int *ncols, const HYPRE_BigInt *rows,
                                  const HYPRE_BigInt *cols, const HYPRE_Complex *values);
HYPRE_Int HYPRE_IJMatrixAddToValues(HYPRE_IJMatrix matrix, HYPRE_Int nrows, HYPRE_Int *ncols, const HYPRE_BigInt *rows,
                                    const HYPRE_BigInt *cols, const HYPRE_Complex *values);
HYPRE_Int HYPRE_IJMatrixAssemble(HYPRE_IJMatrix matrix);
HYPRE_Int HYPRE_IJMatrixGetObject(HYPRE_IJMatrix matrix, void **object);
HYPRE_Int HYPRE_IJMatrixGetLocalRange(HYPRE_IJMatrix matrix, HYPRE_BigInt *ilower, HYPRE_BigInt *iupper,
                                      HYPRE_BigInt *jlower, HYPRE_BigInt *jupper);
HYPRE_Int HYPRE_IJMatrixMigrate(HYPRE_IJMatrix matrix, HYPRE_MemoryLocation loc);
HYPRE_Int HYPRE_IJMatrixRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJMatrix *matrix);
/* Matrix Market coordinate file (reference src/internal/linsys.c:986, linear_system.type mtx) */
HYPRE_Int HYPRE_IJMatrixReadMM(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJMatrix *matrix);
/* hypredrive's multipart binary containers "<prefix>.<part %05d>.bin" (reference
 * src/internal/matrix.c:142 hypredrv_IJMatrixReadMultipartBinary, src/internal/vector.c:92
 * hypredrv_IJVectorReadMultipartBinary; part count as hypredrv_CountNumberOfPartitions) */
int       hda_count_binary_parts(const char *prefix);
HYPRE_Int hda_IJMatrixReadMultipartBinary(const char *prefix, MPI_Comm comm, long long g_nparts, HYPRE_IJMatrix *matrix);
HYPRE_Int hda_IJVectorReadMultipartBinary(const char *prefix, MPI_Comm comm, long long g_nparts, HYPRE_IJVector *vector);
HYPRE_Int HYPRE_IJMatrixPrint(HYPRE_IJMatrix matrix, const char *filename);
/* global rows / nonzeros (what linsys.c reads through hypre_ParCSRMatrix accessors) */
HYPRE_Int HYPRE_ParCSRMatrixGetDims(HYPRE_ParCSRMatrix A, HYPRE_BigInt *M, HYPRE_BigInt *N);
HYPRE_Int HYPRE_ParCSRMatrixGetNumNonzeros(HYPRE_ParCSRMatrix A, HYPRE_BigInt *nnz);

/* ---- IJ vector ---- */
HYPRE_Int HYPRE_IJVectorCreate(MPI_Comm comm, HYPRE_BigInt jlower, HYPRE_BigInt jupper, HYPRE_IJVector *vector);
HYPRE_Int HYPRE_IJVectorDestroy(HYPRE_IJVector vector);
HYPRE_Int HYPRE_IJVectorSetObjectType(HYPRE_IJVector vector, HYPRE_Int type);
HYPRE_Int HYPRE_IJVectorInitialize(HYPRE_IJVector vector);
HYPRE_Int HYPRE_IJVectorInitialize_v2(HYPRE_IJVector vector, HYPRE_MemoryLocation loc);
HYPRE_Int HYPRE_IJVectorSetValues(HYPRE_IJVector vector, HYPRE_Int nvalues, const HYPRE_BigInt *indices,
                                  const HYPRE_Complex *values);
HYPRE_Int HYPRE_IJVectorAddToValues(HYPRE_IJVector vector, HYPRE_Int nvalues, const HYPRE_BigInt *indices,
                                    const HYPRE_Complex *values);
HYPRE_Int HYPRE_IJVectorGetValues(HYPRE_IJVector vector, HYPRE_Int nvalues, const HYPRE_BigInt *indices,
                                  HYPRE_Complex *values);
HYPRE_Int HYPRE_IJVectorAssemble(HYPRE_IJVector vector);
HYPRE_Int HYPRE_IJVectorGetObject(HYPRE_IJVector vector, void **object);
HYPRE_Int HYPRE_IJVectorGetLocalRange(HYPRE_IJVector vector, HYPRE_BigInt *jlower, HYPRE_BigInt *jupper);
HYPRE_Int HYPRE_IJVectorMigrate(HYPRE_IJVector vector, HYPRE_MemoryLocation loc);
HYPRE_Int HYPRE_IJVectorRead(const char *filename, MPI_Comm comm, HYPRE_Int type, HYPRE_IJVector *vector);
HYPRE_Int HYPRE_IJVectorPrint(HYPRE_IJVector vector, const char *filename);

/* ---- ParCSR kernels (src/internal/linsys.c:2875,2964-2967,3030-3032) ---- */
HYPRE_Int HYPRE_ParCSRMatrixMatvec(HYPRE_Complex alpha, HYPRE_ParCSRMatrix A, HYPRE_ParVector x, HYPRE_Complex beta,
                                   HYPRE_ParVector y);
HYPRE_Int HYPRE_ParVectorInnerProd(HYPRE_ParVector x, HYPRE_ParVector y, HYPRE_Real *prod);
HYPRE_Int HYPRE_IJVectorInnerProd(HYPRE_IJVector x, HYPRE_IJVector y, HYPRE_Real *prod); /* lidcavity.c:1404 */
HYPRE_Int HYPRE_ParVectorCopy(HYPRE_ParVector x, HYPRE_ParVector y);
HYPRE_Int HYPRE_ParVectorScale(HYPRE_Complex value, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParVectorAxpy(HYPRE_Complex alpha, HYPRE_ParVector x, HYPRE_ParVector y);
HYPRE_Int HYPRE_ParVectorSetConstantValues(HYPRE_ParVector vector, HYPRE_Complex value);

/* ---- PCG (solver_ops[SOLVER_PCG], src/internal/solver.c:204-216; setters src/internal/pcg.c:59-69) ---- */
HYPRE_Int HYPRE_ParCSRPCGCreate(MPI_Comm comm, HYPRE_Solver *solver);
HYPRE_Int HYPRE_ParCSRPCGDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ParCSRPCGSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRPCGSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_PCGSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_PCGSetTwoNorm(HYPRE_Solver solver, HYPRE_Int two_norm);
HYPRE_Int HYPRE_PCGSetStopCrit(HYPRE_Solver solver, HYPRE_Int stop_crit);
HYPRE_Int HYPRE_PCGSetRelChange(HYPRE_Solver solver, HYPRE_Int rel_change);
HYPRE_Int HYPRE_PCGSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_PCGSetLogging(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_PCGSetRecomputeResidual(HYPRE_Solver solver, HYPRE_Int recompute);
HYPRE_Int HYPRE_PCGSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_PCGSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_PCGSetResidualTol(HYPRE_Solver solver, HYPRE_Real rtol);
HYPRE_Int HYPRE_PCGSetConvergenceFactorTol(HYPRE_Solver solver, HYPRE_Real cf_tol);
HYPRE_Int HYPRE_PCGSetPrecond(HYPRE_Solver solver, HYPRE_PtrToSolverFcn precond, HYPRE_PtrToSolverFcn precond_setup,
                              HYPRE_Solver precond_solver);
HYPRE_Int HYPRE_PCGGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_PCGGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int HYPRE_PCGGetConverged(HYPRE_Solver solver, HYPRE_Int *converged);

/* ---- GMRES (solver_ops[SOLVER_GMRES], src/internal/solver.c:217-228; src/internal/gmres.c:63-74) ---- */
HYPRE_Int HYPRE_ParCSRGMRESCreate(MPI_Comm comm, HYPRE_Solver *solver);
HYPRE_Int HYPRE_ParCSRGMRESDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ParCSRGMRESSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRGMRESSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_GMRESSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_GMRESSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_GMRESSetStopCrit(HYPRE_Solver solver, HYPRE_Int stop_crit);
HYPRE_Int HYPRE_GMRESSetSkipRealResidualCheck(HYPRE_Solver solver, HYPRE_Int skip);
HYPRE_Int HYPRE_GMRESSetKDim(HYPRE_Solver solver, HYPRE_Int k_dim);
HYPRE_Int HYPRE_GMRESSetRelChange(HYPRE_Solver solver, HYPRE_Int rel_change);
HYPRE_Int HYPRE_GMRESSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_GMRESSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_GMRESSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_GMRESSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_GMRESSetConvergenceFactorTol(HYPRE_Solver solver, HYPRE_Real cf_tol);
HYPRE_Int HYPRE_GMRESSetPrecond(HYPRE_Solver solver, HYPRE_PtrToSolverFcn precond, HYPRE_PtrToSolverFcn precond_setup,
                                HYPRE_Solver precond_solver);
HYPRE_Int HYPRE_GMRESGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_GMRESGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int HYPRE_GMRESGetConverged(HYPRE_Solver solver, HYPRE_Int *converged);

/* ---- FlexGMRES (solver_ops[SOLVER_FGMRES], src/internal/solver.c:229-240; src/internal/fgmres.c:36-48) ---- */
HYPRE_Int HYPRE_ParCSRFlexGMRESCreate(MPI_Comm comm, HYPRE_Solver *solver);
HYPRE_Int HYPRE_ParCSRFlexGMRESDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ParCSRFlexGMRESSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRFlexGMRESSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_FlexGMRESSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_FlexGMRESSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_FlexGMRESSetKDim(HYPRE_Solver solver, HYPRE_Int k_dim);
HYPRE_Int HYPRE_FlexGMRESSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_FlexGMRESSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_FlexGMRESSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_FlexGMRESSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_FlexGMRESSetConvergenceFactorTol(HYPRE_Solver solver, HYPRE_Real cf_tol);
HYPRE_Int HYPRE_FlexGMRESSetPrecond(HYPRE_Solver solver, HYPRE_PtrToSolverFcn precond, HYPRE_PtrToSolverFcn precond_setup,
                                    HYPRE_Solver precond_solver);
HYPRE_Int HYPRE_FlexGMRESGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_FlexGMRESGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int HYPRE_FlexGMRESGetConverged(HYPRE_Solver solver, HYPRE_Int *converged);

/* ---- BiCGSTAB (solver_ops[SOLVER_BICGSTAB], src/internal/solver.c:241-252; src/internal/bicgstab.c:41-55) ---- */
HYPRE_Int HYPRE_ParCSRBiCGSTABCreate(MPI_Comm comm, HYPRE_Solver *solver);
HYPRE_Int HYPRE_ParCSRBiCGSTABDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ParCSRBiCGSTABSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ParCSRBiCGSTABSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_BiCGSTABSetMinIter(HYPRE_Solver solver, HYPRE_Int min_iter);
HYPRE_Int HYPRE_BiCGSTABSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_BiCGSTABSetStopCrit(HYPRE_Solver solver, HYPRE_Int stop_crit);
HYPRE_Int HYPRE_BiCGSTABSetLogging(HYPRE_Solver solver, HYPRE_Int logging);
HYPRE_Int HYPRE_BiCGSTABSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_BiCGSTABSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_BiCGSTABSetAbsoluteTol(HYPRE_Solver solver, HYPRE_Real a_tol);
HYPRE_Int HYPRE_BiCGSTABSetConvergenceFactorTol(HYPRE_Solver solver, HYPRE_Real cf_tol);
HYPRE_Int HYPRE_BiCGSTABSetPrecond(HYPRE_Solver solver, HYPRE_PtrToSolverFcn precond, HYPRE_PtrToSolverFcn precond_setup,
                                   HYPRE_Solver precond_solver);
HYPRE_Int HYPRE_BiCGSTABGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_BiCGSTABGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
HYPRE_Int hypre_BiCGSTABGetConverged(HYPRE_Solver solver, HYPRE_Int *converged); /* hypre-internal, used by solver.c:198-202 */

/* ---- BoomerAMG (precon_ops[PRECON_BOOMERAMG], src/internal/precon.c:106-109; setter sequence
 * src/internal/amg.c:868-1032).  Setters for features this build does not implement store the
 * value and make Setup fail loudly if the value selects the unimplemented feature. ---- */
HYPRE_Int HYPRE_BoomerAMGCreate(HYPRE_Solver *solver);
HYPRE_Int HYPRE_BoomerAMGDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_BoomerAMGSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_BoomerAMGSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_BoomerAMGSetInterpType(HYPRE_Solver solver, HYPRE_Int interp_type);
HYPRE_Int HYPRE_BoomerAMGSetRestriction(HYPRE_Solver solver, HYPRE_Int restr_par);
HYPRE_Int HYPRE_BoomerAMGSetStrongThresholdR(HYPRE_Solver solver, HYPRE_Real th);
HYPRE_Int HYPRE_BoomerAMGSetFilterThresholdR(HYPRE_Solver solver, HYPRE_Real th);
HYPRE_Int HYPRE_BoomerAMGSetCoarsenType(HYPRE_Solver solver, HYPRE_Int coarsen_type);
HYPRE_Int HYPRE_BoomerAMGSetSabs(HYPRE_Solver solver, HYPRE_Int sabs);
HYPRE_Int HYPRE_BoomerAMGSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_BoomerAMGSetStrongThreshold(HYPRE_Solver solver, HYPRE_Real th);
HYPRE_Int HYPRE_BoomerAMGSetSeqThreshold(HYPRE_Solver solver, HYPRE_Int th);
HYPRE_Int HYPRE_BoomerAMGSetMaxCoarseSize(HYPRE_Solver solver, HYPRE_Int size);
HYPRE_Int HYPRE_BoomerAMGSetMinCoarseSize(HYPRE_Solver solver, HYPRE_Int size);
HYPRE_Int HYPRE_BoomerAMGSetTruncFactor(HYPRE_Solver solver, HYPRE_Real trunc_factor);
HYPRE_Int HYPRE_BoomerAMGSetPMaxElmts(HYPRE_Solver solver, HYPRE_Int pmax);
HYPRE_Int HYPRE_BoomerAMGSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_BoomerAMGSetLogging(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_BoomerAMGSetRelaxType(HYPRE_Solver solver, HYPRE_Int relax_type);
HYPRE_Int HYPRE_BoomerAMGSetCycleRelaxType(HYPRE_Solver solver, HYPRE_Int relax_type, HYPRE_Int k);
HYPRE_Int HYPRE_BoomerAMGSetRelaxOrder(HYPRE_Solver solver, HYPRE_Int order);
HYPRE_Int HYPRE_BoomerAMGSetRelaxWt(HYPRE_Solver solver, HYPRE_Real wt);
HYPRE_Int HYPRE_BoomerAMGSetOuterWt(HYPRE_Solver solver, HYPRE_Real wt);
HYPRE_Int HYPRE_BoomerAMGSetNumSweeps(HYPRE_Solver solver, HYPRE_Int num_sweeps);
HYPRE_Int HYPRE_BoomerAMGSetCycleNumSweeps(HYPRE_Solver solver, HYPRE_Int num_sweeps, HYPRE_Int k);
HYPRE_Int HYPRE_BoomerAMGSetCycleType(HYPRE_Solver solver, HYPRE_Int cycle_type);
HYPRE_Int HYPRE_BoomerAMGSetMaxLevels(HYPRE_Solver solver, HYPRE_Int max_levels);
HYPRE_Int HYPRE_BoomerAMGSetMaxIter(HYPRE_Solver solver, HYPRE_Int max_iter);
HYPRE_Int HYPRE_BoomerAMGSetMaxRowSum(HYPRE_Solver solver, HYPRE_Real max_row_sum);
HYPRE_Int HYPRE_BoomerAMGSetNumFunctions(HYPRE_Solver solver, HYPRE_Int num_functions);
/* function of every locally owned unknown (copied at Setup; the caller keeps the array) */
HYPRE_Int HYPRE_BoomerAMGSetDofFunc(HYPRE_Solver solver, HYPRE_Int *dof_func);
HYPRE_Int HYPRE_BoomerAMGSetFilterFunctions(HYPRE_Solver solver, HYPRE_Int filter);
HYPRE_Int HYPRE_BoomerAMGSetSmoothType(HYPRE_Solver solver, HYPRE_Int type);
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumSweeps(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetSmoothNumLevels(HYPRE_Solver solver, HYPRE_Int n);
/* ILU arguments of the complex smoother (reference src/internal/amg.c:903-921) */
HYPRE_Int HYPRE_BoomerAMGSetILUType(HYPRE_Solver solver, HYPRE_Int type);
HYPRE_Int HYPRE_BoomerAMGSetILULevel(HYPRE_Solver solver, HYPRE_Int fill);
HYPRE_Int HYPRE_BoomerAMGSetILULocalReordering(HYPRE_Solver solver, HYPRE_Int reordering);
HYPRE_Int HYPRE_BoomerAMGSetILUTriSolve(HYPRE_Solver solver, HYPRE_Int tri_solve);
HYPRE_Int HYPRE_BoomerAMGSetILULowerJacobiIters(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetILUUpperJacobiIters(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetILUDroptol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_BoomerAMGSetILUMaxRowNnz(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetILUMaxIter(HYPRE_Solver solver, HYPRE_Int n);

/* ---- MGR (reference src/internal/mgr.c:3782-3808 and the per-level arrays after :3820).  Implemented on MI355X: C points by
 * dof label, prolongation injection / jacobi / l1-jacobi, restriction injection / jacobi / columped, Galerkin coarse grids,
 * Jacobi / l1-Jacobi / BoomerAMG- or ILU-on-A_FF F-relaxation, hybrid (l1) Gauss-Seidel or ILU(0) global relaxation, BoomerAMG or ILU coarse solver, V-cycle, one rank or a row partition.
 * Anything else is rejected at Setup. */
HYPRE_Int HYPRE_MGRCreate(HYPRE_Solver *solver);
HYPRE_Int HYPRE_MGRDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_MGRSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_MGRSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_MGRSetCpointsByPointMarkerArray(HYPRE_Solver solver, HYPRE_Int block_size, HYPRE_Int max_num_levels,
                                                HYPRE_Int *num_block_coarse_points, HYPRE_Int **block_coarse_indexes,
                                                HYPRE_Int *point_marker_array);
HYPRE_Int HYPRE_MGRSetNonCpointsToFpoints(HYPRE_Solver solver, HYPRE_Int flag);
HYPRE_Int HYPRE_MGRSetPMaxElmts(HYPRE_Solver solver, HYPRE_Int pmax);
HYPRE_Int HYPRE_MGRSetNonGalerkinMaxElmts(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_MGRSetMaxIter(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_MGRSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_MGRSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_MGRSetCycleType(HYPRE_Solver solver, HYPRE_Int cycle);
HYPRE_Int HYPRE_MGRSetFRelaxCycle(HYPRE_Solver solver, HYPRE_Int pos);
HYPRE_Int HYPRE_MGRSetGlobalSmoothCycle(HYPRE_Solver solver, HYPRE_Int pos);
HYPRE_Int HYPRE_MGRSetTruncateCoarseGridThreshold(HYPRE_Solver solver, HYPRE_Real th);
HYPRE_Int HYPRE_MGRSetRelaxType(HYPRE_Solver solver, HYPRE_Int type);
HYPRE_Int HYPRE_MGRSetLevelFRelaxType(HYPRE_Solver solver, HYPRE_Int *types);
HYPRE_Int HYPRE_MGRSetLevelNumRelaxSweeps(HYPRE_Solver solver, HYPRE_Int *sweeps);
HYPRE_Int HYPRE_MGRSetLevelInterpType(HYPRE_Solver solver, HYPRE_Int *types);
HYPRE_Int HYPRE_MGRSetLevelRestrictType(HYPRE_Solver solver, HYPRE_Int *types);
HYPRE_Int HYPRE_MGRSetCoarseGridMethod(HYPRE_Solver solver, HYPRE_Int *methods);
HYPRE_Int HYPRE_MGRSetLevelSmoothType(HYPRE_Solver solver, HYPRE_Int *types);
HYPRE_Int HYPRE_MGRSetLevelSmoothIters(HYPRE_Solver solver, HYPRE_Int *iters);
HYPRE_Int HYPRE_MGRSetFSolverAtLevel(HYPRE_Solver solver, HYPRE_Solver fsolver, HYPRE_Int level);       /* a BoomerAMG or ILU handle for A_FF */
HYPRE_Int HYPRE_MGRSetGlobalSmootherAtLevel(HYPRE_Solver solver, HYPRE_Solver smoother, HYPRE_Int level); /* an ILU handle */
HYPRE_Int HYPRE_MGRSetCoarseSolver(HYPRE_Solver solver, HYPRE_PtrToSolverFcn solve, HYPRE_PtrToSolverFcn setup, HYPRE_Solver coarse_solver);
HYPRE_Int HYPRE_MGRGetNumIterations(HYPRE_Solver solver, HYPRE_Int *n);
HYPRE_Int HYPRE_MGRGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *r);

/* ---- ILU (reference src/internal/ilu.c:63-115; op table src/internal/precon.c).  Implemented on
 * MI355X: type 0 (bj-iluk), fill level 0, no local reordering, exact or Jacobi-iterative triangular
 * solves; every other variant is rejected at Setup. */
HYPRE_Int HYPRE_ILUCreate(HYPRE_Solver *solver);
HYPRE_Int HYPRE_ILUDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ILUSetup(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ILUSolve(HYPRE_Solver solver, HYPRE_ParCSRMatrix A, HYPRE_ParVector b, HYPRE_ParVector x);
HYPRE_Int HYPRE_ILUSetType(HYPRE_Solver solver, HYPRE_Int type);
HYPRE_Int HYPRE_ILUSetLevelOfFill(HYPRE_Solver solver, HYPRE_Int fill);
HYPRE_Int HYPRE_ILUSetLocalReordering(HYPRE_Solver solver, HYPRE_Int reordering);
HYPRE_Int HYPRE_ILUSetTriSolve(HYPRE_Solver solver, HYPRE_Int tri_solve);
HYPRE_Int HYPRE_ILUSetLowerJacobiIters(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_ILUSetUpperJacobiIters(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_ILUSetPrintLevel(HYPRE_Solver solver, HYPRE_Int level);
HYPRE_Int HYPRE_ILUSetMaxIter(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_ILUSetTol(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_ILUSetMaxNnzPerRow(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_ILUSetDropThreshold(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_ILUSetSchurMaxIter(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_ILUSetNSHDropThreshold(HYPRE_Solver solver, HYPRE_Real tol);
HYPRE_Int HYPRE_ILUGetNumIterations(HYPRE_Solver solver, HYPRE_Int *n);
HYPRE_Int HYPRE_ILUGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *r);
HYPRE_Int HYPRE_BoomerAMGSetAggNumLevels(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetAggInterpType(HYPRE_Solver solver, HYPRE_Int t);
HYPRE_Int HYPRE_BoomerAMGSetAggTruncFactor(HYPRE_Solver solver, HYPRE_Real f);
HYPRE_Int HYPRE_BoomerAMGSetAggP12TruncFactor(HYPRE_Solver solver, HYPRE_Real f);
HYPRE_Int HYPRE_BoomerAMGSetAggPMaxElmts(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetAggP12MaxElmts(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetNumPaths(HYPRE_Solver solver, HYPRE_Int n);
HYPRE_Int HYPRE_BoomerAMGSetRAP2(HYPRE_Solver solver, HYPRE_Int rap2);
HYPRE_Int HYPRE_BoomerAMGSetModuleRAP2(HYPRE_Solver solver, HYPRE_Int mod_rap2);
HYPRE_Int HYPRE_BoomerAMGSetKeepTranspose(HYPRE_Solver solver, HYPRE_Int keep);
HYPRE_Int HYPRE_BoomerAMGSetChebyOrder(HYPRE_Solver solver, HYPRE_Int order);
HYPRE_Int HYPRE_BoomerAMGSetChebyFraction(HYPRE_Solver solver, HYPRE_Real ratio);
HYPRE_Int HYPRE_BoomerAMGSetChebyEigEst(HYPRE_Solver solver, HYPRE_Int eig_est);
HYPRE_Int HYPRE_BoomerAMGSetChebyVariant(HYPRE_Solver solver, HYPRE_Int variant);
HYPRE_Int HYPRE_BoomerAMGSetChebyScale(HYPRE_Solver solver, HYPRE_Int scale);
HYPRE_Int HYPRE_BoomerAMGGetNumIterations(HYPRE_Solver solver, HYPRE_Int *num_iterations);
HYPRE_Int HYPRE_BoomerAMGGetFinalRelativeResidualNorm(HYPRE_Solver solver, HYPRE_Real *norm);
/* hierarchy facts the stats/log print (num levels, complexities) */
HYPRE_Int HYPRE_BoomerAMGGetNumLevels(HYPRE_Solver solver, HYPRE_Int *num_levels);
HYPRE_Int HYPRE_BoomerAMGGetComplexities(HYPRE_Solver solver, HYPRE_Real *grid, HYPRE_Real *op);

/* ---- the rest of the names the reference files of SURVEY 8(a) call (round 5), so that an unmodified libHYPREDRV links as it is.
 * FSAI parameter setters (src/internal/amg.c:924-932 calls them for every BoomerAMG): accepted, nothing recorded -- choosing the FSAI
 * smoother itself (HYPRE_BoomerAMGSetSmoothType 4) is refused by HYPRE_BoomerAMGSetup. */
HYPRE_Int HYPRE_BoomerAMGSetFSAIAlgoType(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_BoomerAMGSetFSAILocalSolveType(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_BoomerAMGSetFSAIMaxSteps(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_BoomerAMGSetFSAIMaxStepSize(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_BoomerAMGSetFSAIMaxNnzRow(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_BoomerAMGSetFSAINumLevels(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_BoomerAMGSetFSAIThreshold(HYPRE_Solver solver, HYPRE_Real v);
HYPRE_Int HYPRE_BoomerAMGSetFSAIEigMaxIters(HYPRE_Solver solver, HYPRE_Int v);
HYPRE_Int HYPRE_BoomerAMGSetFSAIKapTolerance(HYPRE_Solver solver, HYPRE_Real v);
/* REFUSED BY NAME (outside SURVEY 8): each sets hypre's error flag with a message and returns non-zero.  amg.c:988-1032
 * (relaxation.points, coarsening.nodal + rigid-body modes), gmres.c:86-98 (reference-solution tracking), precon.c:138-154 (destroy
 * entries of preconditioners this library never creates: NULL is accepted). */
HYPRE_Int HYPRE_BoomerAMGSetGridRelaxPoints(HYPRE_Solver solver, HYPRE_Int **grid_relax_points);
HYPRE_Int HYPRE_BoomerAMGSetNodal(HYPRE_Solver solver, HYPRE_Int nodal);
HYPRE_Int HYPRE_BoomerAMGSetNodalDiag(HYPRE_Solver solver, HYPRE_Int nodal_diag);
HYPRE_Int HYPRE_BoomerAMGSetInterpVecVariant(HYPRE_Solver solver, HYPRE_Int var);
HYPRE_Int HYPRE_BoomerAMGSetInterpVecQMax(HYPRE_Solver solver, HYPRE_Int q_max);
HYPRE_Int HYPRE_BoomerAMGSetSmoothInterpVectors(HYPRE_Solver solver, HYPRE_Int smooth);
HYPRE_Int HYPRE_BoomerAMGSetInterpVectors(HYPRE_Solver solver, HYPRE_Int num_vectors, HYPRE_ParVector *interp_vectors);
HYPRE_Int HYPRE_ParCSRGMRESSetRefSolution(HYPRE_Solver solver, HYPRE_ParVector xref);
HYPRE_Int HYPRE_FSAIDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_AMSDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_ADSDestroy(HYPRE_Solver solver);
HYPRE_Int HYPRE_SchwarzDestroy(HYPRE_Solver solver);
/* amg.c:557, precon.c:770-783: a ParVector is this library's IJ vector; partitioning = {first row, one past the last} of the
 * calling rank, NULL = hypre's even split */
HYPRE_Int HYPRE_ParVectorCreate(MPI_Comm comm, HYPRE_BigInt global_size, HYPRE_BigInt *partitioning, HYPRE_ParVector *vector);
HYPRE_Int HYPRE_ParVectorInitialize(HYPRE_ParVector vector);
HYPRE_Int HYPRE_ParVectorDestroy(HYPRE_ParVector vector);

#ifdef __cplusplus
}
#endif
#endif
