/*
 * hypredrv_amd.h -- kernel-level C ABI of the MI355X solve path (libhypredrv_amd.so).
 *
 * This is the seam the parity tests drive: one entry point per hot-path operation, plain
 * pointers and sizes, host buffers in / host buffers out (the library stages them through
 * HBM).  Each entry names the hypre function it stands in for and the place the reference
 * reaches it (paths relative to the hypredrive tree).  The reference-facing boundary --
 * the HYPREDRV_ functions and the HYPRE_IJ / HYPRE_ParCSR / HYPRE_BoomerAMG subset -- is declared in
 * HYPREDRV.h / HYPRE*.h next to this file and is implemented on top of these.
 *
 * All functions return 0 on success; on failure a non-zero code is returned and
 * hda_last_error() describes it.  Nothing here falls back to the CPU: without a HIP device
 * every compute entry fails with HDA_ERR_NO_DEVICE.
 */
#ifndef HYPREDRV_AMD_H
#define HYPREDRV_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDA_OK 0
#define HDA_ERR_NO_DEVICE 1
#define HDA_ERR_RUNTIME 2
#define HDA_ERR_ARG 3

typedef struct hda_csr_s *hda_csr_t; /* CSR block resident in HBM (int32 idx, fp64) */
typedef struct hda_amg_s *hda_amg_t; /* AMG hierarchy resident in HBM */

/* AMG_args contract: include/internal/amg.h:108-123, defaults src/internal/amg.c:120-238
 * (HYPRE_USING_GPU branch). Same field order as the oracle's orc_amg_params. */
typedef struct {
   int      coarsen_type, interp_type, pmax;
   double   trunc_factor, strong_th, max_row_sum;
   int      max_coarse_size, min_coarse_size, max_levels;
   int      relax_down, relax_up, relax_coarse;
   int      sweeps_down, sweeps_up, sweeps_coarse;
   double   relax_weight, outer_weight;
   uint64_t seed;
   int      num_functions; /* coarsening.num_functions (1); > 1: unknown-based systems AMG, functions interleaved */
   /* relaxation.chebyshev for relax type 16 (reference src/internal/cheby.c:15-20): order 2, eig_est 10, variant 0, scale 1, fraction 0.3 */
   int      cheby_order, cheby_eig_est, cheby_variant, cheby_scale;
   double   cheby_fraction;
   /* complex smoother (src/internal/amg.c:899-921), ILU only: bj-iluk, fill 0, natural order on levels < smooth_num_levels */
   int      smooth_num_levels, smooth_num_sweeps;
   int      ilu_tri_solve, ilu_lower_it, ilu_upper_it; /* ILU_args tri_solve / lower_jac_iters / upper_jac_iters (ilu.c:21-23) */
   /* AMGagg_args (src/internal/amg.c:160-173, forwarded at :938-944): aggressive coarsening on the first agg_num_levels levels
    * (second PMIS pass over the graph of >= agg_num_paths paths of length <= 2), multipass interpolation (agg_interp_type 4) there */
   int      agg_num_levels, agg_num_paths, agg_interp_type;
   int      agg_pmax;         /* aggressive.max_nnz_row (0: no limit) */
   double   agg_trunc_factor; /* aggressive.trunc_factor */
   /* V contiguous row blocks on one GPU = the reference at np = V (its CPU defaults, src/internal/amg.c:141-146 HMIS and :182-189
    * hybrid l1 Gauss-Seidel, are rank-block algorithms): 1 one block, 0 chosen by the setup, V > 1 hypre's even split unless
    * block_part names the V + 1 row starts */
   int            blocks;
   const int64_t *block_part;
   int            struct_size; /* sizeof(hda_amg_params) as the caller was compiled; set by hda_amg_default_params, checked on entry */
} hda_amg_params;

/* PCG_args src/internal/pcg.c:15-25 / GMRES_args src/internal/gmres.c:16-27 */
typedef struct {
   int    max_iter;
   double rtol, atol;
   int    two_norm, krylov_dim;
} hda_krylov_params;

const char *hda_last_error(void);
int         hda_device_count(void);           /* 0 when no HIP device is visible */
int         hda_device_name(char *buf, int len);
/* PCI bus id of visible device dev (len >= 16): which physical GPU an index means (launchers that give every rank its own
 * HIP_VISIBLE_DEVICES make all indices 0; hypredrive_amd/dist.py decides "one GPU per rank" on this) */
int         hda_device_pci_bus_id(int dev, char *buf, int len);
int         hda_device_sync(void);
/* launches the empty kernel hda::k_marker on the library's stream: a boundary that shows in kernel traces and counter passes */
int         hda_marker(int id);

void hda_amg_default_params(hda_amg_params *p);           /* amg.c:120-238, GPU branch */
void hda_krylov_default_params(hda_krylov_params *p, int gmres);

/* ---- matrices -------------------------------------------------------------------- */
/* HYPRE_IJMatrixSetValues + Assemble for a local block (src/internal/linsys.c:1190-1405):
 * copies, converts to int32, column-sorts rows. */
int hda_csr_create(int nrows, int ncols, const int64_t *rowptr, const int64_t *cols,
                   const double *vals, hda_csr_t *out);
int hda_csr_destroy(hda_csr_t A);
int hda_csr_dims(hda_csr_t A, int *nrows, int *ncols, int *nnz);
int hda_csr_download(hda_csr_t A, int *rowptr, int *col, double *val);
/* 7-pt Laplacian of examples/src/C_laplacian/laplacian.c:719-921 generated in HBM for the
 * block (pc) of a P[0] x P[1] x P[2] partition; columns of other blocks are dropped when
 * keep_offproc == 0 (single-rank tests) -- rhs gets 1 on the global y = 0 plane. */
int hda_lap7_create(const int n[3], const int P[3], const int pc[3], const double c[3],
                    hda_csr_t *A, double *rhs_host /* may be NULL */);

/* ---- K1 / K2 / K9 ------------------------------------------------------------------ */
/* HYPRE_ParCSRMatrixMatvec (src/internal/linsys.c:3031): y = alpha*A*x + beta*y */
int hda_spmv(hda_csr_t A, double alpha, const double *x, double beta, double *y);
/* hypre_BoomerAMGRelax (types 0, 7, 18): `sweeps` sweeps of x += w*(b - A x)/d in place */
int hda_relax(hda_csr_t A, int relax_type, double weight, int sweeps, const double *b, double *x);
/* hypre_ParVectorInnerProd (src/internal/linsys.c:2875) */
int hda_dot(int n, const double *x, const double *y, double *result);
/* hypre_ParCSRComputeL1Norms option 1 / 4 */
int hda_l1_norms(hda_csr_t A, int option, double *l1);
/* the same three on nblk contiguous row blocks (part: nblk + 1 row starts) = what the reference computes at np = nblk: hybrid
 * Gauss-Seidel inside a block / Jacobi across blocks (hypre_BoomerAMGRelax types 3, 4, 6, 8, 13, 14), the option-4 l1 divisor with
 * the other blocks in the role of the off-processor part, and hypre_BoomerAMGCoarsenHMIS (Ruge first pass per block + PMIS) */
int hda_relax_blocks(hda_csr_t A, int relax_type, double weight, int sweeps, int nblk, const int64_t *part, const double *b, double *x);
int hda_l1_norms_blocks(hda_csr_t A, int option, int nblk, const int64_t *part, double *l1);
int hda_hmis_blocks(hda_csr_t A, const unsigned char *smask, int nblk, const int64_t *part, uint64_t seed, int level, int *cf);

/* ---- K4 / K5 / K6 ------------------------------------------------------------------ */
/* hypre_BoomerAMGCreateS */
int hda_strength(hda_csr_t A, double theta, double max_row_sum, unsigned char *smask);
/* hypre_BoomerAMGCoarsenPMIS (coarsen type 8) */
int hda_pmis(hda_csr_t A, const unsigned char *smask, uint64_t seed, int level,
             int64_t row_offset, int *cf);
/* hypre_BoomerAMGBuildExtPIInterp + hypre_BoomerAMGInterpTruncation */
int hda_interp_extpi(hda_csr_t A, const unsigned char *smask, const int *cf, int pmax,
                     double trunc_factor, hda_csr_t *P);
/* hypre_BoomerAMGBuildDirInterp with separation of weights (interpolation type 3, "direct_sep_weights",
 * reference src/internal/amg.c:258-270) + hypre_BoomerAMGInterpTruncation */
int hda_interp_direct(hda_csr_t A, const unsigned char *smask, const int *cf, int pmax,
                      double trunc_factor, hda_csr_t *P);
/* hypre's mm-ext+i (interpolation type 17, "mm-ext+i" in reference src/internal/amg.c:267-268; the interpolation of
 * examples/refOutput/ex8.txt:26-78): the matrix-matrix form of extended+i, W = -D^-1 (I + B) A^s_FC, + InterpTruncation */
int hda_interp_mm_extpi(hda_csr_t A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor, hda_csr_t *P);
/* hypre_BoomerAMGBuildStdInterp without separation of weights (interpolation type 8, "standard" in reference
 * src/internal/amg.c:258; the configuration examples/refOutput/ex8.txt:74 echoes for its fifth variant) + InterpTruncation */
int hda_interp_standard(hda_csr_t A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor, hda_csr_t *P);
/* hypre_BoomerAMGBuildCoarseOperator (P^T A P) */
int hda_rap(hda_csr_t A, hda_csr_t P, hda_csr_t *Ac);
/* aggressive coarsening, stage by stage (HYPRE_BoomerAMGSetAggNumLevels / SetNumPaths / SetAggInterpType 4; hypre_BoomerAMGCreate2ndS,
 * second PMIS pass, hypre_BoomerAMGBuildMultipass): S2 = strong connections of distance <= 2 among the C points of cf (values = number
 * of paths); the second pass updates cf in place; multipass P for a given splitting */
int hda_second_strength(hda_csr_t A, const unsigned char *smask, const int *cf, int num_paths, hda_csr_t *S2);
int hda_coarsen_second_pass(hda_csr_t A, const unsigned char *smask, int num_paths, uint64_t seed, int level, int *cf);
int hda_interp_multipass(hda_csr_t A, const unsigned char *smask, const int *cf, hda_csr_t *P);
/* hypre_BoomerAMGInterpTruncation on the finished rows of P, in place (HYPRE_BoomerAMGSetAggPMaxElmts / SetAggTruncFactor) */
int hda_truncate_rows(hda_csr_t P, int pmax, double trunc_factor);
int hda_transpose(hda_csr_t A, hda_csr_t *T);
int hda_spgemm(hda_csr_t X, hda_csr_t Y, hda_csr_t *C);

/* ---- hierarchy + V-cycle ----------------------------------------------------------- */
/* HYPRE_BoomerAMGCreate/Setup (src/internal/precon.c:107) */
int hda_amg_create(const hda_amg_params *p, hda_csr_t A, hda_amg_t *out);
/* systems AMG (params->num_functions > 1): dof_func[i] in [0, num_functions) names the function of
 * unknown i (reference src/internal/amg.c:792-862 hypredrv_AMGSetDofFunc); NULL = i mod num_functions */
int hda_amg_create_dof(const hda_amg_params *params, hda_csr_t A, const int *dof_func, hda_amg_t *out);
int hda_amg_destroy(hda_amg_t h);
/* "preconditioner: ilu" (reference src/internal/ilu.c:63-115; bj-iluk, fill 0, natural order): block-Jacobi ILU(0) of A's
 * diagonal block; the handle is accepted by hda_pcg / hda_gmres / hda_amg_vcycle (= one application from a zero guess)
 * in place of a hierarchy.  max_iter iterations x += M^-1 (b - A x) per application. */
int hda_ilu_create(hda_csr_t A, int max_iter, int tri_solve, int lower_it, int upper_it, hda_amg_t *out);
/* the same on V contiguous row blocks of one GPU = bj-iluk at np = V (hypre factors every rank's diagonal block on its own): entries
 * that leave a row's block are dropped, the exact substitutions (tri_solve 1) run block-parallel.  blocks: 1 one block, 0 chosen from
 * the operator's size and bandwidth, V > 1 with block_part = V + 1 row starts (NULL: hypre's even split, floor(q n / V)). */
int hda_ilu_create_blocks(hda_csr_t A, int max_iter, int tri_solve, int lower_it, int upper_it, int blocks, const int64_t *block_part,
                          hda_amg_t *out);
int hda_ilu_blocks(hda_amg_t h, int level); /* row blocks in use: level < 0 a handle of hda_ilu_create*, else that AMG level's smoother */
/* "preconditioner: mgr" (reference src/internal/mgr.c; MGRlvl_args include/internal/mgr.h:132-147): multigrid reduction
 * by dof labels with BoomerAMG on the coarsest system.  labels = dofmap of A's rows.  Implemented per level:
 * prolongation injection (0) / l1-jacobi (1) / jacobi (2); restriction injection (0) / jacobi (2) / columped (14);
 * f_relaxation jacobi (7) / l1-jacobi (18) / one BoomerAMG cycle (2) or ILU(0) solve (32) on A_FF; g_relaxation none (-1) / hybrid (l1) Gauss-Seidel (3, 4, 6, 13, 14, 88);
 * coarse grid by Galerkin product; coarsest system by BoomerAMG or ILU(0) iterations.  This test entry is single-rank (row partitions go through HYPRE_MGRSetup).  The handle is accepted by the Krylov entry points and hda_amg_vcycle. */
typedef struct {
   int        n_f_labels;
   const int *f_labels; /* level.N.f_dofs */
   int        interp_type, restrict_type;
   int        frelax_type, frelax_sweeps;
   int        grelax_type, grelax_sweeps;
   const hda_amg_params *frelax_amg; /* f_relaxation.amg block for frelax_type 2 (NULL: defaults) */
   int        ilu_tri_solve, ilu_lower_it, ilu_upper_it; /* ILU arguments of this level's ILU components (frelax 32, grelax 16) */
   /* coarsest_level: ilu -- read from the LAST level's entry when hda_mgr_create gets coarsest_amg == NULL */
   int        coarse_ilu_max_iter, coarse_ilu_tri_solve, coarse_ilu_lower_it, coarse_ilu_upper_it;
   /* nested Krylov components (reference src/internal/krylov.c; mgr.c:3938-3960, 4253-4275): 0 none, 1 pcg, 2 gmres, 3 fgmres,
    * 4 bicgstab.  f_relaxation: a solve of A_FF e = r_F from a zero guess, preconditioned by the level's amg / ilu component
    * (frelax_type 2 / 32) when frelax_krylov_precond is set.  coarsest_level: read from the LAST level's entry. */
   int               frelax_krylov, frelax_krylov_precond;
   hda_krylov_params frelax_kp;
   int               coarse_krylov, coarse_krylov_precond;
   hda_krylov_params coarse_kp;
   /* cycle shape, read from the LAST level's entry; 0 = default.  mgr_cycle: 1 V, 2 W; positions (reference mgr.c:614-675,
    * hypre's SetFRelaxCycle / SetGlobalSmoothCycle): 1 before the coarse correction, 2 after it, 3 both */
   int               mgr_cycle, mgr_frelax_pos, mgr_gsmooth_pos;
   /* row blocks of the hybrid Gauss-Seidel global relaxation (g_relaxation 3/4/6/13/14/88) = the reference at np = V (as hda_amg_params.blocks):
    * 1 = one block (the sequential sweep), V > 1 = hypre's even split of the level's rows, 0 = the setup's choice (one block up to 100 000 rows) */
   int               grelax_blocks;
} hda_mgr_level_params;
int hda_mgr_create(hda_csr_t A, const int *labels, int nlevels, const hda_mgr_level_params *levels,
                   const hda_amg_params *coarsest_amg, int max_iter, hda_amg_t *out);
/* borrowed view: which 0 = operator of the level (level == nlevels: the coarsest system), 1 = P, 2 = R */
int hda_mgr_matrix(hda_amg_t h, int level, int which, hda_csr_t *out);
/* borrowed view of the factors: strict lower part = L (unit diagonal), rest = U.  level < 0: the handle of
 * hda_ilu_create; level >= 0: the complex smoother of that AMG level. */
int hda_ilu_factors(hda_amg_t h, int level, hda_csr_t *out);
int hda_amg_num_levels(hda_amg_t h);
/* which: 0 = A_l, 1 = P_l, 2 = R_l; returns a borrowed handle (do not destroy) */
int hda_amg_level_matrix(hda_amg_t h, int level, int which, hda_csr_t *out);
int hda_amg_level_cf(hda_amg_t h, int level, int *cf);
/* row blocks the setup worked with (hda_amg_params.blocks resolved; 1 = none) and a level's block starts (blocks + 1 values) */
int hda_amg_blocks(hda_amg_t h);
int hda_amg_level_blocks(hda_amg_t h, int level, int64_t *part);
int hda_amg_complexities(hda_amg_t h, double *grid, double *op);
double hda_amg_vcycle_bytes(hda_amg_t h);
/* HYPRE_BoomerAMGSolve as preconditioner (src/internal/precon.c:108): x = V-cycle(b) from 0 */
int hda_amg_vcycle(hda_amg_t h, const double *b, double *x);

/* ---- Krylov ------------------------------------------------------------------------ */
/* HYPRE_ParCSRPCGSolve / HYPRE_ParCSRGMRESSolve (src/internal/solver.c:211,222). amg may be
 * NULL (no preconditioner). hist needs max_iter+1 doubles (may be NULL). x in/out. */
int hda_pcg(hda_csr_t A, hda_amg_t amg, const hda_krylov_params *kp, const double *b, double *x,
            double *hist, int *iters, int *converged, double *final_rel);
int hda_gmres(hda_csr_t A, hda_amg_t amg, const hda_krylov_params *kp, const double *b, double *x,
              double *hist, int *iters, int *converged, double *final_rel);
/* hypre_FlexGMRESSolve / hypre_BiCGSTABSolve (solver_ops[SOLVER_FGMRES / SOLVER_BICGSTAB], src/internal/solver.c:229-252) */
int hda_fgmres(hda_csr_t A, hda_amg_t amg, const hda_krylov_params *kp, const double *b, double *x,
              double *hist, int *iters, int *converged, double *final_rel);
int hda_bicgstab(hda_csr_t A, hda_amg_t amg, const hda_krylov_params *kp, const double *b, double *x,
              double *hist, int *iters, int *converged, double *final_rel);

/* ---- measurement (bench.py) -------------------------------------------------------- */
/* Times `reps` launches of one kernel with HIP events on the library's stream, vectors
 * resident in HBM.  kind: 0 SpMV, 1 l1-Jacobi sweep, 2 residual, 3 V-cycle (needs amg).
 * Returns average milliseconds per launch and the algorithmic bytes per launch. */
int hda_time_kernel(int kind, hda_csr_t A, hda_amg_t amg, int reps, double *avg_ms, double *bytes);
/* Device-resident Krylov solves of A x = b, x0 = 0, repeated `nsolves` times against an
 * existing hierarchy (amg may be NULL): the reference's solve loop
 * (examples/src/C_laplacian/laplacian.c:445-463) with the "solve" phase timed like its Stats
 * timer (src/internal/solver.c:668-683: initial and final true-residual evaluations are
 * outside the timed region).  b == NULL uses the generator's rhs stored with A.
 * solve_ms gets nsolves entries.  k1_avg_ms = average duration of the level-0 PCG SpMV
 * kernel measured with HIP events inside those solves (0 for GMRES). */
/* preconditioner applications the last hda_solve_device solve enqueued (PCG skips the one hypre
 * computes and discards after the final residual test) */
int hda_last_precond_calls(void);
int hda_solve_device(hda_csr_t A, hda_amg_t amg, const hda_krylov_params *kp, int solver,
                     const double *b_host, int nsolves, double *solve_ms, int *iters,
                     double *final_rel, double *r0_norm, double *true_rel, double *k1_avg_ms);
/* algorithmic HBM bytes of one PCG iteration without the preconditioner */
double hda_pcg_iteration_bytes(hda_csr_t A);
/* Bytes the kernels really stream when operators are held in the stencil-coded form (1 B per
 * entry + dictionary instead of 12 B; DESIGN.md "coded operators"): one PCG iteration, one
 * V-cycle, one plain product with A.  Equal to the CSR figures when nothing is coded.  *coded = storage form of A's products:
 * 0 plain CSR, 1 entry-coded stencil operator, 2 row-class coded, 3 windowed CSR, 4 value-coded, 5 value-coded + windowed. */
int hda_format_bytes(hda_csr_t A, hda_amg_t amg, double *pcg_iteration, double *vcycle, double *spmv, int *coded);
/* Timing probe: bracket every product launch of `mode` (0 y=Ax, 1 residual, 2 Jacobi sweep) on
 * matrix A (e.g. a view from hda_amg_level_matrix) with HIP events on the library stream;
 * hda_probe_read synchronises and returns the average launch duration.  A = NULL disarms. */
int hda_probe_spmv(hda_csr_t A, int mode);
int hda_probe_read(double *avg_ms, int *count);
/* Several probes at once (bench.py: dominant sweep, level-0 product, level-0 transfer operators): add returns the
 * probe's id; hda_probe_spmv(NULL, 0) clears them all. */
int hda_probe_add(hda_csr_t A, int mode, int *id);
int hda_probe_read_id(int id, double *avg_ms, int *count);
/* Borrowed seam views of a HYPREDRV_t whose solver is set up (include/HYPREDRV.h): its level-0 operator -- with the
 * device right-hand side and, on a row block, the ghost refresh plan -- and its BoomerAMG hierarchy, so that the
 * measurement entries above run on the very objects HYPREDRV_LinearSolverSetup built.  Valid until
 * HYPREDRV_LinearSolverDestroy; release the views with hda_csr_destroy / hda_amg_destroy. */
int hda_borrow_hypredrv(void *hypredrv, hda_csr_t *A, hda_amg_t *amg);
/* Host half of the library's halo plan (who owns every ghost column, what every peer wants from this rank), collective over
 * the communicator joined with HYPREDRV_AMD_CommInitCallbacks; touches no device, so the world_size-2/3 gloo tests drive the
 * library's own partition code on CPU.  part: world + 1 row starts; ghost_gids ascending. */
int hda_halo_plan_host(int nloc, const long long *part, const long long *ghost_gids, int nghost, int *send_counts,
                       int *recv_counts, int *send_idx, int send_cap, int *send_total);
/* rank-to-rank traffic of the solve path since the last reset: [0] device all-reduces, [1] halo exchanges (grouped
 * neighbour send/recv), [2] doubles all-reduced, [3] doubles sent in halo exchanges, [4] exchanges that ran under a
 * product kernel (interior rows computed while the ghost values travel) */
int hda_comm_stats(double out[5], int reset);
/* row-partitioned products: 1 = overlap the ghost refresh of x with the owned-column part (the default on RCCL), 0 = exchange first,
 * then the whole product (the default on host-staged transports), -1 = back to the default (transport, or HDA_OVERLAP).  bench.py
 * measures both on a first multi-GPU run, the serial form first. */
void hda_set_overlap(int mode);
const char *hda_comm_name(void); /* "self", "rccl", "host-callbacks", "threads" */
/* (the test seam "ranks as threads of one process" is declared in hypredrv_amd_testranks.h and built into its own library) */
/* levels of the set-up BoomerAMG hierarchy behind an HYPREDRV_t that are row partitioned (0: one rank, or not set up) */
int hda_amd_partitioned_levels(void *hypredrv);
int hda_amd_hierarchy_levels(void *hypredrv); /* all levels: the partitioned ones + those of the replicated tail */
int         hda_comm_size(void);
/* Exercises the active rank-to-rank transport (RCCL, staged callbacks or self): device
 * all-reduce, host all-reduce, host all-to-all.  Returns 0 when every result is right. */
int hda_comm_selftest(void);
/* allocator statistics (bytes) */
int hda_memory_stats(double *in_use, double *peak);
/* bytes of released device blocks the calling thread's allocator keeps for reuse: never more than twice the thread's peak in use
 * (or HDA_POOL_CACHE_MIN_GB, default 4, if that is larger); older blocks go back to the driver first */
double hda_memory_cached(void);
/* what the DRIVER's allocator cost since the last reset: out[0] hipMalloc calls that reached the driver (the caching allocator missed),
 * out[1] host milliseconds spent inside them, out[2] bytes they returned.  bench.py quotes it for the first (cold) and the second (warm)
 * setup of its process. */
int hda_memory_driver_stats(double out[3], int reset);
/* gives the calling thread's cached blocks back to the driver (after the library stream has drained): for a host application that
 * shares the device with other processes and is about to sit idle.  The allocator does this by itself when an allocation fails --
 * its own cache first, then the caches of the process's other rank threads -- but it cannot reach another PROCESS's cache. */
int hda_memory_trim(void);
/* Matrices are int32-indexed (hypre's HYPRE_Int in its default build, HYPRE_config.h).  Every setup
 * stage that sizes an operator (interpolation, sparse products, routed row blocks) checks the 64-bit
 * sum of its row lengths first and returns HDA_ERR with a message instead of wrapping around.  This
 * entry runs that check on nrows rows of row_len entries (test hook for the guard). */
int hda_check_row_total(long long nrows, int row_len);

#ifdef __cplusplus
}
#endif
#endif
