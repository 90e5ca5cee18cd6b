/*
 * amg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the algorithms that the reference (hypredrive) reaches
 * through its op tables src/internal/solver.c:204-253 and src/internal/precon.c:106-157,
 * i.e. hypre's ParCSR PCG / GMRES + BoomerAMG V-cycle.  hypre itself is a
 * third-party dependency that is NOT vendored under /root/reference and is not
 * version pinned (cmake/HYPREDRV_Deps.cmake:1048, HYPRE_VERSION "master"; refOutputs
 * were made with a v3.0.0 build, examples/refOutput/laplacian.txt:42), so this file
 * restates the published algorithms (Ruge-Stueben strength, PMIS/HMIS coarsening,
 * extended+i interpolation, Galerkin RAP, l1 smoothers, PCG) and is pinned only by
 * the reference's own checked-in outputs (examples/refOutput/{ex1,ex2,laplacian}.txt)
 * and unit-test known answers -- see tests/test_oracle_pins.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef AMG_ORACLE_H
#define AMG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
   int     nrows, ncols;
   int    *rowptr; /* nrows+1 */
   int    *col;    /* nnz, ascending inside each row */
   double *val;    /* nnz */
} orc_csr;

/* Parameter contract = reference AMG_args (include/internal/amg.h:108-123) with the
 * defaults of src/internal/amg.c:120-238. */
typedef struct {
   int    coarsen_type;    /* 8 PMIS (hypre-GPU default), 10 HMIS, 6 Falgout->RS pass */
   int    interp_type;     /* 6 extended+i, 17 mm-ext+i (its matrix-matrix form: a different operator), 3 direct with separation of weights */
   int    pmax;            /* interpolation.max_nnz_row = 4 */
   double trunc_factor;    /* 0.0 */
   double strong_th;       /* 0.25 */
   double max_row_sum;     /* 0.9 */
   int    max_coarse_size; /* 64 */
   int    min_coarse_size; /* 0 */
   int    max_levels;      /* 25 */
   int    relax_down;      /* 18 (GPU default) or 13 */
   int    relax_up;        /* 18 or 14 */
   int    relax_coarse;    /* 9 */
   int    sweeps_down;     /* 1 */
   int    sweeps_up;       /* 1 */
   int    sweeps_coarse;   /* 1 */
   double relax_weight;    /* 1.0 */
   double outer_weight;    /* 1.0 */
   uint64_t seed;          /* PMIS tie-break hash seed */
   int    num_functions;   /* coarsening.num_functions (1); > 1 = unknown-based systems AMG */
   /* relaxation.chebyshev (relax type 16; reference src/internal/cheby.c:15-20): order 2, eig_est 10, variant 0, scale 1, fraction 0.3 */
   int    cheby_order, cheby_eig_est, cheby_variant, cheby_scale;
   double cheby_fraction;
   /* aggressive coarsening (reference AMGagg_args, src/internal/amg.c:160-173, forwarded at :938-944): on the first agg_num_levels
    * levels a second coarsening runs over the distance-two strength graph of the first one's C points (at least agg_num_paths paths of
    * length <= 2) and the interpolation is multipass (agg_interp_type 4, the only one restated; no truncation: the reference's defaults
    * max_nnz_row 0 / trunc_factor 0).  PARITY UNPINNED: no reference output uses it. */
   int    agg_num_levels, agg_num_paths, agg_interp_type;
   /* truncation of the aggressive levels' interpolation (aggressive.max_nnz_row / trunc_factor -> HYPRE_BoomerAMGSetAggPMaxElmts /
    * SetAggTruncFactor, amg.c:940-942; both 0 = none by default): hypre_BoomerAMGInterpTruncation on the finished multipass rows */
   int    agg_pmax;
   double agg_trunc_factor;
   /* V contiguous row blocks on one rank = the reference at np = V: hybrid Gauss-Seidel (relax 3/4/6/8/13/14) is Gauss-Seidel
    * inside a block and Jacobi across blocks with hypre's option-4 l1 divisor, HMIS (coarsen 10) a Ruge first pass per block +
    * PMIS on the rest; coarse levels inherit the blocks through their C points.  blocks <= 1: one block.  block_part: blocks+1
    * row starts of level 0 (NULL: hypre's even split, start q = floor(q * n / blocks)). */
   int            blocks;
   const int64_t *block_part;
   /* PMIS tie-break values: 0 = hash of the global row id (partition independent; the default), 1 = hypre's own stream -- per
    * row block (= rank) Park-Miller seeded 2747 + rank, one draw per row in row order, every level anew (orc_pmis_hypre_stream) */
   int            pmis_rng;
} orc_amg_params;

typedef struct orc_amg orc_amg; /* hierarchy handle */

typedef struct {
   int    max_iter;     /* 100 */
   double rtol;         /* 1e-6 */
   double atol;         /* 0 */
   int    two_norm;     /* 1 */
   int    krylov_dim;   /* GMRES only: 30 */
} orc_krylov_params;

void orc_amg_default_params(orc_amg_params *p, int gpu_defaults);
void orc_krylov_default_params(orc_krylov_params *p, int gmres);

/* CSR utilities */
orc_csr *orc_csr_alloc(int nrows, int ncols, int nnz);
void     orc_csr_free(orc_csr *A);
orc_csr *orc_csr_from_arrays(int nrows, int ncols, const int64_t *rowptr,
                             const int64_t *cols, const double *vals); /* copies + sorts rows */
orc_csr *orc_csr_transpose(const orc_csr *A);

/* Benchmark matrix: examples/src/C_laplacian/laplacian.c:719-921 (7-pt, Dirichlet by
 * truncation) in the block-partition numbering of laplacian.c:504-520.  Returns the
 * GLOBAL matrix; b (length N) gets 1 on the global y=0 plane else 0 (b_mode 0), or all
 * ones (b_mode 1 == the ps3d10pt7 data set of examples/ex1.yml). */
orc_csr *orc_lap7(int nx, int ny, int nz, int px, int py, int pz, double cx, double cy,
                  double cz, int b_mode, double *b);
/* Row range owned by rank r in that numbering. */
void orc_lap7_partition(int nx, int ny, int nz, int px, int py, int pz, int rank,
                        int64_t *ilower, int64_t *iupper);

/* Kernels */
void   orc_spmv(const orc_csr *A, double alpha, const double *x, double beta, double *y);
double orc_dot(int n, const double *x, const double *y);
void   orc_l1_norms(const orc_csr *A, int option, double *l1); /* 1: full row, 4: diag (1 rank) */
void   orc_relax(const orc_csr *A, const double *l1, int type, double weight,
                 const double *b, double *x, double *tmp);
/* Chebyshev smoother (hypre relax type 16; published: Adams, Brezina, Hu, Tuminaro 2003).  PARITY UNPINNED: hypre's
 * eigenvalue estimate starts from hypre_Rand values; here the start vector is the PMIS hash of the row id.
 * orc_cheby_setup fills ds (length n; 1/sqrt(a_ii) when scale, else unused) and coefs[0..order-1] and returns the
 * eigenvalue bounds it used; orc_cheby_apply performs u += p(A) (f - A u). */
void   orc_cheby_setup(const orc_csr *A, int order, int eig_est, int variant, int scale, double fraction, uint64_t seed,
                       int level, double *ds, double coefs[5], double *max_eig, double *min_eig);
void   orc_cheby_apply(const orc_csr *A, int order, int scale, const double *ds, const double coefs[5], const double *f,
                       double *u, double *r, double *v, double *w);

/* Setup pieces (exposed for per-kernel parity tests) */
void orc_strength(const orc_csr *A, double theta, double max_row_sum, unsigned char *smask);
void orc_strength_dof(const orc_csr *A, double theta, double max_row_sum, const int *dof, unsigned char *smask);
void orc_pmis(const orc_csr *A, const unsigned char *smask, uint64_t seed, int level,
              int64_t row_offset, int *cf); /* cf: 1 C, -1 F, -3 special F */
void orc_pmis_weights(const orc_csr *A, const unsigned char *smask, const double *rnd, int *cf); /* caller's tie-break values */
void orc_pmis_hypre_stream(int n, int nb, const int64_t *part, double *rnd);                      /* hypre_Rand seeded 2747 + rank */
void orc_rs_first_pass(const orc_csr *A, const unsigned char *smask, int *cf);
/* the row-block forms (see orc_amg_params.blocks); part = nb+1 ascending row starts */
void orc_hmis_blocks(const orc_csr *A, const unsigned char *smask, int nb, const int64_t *part, uint64_t seed, int level, int *cf);
void orc_l1_norms_blocks(const orc_csr *A, int option, int nb, const int64_t *part, double *l1);
void orc_relax_blocks(const orc_csr *A, const double *l1, int type, double weight, const double *b, double *x, double *tmp,
                      int nb, const int64_t *part);
orc_csr *orc_interp_extpi(const orc_csr *A, const unsigned char *smask, const int *cf,
                          int pmax, double trunc_factor);
orc_csr *orc_interp_extpi_dof(const orc_csr *A, const unsigned char *smask, const int *cf,
                              int pmax, double trunc_factor, const int *dof);
/* interp type 17 (mm-ext+i): the matrix-matrix form of extended+i, W = -D^-1 (I + B) A^s_FC -- its own operator, see amg_oracle.c */
orc_csr *orc_interp_mm_extpi_dof(const orc_csr *A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor,
                                 const int *dof);
/* interp type 3: direct interpolation with separation of weights (strong C neighbours only) */
orc_csr *orc_interp_standard_dof(const orc_csr *A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor,
                                 const int *dof); /* interp type 8 ("standard") */
orc_csr *orc_interp_direct_dof(const orc_csr *A, const unsigned char *smask, const int *cf,
                               int pmax, double trunc_factor, const int *dof);
orc_csr *orc_rap(const orc_csr *A, const orc_csr *P);
/* aggressive coarsening pieces (see orc_amg_params.agg_*): distance-two strength among the C points of cf (n1 x n1, value = number
 * of paths, entries with fewer than num_paths dropped, no diagonal); second PMIS pass folded into cf; multipass interpolation */
orc_csr *orc_second_strength(const orc_csr *A, const unsigned char *smask, const int *cf, int num_paths);
void     orc_coarsen_second_pass(const orc_csr *A, const unsigned char *smask, int num_paths, uint64_t seed, int level, int *cf);
orc_csr *orc_interp_multipass(const orc_csr *A, const unsigned char *smask, const int *cf);
/* hypre_BoomerAMGInterpTruncation on finished (column-sorted) rows, in place: relative threshold, then the pmax largest, row sums kept */
void     orc_truncate_rows(orc_csr *P, int pmax, double trunc_factor);

/* Hierarchy */
orc_amg *orc_amg_setup(const orc_csr *A, const orc_amg_params *p);
/* dof_func of level 0 for p->num_functions > 1 (NULL: i mod num_functions, hypre's default) */
orc_amg *orc_amg_setup_dof(const orc_csr *A, const orc_amg_params *p, const int *dof);
void     orc_amg_free(orc_amg *h);
int      orc_amg_num_levels(const orc_amg *h);
const orc_csr *orc_amg_A(const orc_amg *h, int lvl);
const orc_csr *orc_amg_P(const orc_amg *h, int lvl); /* NULL on coarsest */
const int     *orc_amg_cf(const orc_amg *h, int lvl);
const double  *orc_amg_l1(const orc_amg *h, int lvl, int which); /* 0 down, 1 up */
const int64_t *orc_amg_block_part(const orc_amg *h, int level); /* nblk+1 row starts of the level's blocks; NULL: one block */
double   orc_amg_operator_complexity(const orc_amg *h);
double   orc_amg_grid_complexity(const orc_amg *h);
void     orc_amg_vcycle(orc_amg *h, const double *b, double *x); /* x must hold the initial guess */
/* preconditioner reuse: level 0 := A (same size), the rest of the hierarchy as it was set up; 0 on success */
int      orc_amg_rebind_level0(orc_amg *h, const orc_csr *A);

/* ILU(0) on the diagonal blocks of a row partition (hypre "bj-iluk", fill 0, natural order; reference
 * argument surface src/internal/ilu.c:15-28).  part = nparts+1 row starts (NULL: one block).
 * Returns NULL on a missing diagonal or a zero pivot.  PARITY UNPINNED (see amg_oracle.c). */
typedef struct orc_ilu orc_ilu;
orc_ilu *orc_ilu0_setup(const orc_csr *A, int nparts, const int64_t *part, int tri_solve, int lower_it, int upper_it);
void     orc_ilu_free(orc_ilu *F);
const orc_csr *orc_ilu_factors(const orc_ilu *F);        /* L (strict lower, unit diagonal) and U in A's block pattern */
void     orc_ilu_apply(orc_ilu *F, const double *r, double *z); /* z = U^-1 L^-1 r */
/* AMG complex smoother (amg.c:899-921): ILU replaces the relaxation sweeps on levels < num_levels; one
 * smoothing step = num_sweeps iterations u += M^-1 (f - A u).  Call after orc_amg_setup*. Returns 0 on success. */
int      orc_amg_set_ilu_smoother(orc_amg *h, int num_levels, int num_sweeps, int nparts, const int64_t *part,
                                  int tri_solve, int lower_it, int upper_it);
/* "preconditioner: ilu": a handle usable wherever the Krylov routines take a hierarchy */
orc_amg *orc_precond_ilu(const orc_csr *A, int max_iter, int nparts, const int64_t *part, int tri_solve, int lower_it, int upper_it);

/* MGR (reference src/internal/mgr.c; option subset and algorithm: see amg_oracle.c).  PARITY UNPINNED. */
typedef struct {
   int        n_f_labels;
   const int *f_labels;      /* level.N.f_dofs: labels eliminated on this reduction level */
   int        interp_type;   /* prolongation_type: 0 injection, 1 l1-jacobi, 2 jacobi */
   int        restrict_type; /* restriction_type: 0 injection, 2 jacobi, 14 columped */
   int        frelax_type;   /* f_relaxation: 7 jacobi (default), 18 l1-jacobi, 2 amg (one BoomerAMG cycle on A_FF), 32 ilu (ILU(0) of A_FF) */
   int        frelax_sweeps; /* 1 */
   int        grelax_type;   /* g_relaxation: -1 none (default), 3/4/6/13/14 hybrid GS, 88 l1-hsgs, 16 ilu (ILU(0)) */
   int        grelax_sweeps; /* 1 */
   const orc_amg_params *frelax_amg; /* f_relaxation.amg block (NULL: hypre-GPU defaults) */
   /* ILU arguments of this level's ILU components (f_relaxation 32 and g_relaxation 16 share them; ilu.c:21-23) */
   int        ilu_tri_solve, ilu_lower_it, ilu_upper_it;
   /* coarsest_level: ilu -- read from the LAST level's entry when orc_precond_mgr gets coarsest_amg == NULL */
   int        coarse_ilu_max_iter, coarse_ilu_tri_solve, coarse_ilu_lower_it, coarse_ilu_upper_it;
   /* nested Krylov components (reference src/internal/krylov.c; mgr.c:3938-3960, 4253-4275): 0 none, 1 pcg, 2 gmres, 3 fgmres,
    * 4 bicgstab.  f_relaxation: a solve of A_FF e = r_F from a zero guess, preconditioned by the level's amg component
    * (frelax_type 2) when frelax_krylov_precond is set.  coarsest_level: read from the LAST level's entry, preconditioned by the
    * coarsest BoomerAMG. */
   int               frelax_krylov, frelax_krylov_precond;
   orc_krylov_params frelax_kp;
   int               coarse_krylov, coarse_krylov_precond;
   orc_krylov_params coarse_kp;
   /* cycle shape, read from the LAST level's entry; 0 = default.  mgr_cycle: 1 V, 2 W; positions (reference mgr.c:614-675,
    * hypre's SetFRelaxCycle / SetGlobalSmoothCycle): 1 before the coarse correction, 2 after it, 3 both */
   int               mgr_cycle, mgr_frelax_pos, mgr_gsmooth_pos;
   /* row blocks of the hybrid Gauss-Seidel global relaxation (g_relaxation 3/4/6/13/14/88) = the reference at np = V: Gauss-Seidel inside
    * a block, Jacobi across blocks with hypre's option-4 l1 divisor; V <= 1: one block (the sequential sweep, np = 1); V > 1: hypre's
    * even split of the level's rows (start q = floor(q n / V)) */
   int               grelax_blocks;
} orc_mgr_level_params;
orc_amg *orc_precond_mgr(const orc_csr *A, const int *labels, int nlevels, const orc_mgr_level_params *levels,
                         const orc_amg_params *coarsest_amg, int max_iter);
const orc_csr *orc_mgr_matrix(const orc_amg *h, int level, int which); /* 0 A (level == nlevels: coarsest), 1 P, 2 R */

/* Krylov (hypre_PCGSolve / hypre_GMRESSolve restatements).
 * h == NULL -> unpreconditioned.  resid_hist[k] = ||r_k||_2 for k=0..iters
 * (needs max_iter+1 doubles).  Returns iterations; *converged, *final_rel set. */
int orc_pcg(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b,
            double *x, double *resid_hist, int *converged, double *final_rel);
int orc_gmres(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b,
              double *x, double *resid_hist, int *converged, double *final_rel);
/* hypre_FlexGMRESSolve / hypre_BiCGSTABSolve restatements (solver.c:229-252); parity unpinned */
int orc_fgmres(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b,
               double *x, double *resid_hist, int *converged, double *final_rel);
int orc_bicgstab(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b,
                 double *x, double *resid_hist, int *converged, double *final_rel);

/* Dense no-pivot Gaussian elimination (hypre relax type 9). a is n*n row-major, destroyed. */
int orc_gselim(double *a, double *x, int n);

#ifdef __cplusplus
}
#endif
#endif
