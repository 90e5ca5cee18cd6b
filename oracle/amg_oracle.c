/*
 * amg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See amg_oracle.h for provenance.  Every routine cites the reference call site it
 * stands behind (paths relative to /root/reference) and, because the arithmetic lives in
 * un-vendored hypre, the published algorithm it restates.
 *
 * Determinism contract shared with the HIP product (so hierarchies can be compared
 * entry by entry): rows are column-sorted; setup arithmetic is written as plain
 * mul/add/div in a fixed order (compile with -ffp-contract=off).
 */
#include "amg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_C_PT 1
#define ORC_F_PT (-1)
#define ORC_SF_PT (-3)

/* ------------------------------------------------------------------ params */

/* src/internal/amg.c:120-238 (HYPRE_USING_GPU branch when gpu_defaults) */
void
orc_amg_default_params(orc_amg_params *p, int gpu_defaults)
{
   p->coarsen_type    = gpu_defaults ? 8 : 10;
   p->interp_type     = 6;
   p->pmax            = 4;
   p->trunc_factor    = 0.0;
   p->strong_th       = 0.25;
   p->max_row_sum     = 0.9;
   p->max_coarse_size = 64;
   p->min_coarse_size = 0;
   p->max_levels      = 25;
   p->relax_down      = gpu_defaults ? 18 : 13;
   p->relax_up        = gpu_defaults ? 18 : 14;
   p->relax_coarse    = 9;
   p->sweeps_down     = 1;
   p->sweeps_up       = 1;
   p->sweeps_coarse   = 1;
   p->relax_weight    = 1.0;
   p->outer_weight    = 1.0;
   p->seed            = 2747;
   p->cheby_order = 2; p->cheby_eig_est = 10; p->cheby_variant = 0; p->cheby_scale = 1; p->cheby_fraction = 0.3;
   p->num_functions   = 1;
   p->agg_num_levels = 0; p->agg_num_paths = 1; p->agg_interp_type = 4; /* amg.c:164-171 */
   p->agg_pmax = 0; p->agg_trunc_factor = 0.0;
   p->blocks = 1; p->block_part = NULL; p->pmis_rng = 0;
}

/* src/internal/pcg.c:15-25, src/internal/gmres.c:16-27 */
void
orc_krylov_default_params(orc_krylov_params *p, int gmres)
{
   p->max_iter   = gmres ? 300 : 100;
   p->rtol       = 1.0e-6;
   p->atol       = 0.0;
   p->two_norm   = 1;
   p->krylov_dim = 30;
}

/* --------------------------------------------------------------------- CSR */

orc_csr *
orc_csr_alloc(int nrows, int ncols, int nnz)
{
   orc_csr *A = (orc_csr *)calloc(1, sizeof(orc_csr));
   A->nrows   = nrows;
   A->ncols   = ncols;
   A->rowptr  = (int *)calloc((size_t)nrows + 1, sizeof(int));
   A->col     = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
   A->val     = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
   return A;
}

void
orc_csr_free(orc_csr *A)
{
   if (!A) return;
   free(A->rowptr);
   free(A->col);
   free(A->val);
   free(A);
}

static void
sort_row(int *c, double *v, int n)
{
   /* insertion sort: rows are short */
   for (int a = 1; a < n; a++)
   {
      int    cc = c[a];
      double vv = v[a];
      int    b  = a - 1;
      while (b >= 0 && c[b] > cc)
      {
         c[b + 1] = c[b];
         v[b + 1] = v[b];
         b--;
      }
      c[b + 1] = cc;
      v[b + 1] = vv;
   }
}

orc_csr *
orc_csr_from_arrays(int nrows, int ncols, const int64_t *rowptr, const int64_t *cols,
                    const double *vals)
{
   int64_t  base = rowptr[0];
   int      nnz  = (int)(rowptr[nrows] - base);
   orc_csr *A    = orc_csr_alloc(nrows, ncols, nnz);
   for (int i = 0; i <= nrows; i++) A->rowptr[i] = (int)(rowptr[i] - base);
   for (int k = 0; k < nnz; k++)
   {
      A->col[k] = (int)cols[base + k];
      A->val[k] = vals[base + k];
   }
   for (int i = 0; i < nrows; i++)
      sort_row(A->col + A->rowptr[i], A->val + A->rowptr[i], A->rowptr[i + 1] - A->rowptr[i]);
   return A;
}

orc_csr *
orc_csr_transpose(const orc_csr *A)
{
   int      nnz = A->rowptr[A->nrows];
   orc_csr *T   = orc_csr_alloc(A->ncols, A->nrows, nnz);
   for (int k = 0; k < nnz; k++) T->rowptr[A->col[k] + 1]++;
   for (int j = 0; j < A->ncols; j++) T->rowptr[j + 1] += T->rowptr[j];
   int *pos = (int *)malloc(sizeof(int) * (size_t)(A->ncols + 1));
   memcpy(pos, T->rowptr, sizeof(int) * (size_t)(A->ncols + 1));
   for (int i = 0; i < A->nrows; i++)
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
      {
         int q     = pos[A->col[k]]++;
         T->col[q] = i; /* ascending i: rows of T come out sorted */
         T->val[q] = A->val[k];
      }
   free(pos);
   return T;
}

/* ---------------------------------------------------------- lap7 generator */

typedef struct {
   int     gd[3], pd[3];
   int64_t *ps[3];
} lapmesh;

static void
mesh_init(lapmesh *m, int nx, int ny, int nz, int px, int py, int pz)
{
   /* examples/src/C_laplacian/laplacian.c:561-571: size*j + min(j, rest) */
   m->gd[0] = nx; m->gd[1] = ny; m->gd[2] = nz;
   m->pd[0] = px; m->pd[1] = py; m->pd[2] = pz;
   for (int d = 0; d < 3; d++)
   {
      int size = m->gd[d] / m->pd[d];
      int rest = m->gd[d] - size * m->pd[d];
      m->ps[d] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m->pd[d] + 1));
      for (int j = 0; j <= m->pd[d]; j++) m->ps[d][j] = (int64_t)size * j + (j < rest ? j : rest);
   }
}

static void
mesh_free(lapmesh *m)
{
   for (int d = 0; d < 3; d++) free(m->ps[d]);
}

/* laplacian.c:504-520: blocks numbered in Cartesian rank order (z fastest over blocks),
 * x fastest inside a block. */
static int64_t
mesh_idx(const lapmesh *m, const int64_t g[3], const int bc[3])
{
   int64_t **ps = (int64_t **)m->ps;
   int64_t   lx = ps[0][bc[0] + 1] - ps[0][bc[0]];
   int64_t   ly = ps[1][bc[1] + 1] - ps[1][bc[1]];
   return ps[0][bc[0]] * m->gd[1] * m->gd[2] + ps[1][bc[1]] * m->gd[2] * lx +
          ps[2][bc[2]] * lx * ly + ((g[2] - ps[2][bc[2]]) * ly + (g[1] - ps[1][bc[1]])) * lx +
          (g[0] - ps[0][bc[0]]);
}

static int
block_of(const int64_t *ps, int np, int64_t g)
{
   int b = 0;
   while (b + 1 < np && g >= ps[b + 1]) b++;
   return b;
}

void
orc_lap7_partition(int nx, int ny, int nz, int px, int py, int pz, int rank,
                   int64_t *ilower, int64_t *iupper)
{
   lapmesh m;
   mesh_init(&m, nx, ny, nz, px, py, pz);
   /* MPI_Cart_create row-major: rank = (cx*py + cy)*pz + cz */
   int     bc[3] = {rank / (py * pz), (rank / pz) % py, rank % pz};
   int64_t g[3]  = {m.ps[0][bc[0]], m.ps[1][bc[1]], m.ps[2][bc[2]]};
   *ilower       = mesh_idx(&m, g, bc);
   int64_t cnt   = (m.ps[0][bc[0] + 1] - g[0]) * (m.ps[1][bc[1] + 1] - g[1]) *
                 (m.ps[2][bc[2] + 1] - g[2]);
   *iupper = *ilower + cnt - 1;
   mesh_free(&m);
}

orc_csr *
orc_lap7(int nx, int ny, int nz, int px, int py, int pz, double cx, double cy, double cz,
         int b_mode, double *b)
{
   lapmesh m;
   mesh_init(&m, nx, ny, nz, px, py, pz);
   int64_t N   = (int64_t)nx * ny * nz;
   int64_t nnz = 7 * N - 2 * ((int64_t)nx * ny + (int64_t)ny * nz + (int64_t)nx * nz);
   orc_csr *A  = orc_csr_alloc((int)N, (int)N, (int)nnz);
   /* first pass: per-row counts (rows are visited in grid order, not index order) */
   for (int64_t gz = 0; gz < nz; gz++)
      for (int64_t gy = 0; gy < ny; gy++)
         for (int64_t gx = 0; gx < nx; gx++)
         {
            int     bc[3] = {block_of(m.ps[0], px, gx), block_of(m.ps[1], py, gy),
                             block_of(m.ps[2], pz, gz)};
            int64_t g[3]  = {gx, gy, gz};
            int64_t row   = mesh_idx(&m, g, bc);
            int     cnt   = 1 + (gx > 0) + (gx < nx - 1) + (gy > 0) + (gy < ny - 1) + (gz > 0) +
                      (gz < nz - 1);
            A->rowptr[row + 1] = cnt;
         }
   for (int64_t i = 0; i < N; i++) A->rowptr[i + 1] += A->rowptr[i];
   static const int dx[6] = {0, 0, -1, 1, 0, 0};
   static const int dy[6] = {0, -1, 0, 0, 1, 0};
   static const int dz[6] = {-1, 0, 0, 0, 0, 1};
   const double     cc[6] = {cz, cy, cx, cx, cy, cz};
   for (int64_t gz = 0; gz < nz; gz++)
      for (int64_t gy = 0; gy < ny; gy++)
         for (int64_t gx = 0; gx < nx; gx++)
         {
            int     bc[3] = {block_of(m.ps[0], px, gx), block_of(m.ps[1], py, gy),
                             block_of(m.ps[2], pz, gz)};
            int64_t g[3]  = {gx, gy, gz};
            int64_t row   = mesh_idx(&m, g, bc);
            int     q     = A->rowptr[row];
            A->col[q]     = (int)row;
            A->val[q++]   = 2.0 * (cx + cy + cz); /* laplacian.c:790 */
            for (int s = 0; s < 6; s++)
            {
               int64_t h[3] = {gx + dx[s], gy + dy[s], gz + dz[s]};
               if (h[0] < 0 || h[0] >= nx || h[1] < 0 || h[1] >= ny || h[2] < 0 || h[2] >= nz)
                  continue; /* Dirichlet by truncation */
               int nb[3]   = {block_of(m.ps[0], px, h[0]), block_of(m.ps[1], py, h[1]),
                              block_of(m.ps[2], pz, h[2])};
               A->col[q]   = (int)mesh_idx(&m, h, nb);
               A->val[q++] = -cc[s];
            }
            sort_row(A->col + A->rowptr[row], A->val + A->rowptr[row], q - A->rowptr[row]);
            if (b) b[row] = (b_mode == 1) ? 1.0 : ((gy == 0) ? 1.0 : 0.0); /* laplacian.c:898-905 */
         }
   mesh_free(&m);
   return A;
}

/* ----------------------------------------------------------------- kernels */

/* HYPRE_ParCSRMatrixMatvec, reached from src/internal/linsys.c:3031 */
void
orc_spmv(const orc_csr *A, double alpha, const double *x, double beta, double *y)
{
#pragma omp parallel for schedule(static)
   for (int i = 0; i < A->nrows; i++)
   {
      double s = 0.0;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) s += A->val[k] * x[A->col[k]];
      y[i] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[i];
   }
}

/* hypre_ParVectorInnerProd, reached from src/internal/linsys.c:2875 */
double
orc_dot(int n, const double *x, const double *y)
{
   double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
   for (int i = 0; i < n; i++) s += x[i] * y[i];
   return s;
}

/* hypre_ParCSRComputeL1Norms.  option 1: sum_j |a_ij| (relax 18);
 * option 4: a_ii + 0.5*sum_offd|a_ij| truncated -> on one rank just a_ii (relax 13/14/8). */
void
orc_l1_norms(const orc_csr *A, int option, double *l1)
{
   for (int i = 0; i < A->nrows; i++)
   {
      double s = 0.0, d = 0.0;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
      {
         s += fabs(A->val[k]);
         if (A->col[k] == i) d = A->val[k];
      }
      if (option == 1)
         l1[i] = (d < 0.0) ? -s : s;
      else
         l1[i] = d;
   }
}

/* hypre_BoomerAMGRelax types selectable via src/internal/amg.c:360-375.
 * One rank, relax_order 0 (lexicographic), all points. */
void
orc_relax(const orc_csr *A, const double *l1, int type, double w, const double *b, double *x,
          double *tmp)
{
   const int n = A->nrows;
   switch (type)
   {
      case 0:  /* weighted Jacobi: divide by a_ii */
      case 7:
      case 18: /* l1-Jacobi: divide by l1 */
      {
#pragma omp parallel for schedule(static)
         for (int i = 0; i < n; i++)
         {
            double r = b[i], d = 0.0;
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            {
               r -= A->val[k] * x[A->col[k]];
               if (A->col[k] == i) d = A->val[k];
            }
            if (type == 18) d = l1[i];
            tmp[i] = x[i] + w * r / d;
         }
         memcpy(x, tmp, sizeof(double) * (size_t)n);
         break;
      }
      case 3:  /* hybrid GS forward  */
      case 13: /* l1 hybrid GS forward: one rank => l1 = a_ii */
         for (int i = 0; i < n; i++)
         {
            double r = b[i], d = 0.0;
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            {
               r -= A->val[k] * x[A->col[k]];
               if (A->col[k] == i) d = A->val[k];
            }
            if (type == 13) d = l1[i];
            x[i] += w * r / d;
         }
         break;
      case 4:  /* hybrid GS backward */
      case 14:
         for (int i = n - 1; i >= 0; i--)
         {
            double r = b[i], d = 0.0;
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            {
               r -= A->val[k] * x[A->col[k]];
               if (A->col[k] == i) d = A->val[k];
            }
            if (type == 14) d = l1[i];
            x[i] += w * r / d;
         }
         break;
      case 6: /* symmetric GS */
      case 8: /* l1 symmetric GS */
         orc_relax(A, l1, type == 6 ? 3 : 13, w, b, x, tmp);
         orc_relax(A, l1, type == 6 ? 4 : 14, w, b, x, tmp);
         break;
      default:
         fprintf(stderr, "orc_relax: unsupported relax type %d\n", type);
         abort();
   }
}

/* ---- V contiguous row blocks on one rank = what the reference computes at np = V.
 * hypre's hybrid smoothers are Gauss-Seidel inside a rank's rows and Jacobi across ranks, its option-4 l1 divisor adds half the
 * absolute sum of the entries that leave the rank, and HMIS is a Ruge first pass per rank followed by PMIS on
 * everything that pass did not fix (App. A.5).  part = nb+1 ascending row starts, part[0] = 0, part[nb] = nrows. */
static inline int
blk_find(const int64_t *part, int nb, int i)
{
   int lo = 0, hi = nb; /* part[lo] <= i < part[hi] */
   while (hi - lo > 1)
   {
      int mid = (lo + hi) / 2;
      if (part[mid] <= i) lo = mid;
      else hi = mid;
   }
   return lo;
}

/* hypre_ParCSRComputeL1Norms with the rows outside i's block in the role of the off-processor part.  option 1: full-row sum
 * (blocks make no difference); option 4: |a_ii| + 0.5 * sum over entries leaving the block, truncated to |a_ii| when that is
 * <= 4/3 |a_ii|; sign of a_ii restored.  nb <= 1 gives orc_l1_norms bit for bit. */
void
orc_l1_norms_blocks(const orc_csr *A, int option, int nb, const int64_t *part, double *l1)
{
   if (nb <= 1 || !part || option == 1)
   {
      orc_l1_norms(A, option, l1);
      return;
   }
   for (int b = 0; b < nb; b++)
   {
      const int lo = (int)part[b], hi = (int)part[b + 1];
      for (int i = lo; i < hi; i++)
      {
         double d = 0.0, off = 0.0;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         {
            const int j = A->col[k];
            if (j == i) d = A->val[k];
            else if (j < lo || j >= hi) off += fabs(A->val[k]);
         }
         const double ad = fabs(d);
         double       v  = ad + 0.5 * off;
         if (v <= 4.0 / 3.0 * ad) v = ad;
         l1[i] = (d < 0.0) ? -v : v;
      }
   }
}

/* orc_relax on V row blocks: the Gauss-Seidel family (3/4/6/8/13/14) sees the values the OTHER blocks held when the sweep
 * began (tmp keeps them), its own block's values as they are updated; the Jacobi family is what it was.  nb <= 1 gives
 * orc_relax bit for bit (same products, same order). */
void
orc_relax_blocks(const orc_csr *A, const double *l1, int type, double w, const double *b, double *x, double *tmp, int nb,
                 const int64_t *part)
{
   const int n = A->nrows;
   if (nb <= 1 || !part || type == 0 || type == 7 || type == 18)
   {
      orc_relax(A, l1, type, w, b, x, tmp);
      return;
   }
   if (type == 6 || type == 8)
   {
      orc_relax_blocks(A, l1, type == 6 ? 3 : 13, w, b, x, tmp, nb, part);
      orc_relax_blocks(A, l1, type == 6 ? 4 : 14, w, b, x, tmp, nb, part);
      return;
   }
   if (type != 3 && type != 13 && type != 4 && type != 14)
   {
      fprintf(stderr, "orc_relax_blocks: unsupported relax type %d\n", type);
      abort();
   }
   const int fwd = (type == 3 || type == 13), use_l1 = (type == 13 || type == 14);
   memcpy(tmp, x, sizeof(double) * (size_t)n);
#pragma omp parallel for schedule(dynamic, 1)
   for (int q = 0; q < nb; q++)
   {
      const int lo = (int)part[q], hi = (int)part[q + 1];
      for (int s = 0; s < hi - lo; s++)
      {
         const int i = fwd ? lo + s : hi - 1 - s;
         double    r = b[i], d = 0.0;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         {
            const int j = A->col[k];
            r -= A->val[k] * ((j >= lo && j < hi) ? x[j] : tmp[j]);
            if (j == i) d = A->val[k];
         }
         if (use_l1) d = l1[i];
         x[i] += w * r / d;
      }
   }
}

/* hypre relax type 9 (coarse_type ge, src/internal/amg.c:190): hypre_gselim, no pivoting */
int
orc_gselim(double *a, double *x, int n)
{
   if (n == 1)
   {
      if (a[0] == 0.0) return 1;
      x[0] /= a[0];
      return 0;
   }
   for (int k = 0; k < n - 1; k++)
   {
      if (a[k * n + k] == 0.0) return 1;
      for (int j = k + 1; j < n; j++)
      {
         if (a[j * n + k] != 0.0)
         {
            double f = a[j * n + k] / a[k * n + k];
            for (int m = k + 1; m < n; m++) a[j * n + m] -= f * a[k * n + m];
            x[j] -= f * x[k];
         }
      }
   }
   for (int k = n - 1; k > 0; k--)
   {
      if (a[k * n + k] == 0.0) return 1;
      x[k] /= a[k * n + k];
      for (int j = 0; j < k; j++)
         if (a[j * n + k] != 0.0) x[j] -= x[k] * a[j * n + k];
   }
   if (a[0] == 0.0) return 1;
   x[0] /= a[0];
   return 0;
}

/* ------------------------------------------------------------------- setup */

/* hypre_BoomerAMGCreateS; theta = coarsening.strong_th
 * (src/internal/amg.c:156), max_row_sum (amg.c:155).  smask[k]=1 iff entry k is strong. */
void
orc_strength(const orc_csr *A, double theta, double max_row_sum, unsigned char *smask)
{
   orc_strength_dof(A, theta, max_row_sum, NULL, smask);
}

/* Systems AMG, unknown approach (coarsening.num_functions > 1, src/internal/amg.c:147,792-862;
 * presets elasticity_2d/3d, src/internal/presets.c:19-27): couplings between different
 * functions take no part in the strength decision -- neither in the row maximum, nor in the
 * row sum, nor as strong connections (hypre_BoomerAMGCreateS).  dof == NULL:
 * scalar problem. */
void
orc_strength_dof(const orc_csr *A, double theta, double max_row_sum, const int *dof, unsigned char *smask)
{
   for (int i = 0; i < A->nrows; i++)
   {
      double diag = 0.0, row_sum = 0.0, row_scale = 0.0;
      int    k0 = A->rowptr[i], k1 = A->rowptr[i + 1];
      for (int k = k0; k < k1; k++)
         if (A->col[k] == i) diag = A->val[k];
      for (int k = k0; k < k1; k++)
      {
         if (dof && dof[A->col[k]] != dof[i]) continue;
         row_sum += A->val[k];
         if (A->col[k] == i) continue;
         if (diag < 0.0)
            row_scale = (A->val[k] > row_scale) ? A->val[k] : row_scale;
         else
            row_scale = (A->val[k] < row_scale) ? A->val[k] : row_scale;
      }
      int weak_row = (max_row_sum < 1.0) && (diag != 0.0) && (fabs(row_sum / diag) > max_row_sum);
      for (int k = k0; k < k1; k++)
      {
         int s = 0;
         if (A->col[k] != i && !weak_row && !(dof && dof[A->col[k]] != dof[i]))
         {
            if (diag < 0.0)
               s = A->val[k] > theta * row_scale;
            else
               s = A->val[k] < theta * row_scale;
         }
         smask[k] = (unsigned char)s;
      }
   }
}

static inline uint64_t
mix64(uint64_t z)
{
   z += 0x9E3779B97F4A7C15ULL;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   return z ^ (z >> 31);
}

/* PMIS tie-break weight in [0,1): hypre uses hypre_Rand() seeded 2747+rank (upstream,
 * unverifiable here; CPU and GPU hypre already disagree) -- we use a hash of the global
 * row id so the split is partition independent. */
static inline double
pmis_rand(uint64_t seed, int level, int64_t gid)
{
   uint64_t h = mix64(mix64(seed + (uint64_t)level * 0x100000001B3ULL) ^ (uint64_t)gid);
   return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

/* strong transpose graph: for each j, the rows i with j in S_i */
static void
strong_transpose(const orc_csr *A, const unsigned char *smask, int **tp_out, int **tj_out)
{
   int  n  = A->nrows;
   int *tp = (int *)calloc((size_t)n + 1, sizeof(int));
   for (int i = 0; i < n; i++)
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         if (smask[k]) tp[A->col[k] + 1]++;
   for (int j = 0; j < n; j++) tp[j + 1] += tp[j];
   int *tj  = (int *)malloc(sizeof(int) * (size_t)(tp[n] > 0 ? tp[n] : 1));
   int *pos = (int *)malloc(sizeof(int) * (size_t)(n + 1));
   memcpy(pos, tp, sizeof(int) * (size_t)(n + 1));
   for (int i = 0; i < n; i++)
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         if (smask[k]) tj[pos[A->col[k]]++] = i;
   free(pos);
   *tp_out = tp;
   *tj_out = tj;
}

/* hypre_BoomerAMGCoarsenPMIS (coarsen type 8, src/internal/amg.c:303-308), SURVEY App. A.5,
 * written as synchronous rounds so a data-parallel implementation gives the same split. */
/* hypre's own tie-break stream (hypre_BoomerAMGIndepSetInit with seq_rand 0): every rank seeds hypre_Rand --
 * the Park-Miller minimal standard generator, seed <- 16807 * seed mod (2^31 - 1), value seed / (2^31 - 1) -- with 2747 + rank and
 * draws once per local row in row order, on every level anew.  Row blocks are the ranks (orc_amg_params.blocks).  PARITY: this is
 * the published generator and the upstream convention as SURVEY records it; hypre is not in the reference tree, so whether it
 * reproduces a checked-in hierarchy is an experiment (tests/test_oracle_pins.py). */
static int pmis_stream_rank_offset = 1; /* experiment switch (tools/pmis_rng_experiment.py): 0 = every rank seeded 2747 */
void orc_pmis_stream_rank_offset(int on) { pmis_stream_rank_offset = on; }
void
orc_pmis_hypre_stream(int n, int nb, const int64_t *part, double *rnd)
{
   const int64_t one[2] = {0, n};
   if (nb <= 1 || !part) { nb = 1; part = one; }
   for (int q = 0; q < nb; q++)
   {
      int64_t seed = 2747 + (pmis_stream_rank_offset ? q : 0);
      for (int64_t i = part[q]; i < part[q + 1]; i++)
      {
         seed = (16807 * seed) % 2147483647LL;
         rnd[i] = (double)seed / 2147483647.0;
      }
   }
}

static void pmis_core(const orc_csr *A, const unsigned char *smask, const double *rnd, int *cf);

void
orc_pmis(const orc_csr *A, const unsigned char *smask, uint64_t seed, int level,
         int64_t row_offset, int *cf)
{
   const int n   = A->nrows;
   double   *rnd = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
   for (int i = 0; i < n; i++) rnd[i] = pmis_rand(seed, level, row_offset + i);
   pmis_core(A, smask, rnd, cf);
   free(rnd);
}

/* the same with caller-supplied tie-break values in [0, 1) */
void
orc_pmis_weights(const orc_csr *A, const unsigned char *smask, const double *rnd, int *cf)
{
   pmis_core(A, smask, rnd, cf);
}

/* nb > 1 with part: hypre's PARALLEL rounds on nb ranks (= row blocks).  A rank learns that a remote point became F one round late:
 * the C/F markers of the ghost points are exchanged right after the independent set is picked and before the F points of the round
 * are set, and a ghost's measure is zeroed only on a non-zero marker -- so in the following round's independent-set pass a local
 * row still sees such a ghost column with its full measure and is kept out of the set by it when the ghost's measure is larger
 * (upstream behaviour of hypre_BoomerAMGCoarsenPMIS / hypre_BoomerAMGIndepSet).  nb <= 1: the global rounds. */
static void
pmis_core_ranks(const orc_csr *A, const unsigned char *smask, const double *rnd, int *cf, int nb, const int64_t *part)
{
   int     n = A->nrows;
   int    *tp, *tj;
   double *meas = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
   char   *newc = (char *)malloc((size_t)(n > 0 ? n : 1));
   int    *blk = NULL, *fround = NULL;
   if (nb > 1 && part)
   {
      blk    = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
      fround = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
      for (int q = 0; q < nb; q++)
         for (int64_t i = part[q]; i < part[q + 1]; i++) { blk[i] = q; fround[i] = -2; }
   }
   strong_transpose(A, smask, &tp, &tj);
   int undecided = 0;
   for (int i = 0; i < n; i++)
   {
      int ns = 0;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) ns += smask[k];
      int nt  = tp[i + 1] - tp[i];
      meas[i] = (double)nt + rnd[i];
      if (ns == 0)
         cf[i] = ORC_SF_PT; /* no strong dependence: special F, never interpolated */
      else if (nt == 0)
         cf[i] = ORC_F_PT; /* measure < 1: nobody depends on it */
      else
      {
         cf[i] = 0;
         undecided++;
      }
   }
   for (int round = 0; undecided > 0; round++)
   {
      for (int i = 0; i < n; i++)
      {
         newc[i] = 0;
         if (cf[i] != 0) continue;
         int    is_max = 1;
         double mi     = meas[i];
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1] && is_max; k++)
         {
            int j = A->col[k];
            if (!smask[k]) continue;
            /* a column counts while it is undecided -- or, on another rank, while this rank has not yet heard that it became F */
            const int stale = blk && blk[j] != blk[i] && cf[j] == ORC_F_PT && fround[j] == round - 1;
            if (cf[j] != 0 && !stale) continue;
            if (meas[j] > mi || (meas[j] == mi && j > i)) is_max = 0;
         }
         for (int k = tp[i]; k < tp[i + 1] && is_max; k++)
         {
            int j = tj[k];
            if (cf[j] != 0) continue;
            if (meas[j] > mi || (meas[j] == mi && j > i)) is_max = 0;
         }
         newc[i] = (char)is_max;
      }
      for (int i = 0; i < n; i++)
         if (newc[i])
         {
            cf[i] = ORC_C_PT;
            undecided--;
         }
      for (int i = 0; i < n; i++)
      {
         if (cf[i] != 0) continue;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            if (smask[k] && cf[A->col[k]] == ORC_C_PT)
            {
               cf[i] = ORC_F_PT;
               if (fround) fround[i] = round;
               undecided--;
               break;
            }
      }
   }
   free(tp);
   free(tj);
   free(meas);
   free(newc);
   free(blk);
   free(fround);
}
static void
pmis_core(const orc_csr *A, const unsigned char *smask, const double *rnd, int *cf)
{
   pmis_core_ranks(A, smask, rnd, cf, 1, NULL);
}

/* Measure buckets as FIFO doubly-linked lists: the structure hypre's Ruge first pass
 * keeps (one list per measure value, new/updated points appended at the TAIL, next C
 * point = HEAD of the highest non-empty list).  FIFO order is what makes the first
 * level of a 7-pt grid come out red-black. */
typedef struct {
   int *head, *tail, *prev, *next, *key, nb, maxkey;
} rsbuckets;

static void
bk_enter(rsbuckets *b, int i, int key)
{
   if (key >= b->nb)
   {
      int nn  = 2 * key + 8;
      b->head = (int *)realloc(b->head, sizeof(int) * (size_t)nn);
      b->tail = (int *)realloc(b->tail, sizeof(int) * (size_t)nn);
      for (int q = b->nb; q < nn; q++) b->head[q] = b->tail[q] = -1;
      b->nb = nn;
   }
   b->key[i]  = key;
   b->next[i] = -1;
   b->prev[i] = b->tail[key];
   if (b->tail[key] >= 0) b->next[b->tail[key]] = i;
   else b->head[key] = i;
   b->tail[key] = i;
   if (key > b->maxkey) b->maxkey = key;
}
static void
bk_remove(rsbuckets *b, int i)
{
   int key = b->key[i];
   if (b->prev[i] >= 0) b->next[b->prev[i]] = b->next[i];
   else b->head[key] = b->next[i];
   if (b->next[i] >= 0) b->prev[b->next[i]] = b->prev[i];
   else b->tail[key] = b->prev[i];
   b->prev[i] = b->next[i] = -1;
}
static int
bk_top(rsbuckets *b)
{
   while (b->maxkey > 0 && b->head[b->maxkey] < 0) b->maxkey--;
   return (b->maxkey > 0) ? b->head[b->maxkey] : -1;
}

/* hypre_BoomerAMGCoarsenRuge first pass on the rows [lo, hi) of A with every connection that leaves the range ignored (a rank's
 * S_diag): measures count in-range dependants only (measure_type 0), a row is "special F" only when its FULL row has no strong
 * entry.  tp/tj = strong transpose of the whole matrix; prev/next/key/meas are indexed by row, so disjoint ranges may run
 * concurrently. */
static void
rs_pass_range(const orc_csr *A, const unsigned char *smask, const int *tp, const int *tj, int lo, int hi, int *cf, int *prev,
              int *next, int *key, int *meas)
{
   rsbuckets B;
   B.nb     = 64;
   B.maxkey = 0;
   B.head   = (int *)malloc(sizeof(int) * (size_t)B.nb);
   B.tail   = (int *)malloc(sizeof(int) * (size_t)B.nb);
   for (int q = 0; q < B.nb; q++) B.head[q] = B.tail[q] = -1;
   B.prev = prev;
   B.next = next;
   B.key  = key;
#define IN_RANGE(j) ((j) >= lo && (j) < hi)
   for (int i = lo; i < hi; i++)
   {
      int ns = 0, nt = 0;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) ns += smask[k];
      for (int k = tp[i]; k < tp[i + 1]; k++) nt += IN_RANGE(tj[k]);
      meas[i]   = nt;
      B.prev[i] = B.next[i] = -1;
      B.key[i]  = 0;
      cf[i]     = (ns == 0) ? ORC_SF_PT : 0;
      if (cf[i] == ORC_SF_PT) meas[i] = 0;
   }
   /* ascending-index insertion; measure-0 points become F and the points they depend on
    * gain weight (re-listed at the tail when already listed) */
   for (int j = lo; j < hi; j++)
   {
      if (cf[j] == ORC_SF_PT) continue;
      if (meas[j] > 0) { bk_enter(&B, j, meas[j]); continue; }
      cf[j] = ORC_F_PT;
      for (int k = A->rowptr[j]; k < A->rowptr[j + 1]; k++)
      {
         int nb = A->col[k];
         if (!smask[k] || !IN_RANGE(nb) || cf[nb] == ORC_SF_PT) continue;
         if (nb < j)
         {
            if (cf[nb] != 0) { meas[nb]++; continue; }
            if (meas[nb] > 0) bk_remove(&B, nb);
            meas[nb]++;
            bk_enter(&B, nb, meas[nb]);
         }
         else
            meas[nb]++;
      }
   }
   for (;;)
   {
      int i = bk_top(&B);
      if (i < 0) break;
      bk_remove(&B, i);
      cf[i]   = ORC_C_PT;
      meas[i] = 0;
      /* everything that strongly depends on i becomes F */
      for (int k = tp[i]; k < tp[i + 1]; k++)
      {
         int j = tj[k];
         if (!IN_RANGE(j) || cf[j] != 0) continue;
         cf[j] = ORC_F_PT;
         bk_remove(&B, j);
         for (int kk = A->rowptr[j]; kk < A->rowptr[j + 1]; kk++)
         {
            int m = A->col[kk];
            if (smask[kk] && IN_RANGE(m) && cf[m] == 0)
            {
               bk_remove(&B, m);
               meas[m]++;
               bk_enter(&B, m, meas[m]);
            }
         }
      }
      /* points i depends on lose one potential dependant */
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
      {
         int j = A->col[k];
         if (!smask[k] || !IN_RANGE(j) || cf[j] != 0) continue;
         bk_remove(&B, j);
         meas[j]--;
         if (meas[j] > 0)
            bk_enter(&B, j, meas[j]);
         else
         {
            cf[j] = ORC_F_PT;
            for (int kk = A->rowptr[j]; kk < A->rowptr[j + 1]; kk++)
            {
               int m = A->col[kk];
               if (smask[kk] && IN_RANGE(m) && cf[m] == 0)
               {
                  bk_remove(&B, m);
                  meas[m]++;
                  bk_enter(&B, m, meas[m]);
               }
            }
         }
      }
   }
#undef IN_RANGE
   free(B.head); free(B.tail);
}

/* hypre_BoomerAMGCoarsenRuge first pass == HMIS (type 10) on a single rank, where the
 * "interior" is the whole grid and the trailing PMIS finds nothing left. */
void
orc_rs_first_pass(const orc_csr *A, const unsigned char *smask, int *cf)
{
   int  n = A->nrows;
   int *tp, *tj;
   strong_transpose(A, smask, &tp, &tj);
   int *w = (int *)malloc(sizeof(int) * 4 * (size_t)(n + 1));
   rs_pass_range(A, smask, tp, tj, 0, n, cf, w, w + (n + 1), w + 2 * (size_t)(n + 1), w + 3 * (size_t)(n + 1));
   free(tp); free(tj); free(w);
}

/* HMIS (coarsen type 10, the reference's CPU default src/internal/amg.c:141-146) on V row blocks = hypre at np = V (De Sterck,
 * Yang, Heys 2006, "HMIS"; hypre_BoomerAMGCoarsenHMIS = CoarsenRuge(type 10) + CoarsenPMIS(CF_init 1)): a Ruge first pass inside
 * every block on the block's own connections; of its result only the C points of INTERIOR rows (rows without a strong connection
 * leaving the block) are kept, every other point goes back to undecided; PMIS then starts from those C points as its first
 * independent set, with measures counted over the whole matrix.  One block: every C point of the first pass is interior. */
static void hmis_core(const orc_csr *A, const unsigned char *smask, int nb, const int64_t *part, const double *rnd, int *cf);
void
orc_hmis_blocks(const orc_csr *A, const unsigned char *smask, int nb, const int64_t *part, uint64_t seed, int level, int *cf)
{
   const int n   = A->nrows;
   double   *rnd = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
   for (int i = 0; i < n; i++) rnd[i] = pmis_rand(seed, level, i);
   hmis_core(A, smask, nb, part, rnd, cf);
   free(rnd);
}
static void
hmis_core(const orc_csr *A, const unsigned char *smask, int nb, const int64_t *part, const double *rnd, int *cf)
{
   const int     n      = A->nrows;
   const int64_t one[2] = {0, n};
   if (nb <= 1 || !part) { nb = 1; part = one; }
   int *tp, *tj;
   strong_transpose(A, smask, &tp, &tj);
   int *w = (int *)malloc(sizeof(int) * 4 * (size_t)(n + 1));
#pragma omp parallel for schedule(dynamic, 1)
   for (int q = 0; q < nb; q++)
      rs_pass_range(A, smask, tp, tj, (int)part[q], (int)part[q + 1], cf, w, w + (n + 1), w + 2 * (size_t)(n + 1),
                    w + 3 * (size_t)(n + 1));
   free(w);
   /* keep the interior C points; the trailing PMIS decides the rest */
   for (int q = 0; q < nb; q++)
   {
      const int lo = (int)part[q], hi = (int)part[q + 1];
      for (int i = lo; i < hi; i++)
      {
         if (cf[i] == ORC_SF_PT) continue;
         int boundary = 0;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1] && !boundary; k++)
            boundary = smask[k] && (A->col[k] < lo || A->col[k] >= hi);
         if (boundary || cf[i] != ORC_C_PT) cf[i] = 0;
      }
   }
   /* PMIS with the kept C points as first independent set (orc_pmis's synchronous rounds) */
   double *meas = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
   char   *newc = (char *)malloc((size_t)(n > 0 ? n : 1));
   int     undecided = 0;
   for (int i = 0; i < n; i++)
   {
      const int nt = tp[i + 1] - tp[i];
      meas[i]      = (double)nt + rnd[i];
      if (cf[i] != 0) continue; /* special F, kept C */
      if (nt == 0) cf[i] = ORC_F_PT; /* measure < 1 */
      else undecided++;
   }
   /* points that depend on a kept C point are F before the first round */
   for (int i = 0; i < n; i++)
   {
      if (cf[i] != 0) continue;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         if (smask[k] && cf[A->col[k]] == ORC_C_PT)
         {
            cf[i] = ORC_F_PT;
            undecided--;
            break;
         }
   }
   while (undecided > 0)
   {
      for (int i = 0; i < n; i++)
      {
         newc[i] = 0;
         if (cf[i] != 0) continue;
         int    is_max = 1;
         double mi     = meas[i];
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1] && is_max; k++)
         {
            int j = A->col[k];
            if (!smask[k] || cf[j] != 0) continue;
            if (meas[j] > mi || (meas[j] == mi && j > i)) is_max = 0;
         }
         for (int k = tp[i]; k < tp[i + 1] && is_max; k++)
         {
            int j = tj[k];
            if (cf[j] != 0) continue;
            if (meas[j] > mi || (meas[j] == mi && j > i)) is_max = 0;
         }
         newc[i] = (char)is_max;
      }
      for (int i = 0; i < n; i++)
         if (newc[i])
         {
            cf[i] = ORC_C_PT;
            undecided--;
         }
      for (int i = 0; i < n; i++)
      {
         if (cf[i] != 0) continue;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            if (smask[k] && cf[A->col[k]] == ORC_C_PT)
            {
               cf[i] = ORC_F_PT;
               undecided--;
               break;
            }
      }
   }
   free(tp); free(tj); free(meas); free(newc);
}

typedef struct {
   int    c;
   double w;
} pent;

static int
pent_cmp_col(const void *a, const void *b)
{
   return ((const pent *)a)->c - ((const pent *)b)->c;
}
/* Descending-|w| quicksort in the K&R form hypre's hypre_qsort2_abs uses (pivot = middle
 * element swapped to the front, strict '>' partition).  The exact form matters: on
 * structured grids interpolation weights tie exactly (six 1/6 weights on a red-black
 * 7-pt F point) and the tie order decides which pmax entries survive truncation.
 * Iterative with an explicit stack so a GPU thread can run the identical sequence. */
static void
pent_qsort_abs(pent *v, int n)
{
   int stack[128], sp = 0;
   stack[sp++] = 0;
   stack[sp++] = n - 1;
   while (sp > 0)
   {
      int right = stack[--sp], left = stack[--sp];
      while (left < right)
      {
         int  mid = (left + right) / 2, last = left;
         pent t   = v[left]; v[left] = v[mid]; v[mid] = t;
         for (int i = left + 1; i <= right; i++)
            if (fabs(v[i].w) > fabs(v[left].w))
            {
               ++last;
               t = v[last]; v[last] = v[i]; v[i] = t;
            }
         t = v[left]; v[left] = v[last]; v[last] = t;
         /* the two sub-ranges are disjoint, so processing order does not change the
          * result: loop on the smaller one, push the larger (stack depth <= log2 n) */
         if (last - left < right - last)
         {
            stack[sp++] = last + 1; stack[sp++] = right;
            right = last - 1;
         }
         else
         {
            stack[sp++] = left; stack[sp++] = last - 1;
            left = last + 1;
         }
      }
   }
}

/* hypre_BoomerAMGInterpTruncation on one row (still in discovery order): relative threshold, then keep the
 * pmax largest; each step rescales to preserve the row sum.  Returns the new entry count. */
static int
orc_truncate_row(pent *row, int cnt, int pmax, double trunc_factor)
{
   if (trunc_factor > 0.0 && cnt > 0)
   {
      double mx = 0.0, tot = 0.0, kept = 0.0;
      for (int q = 0; q < cnt; q++) { if (fabs(row[q].w) > mx) mx = fabs(row[q].w); tot += row[q].w; }
      int c2 = 0;
      for (int q = 0; q < cnt; q++)
         if (fabs(row[q].w) >= trunc_factor * mx) { row[c2++] = row[q]; kept += row[c2 - 1].w; }
      cnt = c2;
      if (kept != 0.0) { double sc = tot / kept; for (int q = 0; q < cnt; q++) row[q].w *= sc; }
   }
   if (pmax > 0 && cnt > pmax)
   {
      double tot = 0.0, kept = 0.0;
      for (int q = 0; q < cnt; q++) tot += row[q].w;
      pent_qsort_abs(row, cnt);
      cnt = pmax;
      /* the kept SET is what the sort decides; sums run in column order so that a
       * data-parallel top-k selection gives bit-identical weights */
      qsort(row, (size_t)cnt, sizeof(pent), pent_cmp_col);
      for (int q = 0; q < cnt; q++) kept += row[q].w;
      if (kept != 0.0) { double sc = tot / kept; for (int q = 0; q < cnt; q++) row[q].w *= sc; }
   }
   return cnt;
}

/* hypre_BoomerAMGBuildExtPIInterp + hypre_BoomerAMGInterpTruncation (interp type 6,
 * src/internal/amg.c:122-125,869,883-884), SURVEY App. A.6.  Rows column-sorted. */
orc_csr *
orc_interp_extpi(const orc_csr *A, const unsigned char *smask, const int *cf, int pmax,
                 double trunc_factor)
{
   return orc_interp_extpi_dof(A, smask, cf, pmax, trunc_factor, NULL);
}

/* dof != NULL (num_functions > 1): a weak connection to another function is dropped instead of
 * being lumped into the diagonal (hypre's ext+i: "if (num_functions == 1 || dof_func[i] ==
 * dof_func[i1]) diagonal += a_ii1"). */
orc_csr *
orc_interp_extpi_dof(const orc_csr *A, const unsigned char *smask, const int *cf, int pmax,
                     double trunc_factor, const int *dof)
{
   int  n    = A->nrows;
   int *cidx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   int  nc   = 0;
   for (int i = 0; i < n; i++) cidx[i] = (cf[i] == ORC_C_PT) ? nc++ : -1;

   int   *pm   = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* position in current row */
   int   *pst  = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* stamp for pm */
   int   *sfm  = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* strong-F stamp */
   for (int i = 0; i < n; i++) { pst[i] = -1; sfm[i] = -1; }
   int    cap  = 64;
   pent  *row  = (pent *)malloc(sizeof(pent) * (size_t)cap);
   int   *rfine = (int *)malloc(sizeof(int) * (size_t)cap);

   int     pcap = 4 * n + 16, pnnz = 0;
   int    *prow = (int *)calloc((size_t)n + 1, sizeof(int));
   int    *pcol = (int *)malloc(sizeof(int) * (size_t)pcap);
   double *pval = (double *)malloc(sizeof(double) * (size_t)pcap);

#define ENSURE_ROW(cnt)                                               \
   if ((cnt) >= cap)                                                  \
   {                                                                  \
      cap *= 2;                                                       \
      row   = (pent *)realloc(row, sizeof(pent) * (size_t)cap);       \
      rfine = (int *)realloc(rfine, sizeof(int) * (size_t)cap);       \
   }
#define ADD_CHAT(j)                    \
   if (pst[j] != i)                    \
   {                                   \
      ENSURE_ROW(cnt);                 \
      pst[j]     = i;                  \
      pm[j]      = cnt;                \
      rfine[cnt] = (j);                \
      row[cnt].c = cidx[j];            \
      row[cnt].w = 0.0;                \
      cnt++;                           \
   }

   for (int i = 0; i < n; i++)
   {
      int cnt = 0;
      if (cf[i] == ORC_C_PT)
      {
         row[0].c = cidx[i];
         row[0].w = 1.0;
         cnt      = 1;
      }
      else if (cf[i] == ORC_F_PT)
      {
         int k0 = A->rowptr[i], k1 = A->rowptr[i + 1];
         /* C-hat_i = C_i U (U_{k in F_i^s} C_k) */
         for (int k = k0; k < k1; k++)
         {
            if (!smask[k]) continue;
            int j = A->col[k];
            if (cf[j] == ORC_C_PT) { ADD_CHAT(j); }
            else if (cf[j] == ORC_F_PT)
            {
               sfm[j] = i;
               for (int kk = A->rowptr[j]; kk < A->rowptr[j + 1]; kk++)
               {
                  int m = A->col[kk];
                  if (smask[kk] && cf[m] == ORC_C_PT) { ADD_CHAT(m); }
               }
            }
         }
         double diagonal = 0.0;
         for (int k = k0; k < k1; k++)
            if (A->col[k] == i) diagonal = A->val[k];
         for (int k = k0; k < k1; k++)
         {
            int j = A->col[k];
            if (j == i) continue;
            double aij = A->val[k];
            if (pst[j] == i) row[pm[j]].w += aij;
            else if (sfm[j] == i)
            {
               double ajj = 0.0, sum = 0.0;
               int    j0 = A->rowptr[j], j1 = A->rowptr[j + 1];
               for (int kk = j0; kk < j1; kk++)
                  if (A->col[kk] == j) ajj = A->val[kk];
               double sgn = (ajj < 0.0) ? -1.0 : 1.0;
               for (int kk = j0; kk < j1; kk++)
               {
                  int m = A->col[kk];
                  if ((pst[m] == i || m == i) && sgn * A->val[kk] < 0.0) sum += A->val[kk];
               }
               if (sum != 0.0)
               {
                  double distribute = aij / sum;
                  for (int kk = j0; kk < j1; kk++)
                  {
                     int m = A->col[kk];
                     if (sgn * A->val[kk] < 0.0)
                     {
                        if (pst[m] == i) row[pm[m]].w += distribute * A->val[kk];
                        else if (m == i) diagonal += distribute * A->val[kk];
                     }
                  }
               }
               else
                  diagonal += aij;
            }
            else if (cf[j] != ORC_SF_PT && !(dof && dof[j] != dof[i]))
               diagonal += aij; /* weak connection lumped into the diagonal */
         }
         if (diagonal != 0.0)
            for (int q = 0; q < cnt; q++) row[q].w = row[q].w / (-diagonal);
         cnt = orc_truncate_row(row, cnt, pmax, trunc_factor);
      }
      qsort(row, (size_t)cnt, sizeof(pent), pent_cmp_col); /* storage order: by column */
      if (pnnz + cnt > pcap)
      {
         pcap = 2 * pcap + cnt;
         pcol = (int *)realloc(pcol, sizeof(int) * (size_t)pcap);
         pval = (double *)realloc(pval, sizeof(double) * (size_t)pcap);
      }
      for (int q = 0; q < cnt; q++) { pcol[pnnz] = row[q].c; pval[pnnz++] = row[q].w; }
      prow[i + 1] = pnnz;
   }
#undef ADD_CHAT
#undef ENSURE_ROW
   orc_csr *P = (orc_csr *)calloc(1, sizeof(orc_csr));
   P->nrows = n; P->ncols = nc; P->rowptr = prow; P->col = pcol; P->val = pval;
   free(cidx); free(pm); free(pst); free(sfm); free(row); free(rfine);
   return P;
}

/* Interpolation type 8, "standard" (reference name map src/internal/amg.c:258; the configuration examples/refOutput/ex8.txt:74 echoes
 * for its fifth variant, pinned at 6 iterations by ex8.txt:96).  hypre_BoomerAMGBuildStdInterp is not in the reference tree: this
 * restates the published algorithm (De Sterck, Falgout, Nolting, Yang, "Distance-two interpolation for parallel algebraic multigrid",
 * 2008, section 4.1) in the form without separation of weights.  The interpolatory set is the extended one,
 * C-hat_i = C_i U (U_{j in F_i^s} C_j); every strong F neighbour j of i is eliminated through its OWN equation,
 *    e_j = -sum_{m != j} a_jm e_m / a_jj,
 * which turns row i into a wider stencil a-hat (entries to C-hat_i, to i itself, and to points outside both), and direct
 * interpolation is applied to that stencil:
 *    w_ic = -alfa a-hat_ic / a-hat_ii,   alfa = (sum of a-hat_im over all m != i) / (sum of a-hat_ic over C-hat_i)   (1 if that is 0).
 * Order of the sums (part of the definition, the device kernel follows it): row i in storage order; a strong F neighbour's row in
 * storage order; the C-hat sum in discovery order of C-hat_i.  Truncation as for the other operators. */
orc_csr *
orc_interp_standard_dof(const orc_csr *A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor, const int *dof)
{
   int  n    = A->nrows;
   int *cidx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   int  nc   = 0;
   for (int i = 0; i < n; i++) cidx[i] = (cf[i] == ORC_C_PT) ? nc++ : -1;
   int *pm  = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* position in current row */
   int *pst = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* stamp for pm */
   for (int i = 0; i < n; i++) pst[i] = -1;
   int     cap  = 64;
   pent   *row  = (pent *)malloc(sizeof(pent) * (size_t)cap);
   int     pcap = 4 * n + 16, pnnz = 0;
   int    *prow = (int *)calloc((size_t)n + 1, sizeof(int));
   int    *pcol = (int *)malloc(sizeof(int) * (size_t)pcap);
   double *pval = (double *)malloc(sizeof(double) * (size_t)pcap);
#define ADD_CHAT(j)                                             \
   if (pst[j] != i)                                             \
   {                                                            \
      if (cnt >= cap)                                           \
      {                                                         \
         cap *= 2;                                              \
         row = (pent *)realloc(row, sizeof(pent) * (size_t)cap); \
      }                                                         \
      pst[j]     = i;                                           \
      pm[j]      = cnt;                                         \
      row[cnt].c = cidx[j];                                     \
      row[cnt].w = 0.0;                                         \
      cnt++;                                                    \
   }
   for (int i = 0; i < n; i++)
   {
      int cnt = 0;
      if (cf[i] == ORC_C_PT)
      {
         row[0].c = cidx[i];
         row[0].w = 1.0;
         cnt      = 1;
      }
      else if (cf[i] == ORC_F_PT)
      {
         int k0 = A->rowptr[i], k1 = A->rowptr[i + 1];
         for (int k = k0; k < k1; k++)
         {
            if (!smask[k]) continue;
            int j = A->col[k];
            if (cf[j] == ORC_C_PT) { ADD_CHAT(j); }
            else if (cf[j] == ORC_F_PT)
               for (int kk = A->rowptr[j]; kk < A->rowptr[j + 1]; kk++)
               {
                  int m = A->col[kk];
                  if (smask[kk] && cf[m] == ORC_C_PT) { ADD_CHAT(m); }
               }
         }
         double diagonal = 0.0, other = 0.0;
         for (int k = k0; k < k1; k++)
            if (A->col[k] == i) diagonal = A->val[k];
         for (int k = k0; k < k1; k++)
         {
            int j = A->col[k];
            if (j == i) continue;
            double aij = A->val[k];
            if (smask[k] && cf[j] == ORC_F_PT && !(dof && dof[j] != dof[i]))
            { /* a strong F neighbour: replaced by the rest of its own row */
               int    j0 = A->rowptr[j], j1 = A->rowptr[j + 1];
               double ajj = 0.0;
               for (int kk = j0; kk < j1; kk++)
                  if (A->col[kk] == j) ajj = A->val[kk];
               double distribute = aij / ajj;
               for (int kk = j0; kk < j1; kk++)
               {
                  int m = A->col[kk];
                  if (m == j) continue;
                  double t = A->val[kk] * distribute;
                  if (pst[m] == i) row[pm[m]].w -= t;
                  else if (m == i) diagonal -= t;
                  else other -= t;
               }
            }
            else if (pst[j] == i) row[pm[j]].w += aij;
            else other += aij;
         }
         double sum_C = 0.0, alfa = 1.0;
         for (int q = 0; q < cnt; q++) sum_C += row[q].w;
         double sum = sum_C + other;
         if (sum_C * diagonal != 0.0) alfa = sum / sum_C / diagonal;
         for (int q = 0; q < cnt; q++) row[q].w = -alfa * row[q].w;
         cnt = orc_truncate_row(row, cnt, pmax, trunc_factor);
      }
      qsort(row, (size_t)cnt, sizeof(pent), pent_cmp_col); /* storage order: by column */
      if (pnnz + cnt > pcap)
      {
         pcap = 2 * pcap + cnt;
         pcol = (int *)realloc(pcol, sizeof(int) * (size_t)pcap);
         pval = (double *)realloc(pval, sizeof(double) * (size_t)pcap);
      }
      for (int q = 0; q < cnt; q++) { pcol[pnnz] = row[q].c; pval[pnnz++] = row[q].w; }
      prow[i + 1] = pnnz;
   }
#undef ADD_CHAT
   orc_csr *P = (orc_csr *)calloc(1, sizeof(orc_csr));
   P->nrows = n; P->ncols = nc; P->rowptr = prow; P->col = pcol; P->val = pval;
   free(cidx); free(pm); free(pst); free(row);
   return P;
}

/* Interpolation type 17, "mm-ext+i" (reference name map src/internal/amg.c:267-268; the interpolation of all four pinned variants
 * of examples/refOutput/ex8.txt:26-78): hypre's matrix-matrix form of extended+i (Li, Sjogreen, Yang 2021, "A new class of AMG
 * interpolation methods based on matrix-matrix multiplications"; hypre par_mod_lr_interp.c, which is not in the reference tree).  It is
 * a DIFFERENT operator from the classical formula above: the weight a strong F neighbour k passes on is normalised over k's OWN strong C
 * neighbours (plus i), not over the interpolatory set of i, and entries are not filtered by sign -- which is what lets the whole
 * operator be written as sparse products:
 *    q_k    = sum of a_kl over the strong C neighbours l of k
 *    b_ik   = a_ik / (q_k + s_ki),  s_ki = a_ki when k depends strongly on i, else 0        (k a strong F neighbour of i)
 *    d_i    = a_ii + sum of the weak a_in + sum_k b_ik s_ki      (a strong F neighbour with q_k + s_ki = 0 is lumped like a weak one)
 *    W      = -D^-1 (I + B) A^s_FC                                (A^s_FC: the strong F-to-C entries)
 * then hypre_BoomerAMGInterpTruncation on the finished, column-sorted rows.  Order of every sum (the device reproduces it bit for bit):
 * d_i in the column order of row i; an output entry over k ascending (i itself in its place), i.e. the order in which the product
 * (I + B) A^s_FC enumerates its terms.  Entries towards special F points and other functions' unknowns are neither strong nor lumped. */
orc_csr *
orc_interp_mm_extpi_dof(const orc_csr *A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor, const int *dof)
{
   const int n    = A->nrows;
   int      *cidx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   int       nc   = 0;
   for (int i = 0; i < n; i++) cidx[i] = (cf[i] == ORC_C_PT) ? nc++ : -1;
   double *qk = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
   for (int k = 0; k < n; k++)
   {
      double s = 0.0;
      if (cf[k] == ORC_F_PT)
         for (int kk = A->rowptr[k]; kk < A->rowptr[k + 1]; kk++)
            if (smask[kk] && cf[A->col[kk]] == ORC_C_PT) s += A->val[kk];
      qk[k] = s;
   }
   double *acc   = (double *)calloc((size_t)(nc > 0 ? nc : 1), sizeof(double));
   int    *stamp = (int *)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
   int    *list  = (int *)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
   for (int j = 0; j < nc; j++) stamp[j] = -1;
   int     pcap = 8 * n + 16, pnnz = 0;
   int    *prow = (int *)calloc((size_t)n + 1, sizeof(int));
   int    *pcol = (int *)malloc(sizeof(int) * (size_t)pcap);
   double *pval = (double *)malloc(sizeof(double) * (size_t)pcap);
   for (int i = 0; i < n; i++)
   {
      int cnt = 0;
      if (cf[i] == ORC_C_PT)
      {
         list[0]      = cidx[i];
         acc[cidx[i]] = 1.0;
         cnt          = 1;
      }
      else if (cf[i] == ORC_F_PT)
      {
         double d = 0.0;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            if (A->col[k] == i) d = A->val[k];
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         { /* one pass in column order: the row's own strong C entries when the walk reaches i, a strong F neighbour's when it reaches it */
            const int    j   = A->col[k];
            const double aij = A->val[k];
            double       coef = 0.0;
            int          src  = -1;
            if (j == i) { coef = 1.0; src = i; }
            else if (smask[k] && cf[j] == ORC_F_PT)
            {
               double ski = 0.0;
               for (int kk = A->rowptr[j]; kk < A->rowptr[j + 1]; kk++)
                  if (A->col[kk] == i && smask[kk]) ski = A->val[kk];
               const double den = qk[j] + ski;
               if (den != 0.0)
               {
                  coef = aij / den;
                  src  = j;
                  d += coef * ski;
               }
               else
                  d += aij;
            }
            else if (smask[k] && cf[j] == ORC_C_PT) { /* reaches the row through src == i */ }
            else if (cf[j] != ORC_SF_PT && !(dof && dof[j] != dof[i]))
               d += aij; /* weak connection (or a strong one to a point that is neither C nor F): lumped */
            if (src < 0) continue;
            for (int kk = A->rowptr[src]; kk < A->rowptr[src + 1]; kk++)
            {
               const int l = A->col[kk];
               if (!smask[kk] || cf[l] != ORC_C_PT) continue;
               const int    c = cidx[l];
               const double t = coef * A->val[kk];
               if (stamp[c] != i)
               {
                  stamp[c]    = i;
                  acc[c]      = t;
                  list[cnt++] = c;
               }
               else
                  acc[c] += t;
            }
         }
         if (d != 0.0)
            for (int q = 0; q < cnt; q++) acc[list[q]] = acc[list[q]] / (-d);
      }
      /* storage order: by column */
      for (int a = 1; a < cnt; a++)
      {
         int v = list[a], b = a - 1;
         while (b >= 0 && list[b] > v) { list[b + 1] = list[b]; b--; }
         list[b + 1] = v;
      }
      if (pnnz + cnt > pcap)
      {
         pcap = 2 * pcap + cnt;
         pcol = (int *)realloc(pcol, sizeof(int) * (size_t)pcap);
         pval = (double *)realloc(pval, sizeof(double) * (size_t)pcap);
      }
      for (int q = 0; q < cnt; q++) { pcol[pnnz] = list[q]; pval[pnnz++] = acc[list[q]]; }
      prow[i + 1] = pnnz;
   }
   orc_csr *P = (orc_csr *)calloc(1, sizeof(orc_csr));
   P->nrows = n; P->ncols = nc; P->rowptr = prow; P->col = pcol; P->val = pval;
   free(cidx); free(qk); free(acc); free(stamp); free(list);
   orc_truncate_rows(P, pmax, trunc_factor);
   return P;
}

/* hypre_BoomerAMGBuildDirInterp with separation of weights (interp type 3, "direct_sep_weights" in
 * src/internal/amg.c:258-270; examples/ex8-amg-5.yml, pinned by examples/refOutput/ex8.txt:96) followed by
 * hypre_BoomerAMGInterpTruncation.  hypre itself is not in /root/reference: this restates the published
 * algorithm -- an F point interpolates from its strong C neighbours only,
 *    w_ij = -alpha a_ij / a_ii (a_ij < 0),  -beta a_ij / a_ii (a_ij > 0),
 *    alpha = sum of the negative off-diagonals of row i / sum of the negative a_ij over the strong C neighbours,
 *    beta likewise for the positive ones (1 when the interpolatory set has no entry of that sign).
 * Other functions' unknowns take no part in the row sums (dof != NULL). */
orc_csr *
orc_interp_direct_dof(const orc_csr *A, const unsigned char *smask, const int *cf, int pmax,
                      double trunc_factor, const int *dof)
{
   int  n    = A->nrows;
   int *cidx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   int  nc   = 0;
   for (int i = 0; i < n; i++) cidx[i] = (cf[i] == ORC_C_PT) ? nc++ : -1;
   int     cap = 64;
   pent   *row = (pent *)malloc(sizeof(pent) * (size_t)cap);
   int     pcap = 4 * n + 16, pnnz = 0;
   int    *prow = (int *)calloc((size_t)n + 1, sizeof(int));
   int    *pcol = (int *)malloc(sizeof(int) * (size_t)pcap);
   double *pval = (double *)malloc(sizeof(double) * (size_t)pcap);
   for (int i = 0; i < n; i++)
   {
      int cnt = 0;
      if (cf[i] == ORC_C_PT)
      {
         row[0].c = cidx[i];
         row[0].w = 1.0;
         cnt      = 1;
      }
      else if (cf[i] == ORC_F_PT)
      {
         int    k0 = A->rowptr[i], k1 = A->rowptr[i + 1];
         double diagonal = 0.0, sum_N_pos = 0.0, sum_N_neg = 0.0, sum_P_pos = 0.0, sum_P_neg = 0.0;
         if (k1 - k0 >= cap)
         {
            cap = 2 * (k1 - k0) + 16;
            row = (pent *)realloc(row, sizeof(pent) * (size_t)cap);
         }
         for (int k = k0; k < k1; k++)
            if (A->col[k] == i) diagonal = A->val[k];
         for (int k = k0; k < k1; k++)
         {
            int j = A->col[k];
            if (j == i) continue;
            double a = A->val[k];
            if (!(dof && dof[j] != dof[i]))
            {
               if (a > 0.0) sum_N_pos += a;
               else sum_N_neg += a;
            }
            if (smask[k] && cf[j] == ORC_C_PT)
            {
               row[cnt].c = cidx[j];
               row[cnt].w = a;
               cnt++;
               if (a > 0.0) sum_P_pos += a;
               else sum_P_neg += a;
            }
         }
         double alfa = 1.0, beta = 1.0;
         if (sum_P_neg != 0.0) alfa = sum_N_neg / sum_P_neg / diagonal;
         if (sum_P_pos != 0.0) beta = sum_N_pos / sum_P_pos / diagonal;
         for (int q = 0; q < cnt; q++) row[q].w *= (row[q].w > 0.0) ? -beta : -alfa;
         cnt = orc_truncate_row(row, cnt, pmax, trunc_factor);
      }
      qsort(row, (size_t)cnt, sizeof(pent), pent_cmp_col); /* storage order: by column */
      if (pnnz + cnt > pcap)
      {
         pcap = 2 * pcap + cnt;
         pcol = (int *)realloc(pcol, sizeof(int) * (size_t)pcap);
         pval = (double *)realloc(pval, sizeof(double) * (size_t)pcap);
      }
      for (int q = 0; q < cnt; q++) { pcol[pnnz] = row[q].c; pval[pnnz++] = row[q].w; }
      prow[i + 1] = pnnz;
   }
   orc_csr *P = (orc_csr *)calloc(1, sizeof(orc_csr));
   P->nrows = n; P->ncols = nc; P->rowptr = prow; P->col = pcol; P->val = pval;
   free(cidx); free(row);
   return P;
}

/* row-wise Gustavson product C = X*Y with column-sorted rows */
static orc_csr *
spgemm(const orc_csr *X, const orc_csr *Y)
{
   int     n = X->nrows, m = Y->ncols;
   int    *mark = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
   double *acc  = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
   int    *list = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
   for (int j = 0; j < m; j++) mark[j] = -1;
   int     cap = X->rowptr[n] * 4 + 16, nnz = 0;
   int    *cp  = (int *)calloc((size_t)n + 1, sizeof(int));
   int    *cj  = (int *)malloc(sizeof(int) * (size_t)cap);
   double *cv  = (double *)malloc(sizeof(double) * (size_t)cap);
   for (int i = 0; i < n; i++)
   {
      int cnt = 0;
      for (int k = X->rowptr[i]; k < X->rowptr[i + 1]; k++)
      {
         int    r = X->col[k];
         double a = X->val[k];
         for (int q = Y->rowptr[r]; q < Y->rowptr[r + 1]; q++)
         {
            int    j = Y->col[q];
            double t = a * Y->val[q];
            if (mark[j] != i) { mark[j] = i; acc[j] = t; list[cnt++] = j; }
            else acc[j] += t;
         }
      }
      /* sort the column list */
      for (int a = 1; a < cnt; a++)
      {
         int v = list[a], b = a - 1;
         while (b >= 0 && list[b] > v) { list[b + 1] = list[b]; b--; }
         list[b + 1] = v;
      }
      if (nnz + cnt > cap)
      {
         cap = 2 * cap + cnt;
         cj  = (int *)realloc(cj, sizeof(int) * (size_t)cap);
         cv  = (double *)realloc(cv, sizeof(double) * (size_t)cap);
      }
      for (int a = 0; a < cnt; a++) { cj[nnz] = list[a]; cv[nnz++] = acc[list[a]]; }
      cp[i + 1] = nnz;
   }
   free(mark); free(acc); free(list);
   orc_csr *C = (orc_csr *)calloc(1, sizeof(orc_csr));
   C->nrows = n; C->ncols = m; C->rowptr = cp; C->col = cj; C->val = cv;
   return C;
}

/* Galerkin product P^T A P (hypre_BoomerAMGBuildCoarseOperator; rap2/mod_rap2 flags at
 * src/internal/amg.c:946-950 select an equivalent two-SpGEMM formulation), App. A.7 */
orc_csr *
orc_rap(const orc_csr *A, const orc_csr *P)
{
   orc_csr *AP = spgemm(A, P);
   orc_csr *R  = orc_csr_transpose(P);
   orc_csr *Ac = spgemm(R, AP);
   orc_csr_free(AP);
   orc_csr_free(R);
   return Ac;
}


/* ------------------------------------------------------------------ aggressive coarsening
 * HYPRE_BoomerAMGSetAggNumLevels / SetNumPaths / SetAggInterpType as forwarded by src/internal/amg.c:938-944.  hypre is not in the
 * reference tree and no checked-in output uses these options: restated from the published method (Stueben 1999, "aggressive
 * coarsening" A1 / A2 and "multipass interpolation"; Yang 2010, "On long-range interpolation operators for aggressive coarsening").
 * PARITY UNPINNED.
 *
 * Second strength graph: C points i, j of a first coarsening are strongly n-connected when at least num_paths paths of length <= 2
 * lead from i to j along strong connections (the direct connection counts as one path, every intermediate point k with i -> k -> j
 * as one).  A second PMIS pass over that graph keeps a subset of the first pass's C points. */
orc_csr *
orc_second_strength(const orc_csr *A, const unsigned char *smask, const int *cf, int num_paths)
{
   const int n = A->nrows;
   int *c1 = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   int  n1 = 0;
   for (int i = 0; i < n; i++) c1[i] = (cf[i] == ORC_C_PT) ? n1++ : -1;
   /* S restricted to C rows (n1 x n, unit values) and S restricted to C columns (n x n1): paths of length two = their product */
   int nsr = 0, nsc = 0;
   for (int i = 0; i < n; i++)
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         if (smask[k]) { nsr += (c1[i] >= 0); nsc += (c1[A->col[k]] >= 0); }
   orc_csr *Sr = orc_csr_alloc(n1, n, nsr), *Sc = orc_csr_alloc(n, n1, nsc), *D = orc_csr_alloc(n1, n1, nsr);
   int a = 0, b = 0, d = 0;
   for (int i = 0; i < n; i++)
   {
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
      {
         if (!smask[k]) continue;
         const int j = A->col[k];
         if (c1[i] >= 0) { Sr->col[a] = j; Sr->val[a++] = 1.0; }
         if (c1[j] >= 0) { Sc->col[b] = c1[j]; Sc->val[b++] = 1.0; }
         if (c1[i] >= 0 && c1[j] >= 0) { D->col[d] = c1[j]; D->val[d++] = 1.0; }
      }
      if (c1[i] >= 0) { Sr->rowptr[c1[i] + 1] = a; D->rowptr[c1[i] + 1] = d; }
      Sc->rowptr[i + 1] = b;
   }
   orc_csr *T = spgemm(Sr, Sc); /* n1 x n1: number of two-step paths */
   /* S2 = entries of T + D with at least num_paths paths, diagonal dropped; rows column-sorted (both operands are) */
   int cap = T->rowptr[n1] + D->rowptr[n1];
   orc_csr *S2 = orc_csr_alloc(n1, n1, cap);
   int q = 0;
   for (int i = 0; i < n1; i++)
   {
      int t = T->rowptr[i], te = T->rowptr[i + 1], e = D->rowptr[i], ee = D->rowptr[i + 1];
      while (t < te || e < ee)
      {
         const int jt = (t < te) ? T->col[t] : 0x7fffffff, jd = (e < ee) ? D->col[e] : 0x7fffffff, j = jt < jd ? jt : jd;
         double cnt = 0.0;
         if (jt == j) cnt += T->val[t++];
         if (jd == j) cnt += D->val[e++];
         if (j != i && cnt >= (double)num_paths) { S2->col[q] = j; S2->val[q++] = cnt; }
      }
      S2->rowptr[i + 1] = q;
   }
   orc_csr_free(Sr); orc_csr_free(Sc); orc_csr_free(D); orc_csr_free(T);
   free(c1);
   return S2;
}

/* second coarsening of an aggressive level: PMIS over the second strength graph of cf's C points (every entry strong; tie-break
 * weights hash the C point's rank among them under the level salt + 64); C points that become F there become F points of the level.
 * A C point without any second-graph neighbour stays C. */
void
orc_coarsen_second_pass(const orc_csr *A, const unsigned char *smask, int num_paths, uint64_t seed, int level, int *cf)
{
   const int n = A->nrows;
   orc_csr  *S2 = orc_second_strength(A, smask, cf, num_paths);
   const int n1 = S2->nrows, nnz2 = S2->rowptr[n1];
   unsigned char *all = (unsigned char *)malloc((size_t)(nnz2 > 0 ? nnz2 : 1));
   memset(all, 1, (size_t)(nnz2 > 0 ? nnz2 : 1));
   int *cf2 = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
   orc_pmis(S2, all, seed, level + 64, 0, cf2);
   int q = 0;
   for (int i = 0; i < n; i++)
      if (cf[i] == ORC_C_PT)
      {
         if (cf2[q] == ORC_F_PT) cf[i] = ORC_F_PT;
         q++;
      }
   free(all); free(cf2);
   orc_csr_free(S2);
}

/* Multipass interpolation (hypre agg_interp_type 4).  Pass 0: C points (identity).  Pass 1: F points with a strong C neighbour,
 * direct interpolation w_ij = alfa_i a_ij over the strong C neighbours, alfa_i = -(sum of ALL off-diagonals of row i) /
 * (a_ii * sum over those neighbours) -- constants are reproduced.  Pass p >= 2: F points with a strong neighbour of pass p - 1
 * interpolate THROUGH those neighbours' rows: w_i = alfa_i sum_k a_ik w_k with the same alfa over the pass-(p-1) strong neighbours.
 * Points no pass reaches (and special F points) get empty rows.  Passes >= 2 are formed as sparse products M_p W (M_p: the scaled
 * strong pass-(p-1) entries of the pass-p rows) with the Galerkin product's own row-wise accumulation order, so that a device
 * implementation built on its SpGEMM kernel is bit-comparable. */
orc_csr *
orc_interp_multipass(const orc_csr *A, const unsigned char *smask, const int *cf)
{
   const int n = A->nrows;
   int *cidx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *pass = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   double *alfa = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
   int nc = 0;
   for (int i = 0; i < n; i++) { cidx[i] = (cf[i] == ORC_C_PT) ? nc++ : -1; pass[i] = (cf[i] == ORC_C_PT) ? 0 : -1; }
   /* pass numbers */
   int npass = 0;
   for (int p = 1;; p++)
   {
      int found = 0;
      for (int i = 0; i < n; i++)
      {
         if (pass[i] >= 0 || cf[i] != ORC_F_PT) continue;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            if (smask[k] && pass[A->col[k]] == p - 1) { pass[i] = -2; found++; break; } /* marked; committed below (synchronous rounds) */
      }
      for (int i = 0; i < n; i++)
         if (pass[i] == -2) pass[i] = p;
      if (!found) break;
      npass = p;
   }
   /* alfa_i for every interpolated F point */
   for (int i = 0; i < n; i++)
   {
      if (pass[i] < 1) continue;
      double diag = 0.0, sum_n = 0.0, sum_c = 0.0;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
      {
         const int j = A->col[k];
         if (j == i) { diag = A->val[k]; continue; }
         sum_n += A->val[k];
         if (smask[k] && pass[j] == pass[i] - 1) sum_c += A->val[k];
      }
      alfa[i] = (sum_c * diag != 0.0) ? -sum_n / (sum_c * diag) : 0.0;
   }
   /* W after pass 1: identity rows of the C points, direct rows of the pass-1 points */
   int nnz = 0;
   for (int i = 0; i < n; i++)
   {
      if (pass[i] == 0) nnz++;
      else if (pass[i] == 1)
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) nnz += (smask[k] && pass[A->col[k]] == 0);
   }
   orc_csr *W = orc_csr_alloc(n, nc, nnz);
   int q = 0;
   for (int i = 0; i < n; i++)
   {
      if (pass[i] == 0) { W->col[q] = cidx[i]; W->val[q++] = 1.0; }
      else if (pass[i] == 1)
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            if (smask[k] && pass[A->col[k]] == 0) { W->col[q] = cidx[A->col[k]]; W->val[q++] = alfa[i] * A->val[k]; }
      W->rowptr[i + 1] = q;
   }
   for (int p = 2; p <= npass; p++)
   {
      int mnz = 0;
      for (int i = 0; i < n; i++)
         if (pass[i] == p)
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) mnz += (smask[k] && pass[A->col[k]] == p - 1);
      orc_csr *M = orc_csr_alloc(n, n, mnz);
      int m = 0;
      for (int i = 0; i < n; i++)
      {
         if (pass[i] == p)
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
               if (smask[k] && pass[A->col[k]] == p - 1) { M->col[m] = A->col[k]; M->val[m++] = alfa[i] * A->val[k]; }
         M->rowptr[i + 1] = m;
      }
      orc_csr *T = spgemm(M, W);
      /* W := W + T (the rows of T are rows W does not have yet) */
      orc_csr *W2 = orc_csr_alloc(n, nc, W->rowptr[n] + T->rowptr[n]);
      int z = 0;
      for (int i = 0; i < n; i++)
      {
         const orc_csr *src = (pass[i] == p) ? T : W;
         for (int k = src->rowptr[i]; k < src->rowptr[i + 1]; k++) { W2->col[z] = src->col[k]; W2->val[z++] = src->val[k]; }
         W2->rowptr[i + 1] = z;
      }
      orc_csr_free(M); orc_csr_free(T); orc_csr_free(W);
      W = W2;
   }
   free(cidx); free(pass); free(alfa);
   return W;
}

/* hypre_BoomerAMGInterpTruncation applied to a finished interpolation (the aggressive levels' multipass rows): every row, in its
 * stored (column) order, goes through orc_truncate_row -- the same threshold / keep-the-largest / rescale steps and the same tie
 * order as the truncation inside the extended+i routine. */
void
orc_truncate_rows(orc_csr *P, int pmax, double trunc_factor)
{
   if (pmax <= 0 && trunc_factor <= 0.0) return;
   const int n = P->nrows;
   int cap = 64, q = 0;
   pent *row = (pent *)malloc(sizeof(pent) * (size_t)cap);
   int *nrp = (int *)calloc((size_t)n + 1, sizeof(int));
   for (int i = 0; i < n; i++)
   {
      int cnt = P->rowptr[i + 1] - P->rowptr[i];
      if (cnt > cap) { cap = 2 * cnt; row = (pent *)realloc(row, sizeof(pent) * (size_t)cap); }
      for (int k = 0; k < cnt; k++) { row[k].c = P->col[P->rowptr[i] + k]; row[k].w = P->val[P->rowptr[i] + k]; }
      cnt = orc_truncate_row(row, cnt, pmax, trunc_factor);
      qsort(row, (size_t)cnt, sizeof(pent), pent_cmp_col);
      for (int k = 0; k < cnt; k++) { P->col[q] = row[k].c; P->val[q++] = row[k].w; } /* (q never passes the read position) */
      nrp[i + 1] = q;
   }
   memcpy(P->rowptr, nrp, sizeof(int) * (size_t)(n + 1));
   free(nrp); free(row);
}

/* ------------------------------------------------------------------ Chebyshev smoother (relax type 16)
 * hypre_ParCSRRelax_Cheby_Setup / _Solve and hypre_ParCSRMaxEigEstimateCG as configured by
 * HYPRE_BoomerAMGSetCheby{Order,Fraction,EigEst,Variant,Scale} (reference src/internal/amg.c:886-890, cheby.c:15-20).
 * Restated from the published method: residual polynomial T_k((theta - t)/delta) / T_k(theta/delta) on
 * [lower, upper], upper = 1.1 * lambda_max, lower = lambda_min + fraction (upper - lambda_min), eigenvalues of
 * D^-1/2 A D^-1/2 (scale) estimated by eig_est CG / Lanczos steps (Gershgorin rows when eig_est = 0). */
static void
tridiag_extremes(int m, const double *d, const double *e, double *lo, double *hi)
{ /* extreme eigenvalues of the symmetric tridiagonal (d, e) by bisection on the Sturm count */
   double g0 = d[0], g1 = d[0];
   for (int i = 0; i < m; i++)
   {
      double r = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i < m - 1 ? fabs(e[i]) : 0.0);
      if (d[i] - r < g0) g0 = d[i] - r;
      if (d[i] + r > g1) g1 = d[i] + r;
   }
   for (int which = 0; which < 2; which++)
   { /* which 0: smallest (count(x) >= 1), 1: largest (count(x) >= m) */
      double a = g0, b = g1;
      for (int it = 0; it < 200; it++)
      {
         double x = 0.5 * (a + b), q = d[0] - x;
         int    cnt = (q < 0.0);
         for (int i = 1; i < m; i++)
         {
            if (q == 0.0) q = 1e-300;
            q = d[i] - x - e[i - 1] * e[i - 1] / q;
            cnt += (q < 0.0);
         }
         if (cnt >= (which ? m : 1)) b = x; else a = x;
      }
      if (which) *hi = 0.5 * (a + b); else *lo = 0.5 * (a + b);
   }
}

void
orc_cheby_setup(const orc_csr *A, int order, int eig_est, int variant, int scale, double fraction, uint64_t seed, int level,
                double *ds, double coefs[5], double *max_eig_out, double *min_eig_out)
{
   (void)variant;
   const int n = A->nrows;
   if (order > 4) order = 4;
   if (order < 1) order = 1;
   for (int i = 0; i < n; i++)
   {
      double d = 1.0;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         if (A->col[k] == i) d = A->val[k];
      ds[i] = scale ? 1.0 / sqrt(fabs(d)) : 1.0;
   }
   double max_eig = 0.0, min_eig = 0.0;
   if (eig_est > 0)
   {
      double *r = (double *)malloc(sizeof(double) * (size_t)n), *p = (double *)malloc(sizeof(double) * (size_t)n);
      double *s = (double *)malloc(sizeof(double) * (size_t)n), *t = (double *)malloc(sizeof(double) * (size_t)n);
      double  td[64], te[64];
      for (int i = 0; i < n; i++) p[i] = r[i] = pmis_rand(seed, 1000 + level, i);
      double gamma = orc_dot(n, r, r), beta = 1.0, alpha_old = 1.0;
      int    m = 0;
      const int steps = eig_est < 60 ? eig_est : 60;
      while (m < steps && gamma > 0.0)
      {
         for (int i = 0; i < n; i++) t[i] = ds[i] * p[i];
         orc_spmv(A, 1.0, t, 0.0, s);
         for (int i = 0; i < n; i++) s[i] *= ds[i];
         const double sp = orc_dot(n, s, p);
         if (sp == 0.0) break;
         const double alpha = gamma / sp;
         td[m] = 1.0 / alpha + (m > 0 ? beta / alpha_old : 0.0);
         if (m > 0) te[m - 1] = sqrt(beta) / alpha_old;
         for (int i = 0; i < n; i++) r[i] -= alpha * s[i];
         const double gnew = orc_dot(n, r, r);
         beta      = gnew / gamma;
         gamma     = gnew;
         alpha_old = alpha;
         for (int i = 0; i < n; i++) p[i] = r[i] + beta * p[i];
         m++;
      }
      if (m > 0) tridiag_extremes(m, td, te, &min_eig, &max_eig);
      free(r); free(p); free(s); free(t);
   }
   else
   { /* Gershgorin: largest scaled absolute row sum */
      for (int i = 0; i < n; i++)
      {
         double rs = 0.0;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) rs += fabs(A->val[k]) * ds[i] * ds[A->col[k]];
         if (rs > max_eig) max_eig = rs;
      }
   }
   if (min_eig < 0.0) min_eig = 0.0;
   const double upper = 1.1 * max_eig, lower = min_eig + fraction * (upper - min_eig);
   const double th = 0.5 * (upper + lower), de = 0.5 * (upper - lower);
   double den;
   for (int i = 0; i < 5; i++) coefs[i] = 0.0;
   switch (order)
   { /* q(t) with 1 - t q(t) = T_k((th - t)/de) / T_k(th/de) */
      case 1: coefs[0] = 1.0 / th; break;
      case 2:
         den      = de * de - 2.0 * th * th;
         coefs[0] = -4.0 * th / den;
         coefs[1] = 2.0 / den;
         break;
      case 3:
         den      = 3.0 * de * de * th - 4.0 * th * th * th;
         coefs[0] = (3.0 * de * de - 12.0 * th * th) / den;
         coefs[1] = 12.0 * th / den;
         coefs[2] = -4.0 / den;
         break;
      default:
         den      = de * de * de * de - 8.0 * de * de * th * th + 8.0 * th * th * th * th;
         coefs[0] = (32.0 * th * th * th - 16.0 * de * de * th) / den;
         coefs[1] = (8.0 * de * de - 48.0 * th * th) / den;
         coefs[2] = 32.0 * th / den;
         coefs[3] = -8.0 / den;
         break;
   }
   if (max_eig_out) *max_eig_out = max_eig;
   if (min_eig_out) *min_eig_out = min_eig;
}

/* u += D^-1/2 q(D^-1/2 A D^-1/2) D^-1/2 (f - A u), q by Horner; r, v, w: work vectors of length n */
void
orc_cheby_apply(const orc_csr *A, int order, int scale, const double *ds, const double coefs[5], const double *f, double *u, double *r,
                double *v, double *w)
{
   (void)scale; /* ds is all ones without scaling */
   const int n = A->nrows;
   if (order > 4) order = 4;
   if (order < 1) order = 1;
   memcpy(r, f, sizeof(double) * (size_t)n);
   orc_spmv(A, -1.0, u, 1.0, r);
   for (int i = 0; i < n; i++) r[i] *= ds[i];
   for (int i = 0; i < n; i++) w[i] = coefs[order - 1] * r[i];
   for (int c = order - 2; c >= 0; c--)
   {
      for (int i = 0; i < n; i++) v[i] = ds[i] * w[i];
      orc_spmv(A, 1.0, v, 0.0, w);
      for (int i = 0; i < n; i++) w[i] = coefs[c] * r[i] + ds[i] * w[i];
   }
   for (int i = 0; i < n; i++) u[i] += ds[i] * w[i];
}

/* --------------------------------------------------------------- hierarchy */


/* ------------------------------------------------------------------ ILU(0), block Jacobi
 * hypre's "bj-iluk" with fill level 0 and no local reordering, as the reference configures it
 * (src/internal/ilu.c:15-28 defaults, :63-115 setter sequence; smoother form amg.c:899-921).
 * hypre is not in the reference tree: this is the textbook IKJ ILU(0) (Saad, Iterative Methods,
 * Alg. 10.4) on the diagonal blocks of the row partition, rows in natural order.  PARITY
 * UNPINNED: no checked-in reference output uses ILU on data that is present. */
struct orc_ilu {
   int      n;
   orc_csr *LU;   /* pattern of the diagonal blocks; strict lower part = L (unit diagonal), rest = U */
   int     *diag; /* position of the diagonal entry of every row */
   int      tri_solve, lower_it, upper_it;
   double  *y, *w;
};

void orc_ilu_free(orc_ilu *F);

orc_ilu *
orc_ilu0_setup(const orc_csr *A, int nparts, const int64_t *part, int tri_solve, int lower_it, int upper_it)
{
   int      n   = A->nrows;
   int64_t  one[2] = {0, n};
   if (nparts <= 0 || !part) { nparts = 1; part = one; }
   orc_ilu *F = (orc_ilu *)calloc(1, sizeof(orc_ilu));
   F->n = n; F->tri_solve = tri_solve; F->lower_it = lower_it < 1 ? 1 : lower_it; F->upper_it = upper_it < 1 ? 1 : upper_it;
   int *blk_lo = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *blk_hi = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   for (int p = 0; p < nparts; p++)
      for (int64_t i = part[p]; i < part[p + 1]; i++) { blk_lo[i] = (int)part[p]; blk_hi[i] = (int)part[p + 1]; }
   int nnz = 0;
   for (int i = 0; i < n; i++)
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) nnz += (A->col[k] >= blk_lo[i] && A->col[k] < blk_hi[i]);
   F->LU   = orc_csr_alloc(n, n, nnz);
   F->diag = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
   int q = 0;
   for (int i = 0; i < n; i++)
   {
      F->LU->rowptr[i] = q;
      F->diag[i]       = -1;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         if (A->col[k] >= blk_lo[i] && A->col[k] < blk_hi[i])
         {
            if (A->col[k] == i) F->diag[i] = q;
            F->LU->col[q] = A->col[k];
            F->LU->val[q] = A->val[k];
            q++;
         }
   }
   F->LU->rowptr[n] = q;
   free(blk_lo); free(blk_hi);
   const int *rp = F->LU->rowptr, *cj = F->LU->col;
   double    *v  = F->LU->val;
   for (int i = 0; i < n; i++)
   {
      if (F->diag[i] < 0) { orc_ilu_free(F); return NULL; } /* structurally missing diagonal */
      for (int kk = rp[i]; kk < F->diag[i]; kk++)
      {
         const int    k   = cj[kk];
         const double lik = v[kk] / v[F->diag[k]];
         v[kk]            = lik;
         int pi = kk + 1;
         for (int jj = F->diag[k] + 1; jj < rp[k + 1]; jj++)
         {
            const int j = cj[jj];
            while (pi < rp[i + 1] && cj[pi] < j) pi++;
            if (pi == rp[i + 1]) break;
            if (cj[pi] == j) v[pi] -= lik * v[jj];
         }
      }
      if (v[F->diag[i]] == 0.0) { orc_ilu_free(F); return NULL; } /* zero pivot */
   }
   F->y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
   F->w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
   return F;
}

void
orc_ilu_free(orc_ilu *F)
{
   if (!F) return;
   orc_csr_free(F->LU);
   free(F->diag); free(F->y); free(F->w);
   free(F);
}

const orc_csr *orc_ilu_factors(const orc_ilu *F) { return F->LU; }

/* z = U^{-1} L^{-1} r.  tri_solve 1: exact substitutions; 0: lower_it / upper_it Jacobi
 * iterations from a zero guess on (I + L~) y = r and (D + U~) z = y (ilu.c:21-23). */
void
orc_ilu_apply(orc_ilu *F, const double *r, double *z)
{
   const int     n  = F->n;
   const int    *rp = F->LU->rowptr, *cj = F->LU->col, *dg = F->diag;
   const double *v  = F->LU->val;
   double       *y  = F->y, *w = F->w;
   if (F->tri_solve)
   {
      for (int i = 0; i < n; i++)
      {
         double s = r[i];
         for (int k = rp[i]; k < dg[i]; k++) s -= v[k] * y[cj[k]];
         y[i] = s;
      }
      for (int i = n - 1; i >= 0; i--)
      {
         double s = y[i];
         for (int k = dg[i] + 1; k < rp[i + 1]; k++) s -= v[k] * z[cj[k]];
         z[i] = s / v[dg[i]];
      }
      return;
   }
   memcpy(y, r, sizeof(double) * (size_t)n); /* first iteration from y = 0 */
   for (int it = 1; it < F->lower_it; it++)
   {
      for (int i = 0; i < n; i++)
      {
         double s = r[i];
         for (int k = rp[i]; k < dg[i]; k++) s -= v[k] * y[cj[k]];
         w[i] = s;
      }
      memcpy(y, w, sizeof(double) * (size_t)n);
   }
   for (int i = 0; i < n; i++) z[i] = y[i] / v[dg[i]];
   for (int it = 1; it < F->upper_it; it++)
   {
      for (int i = 0; i < n; i++)
      {
         double s = y[i];
         for (int k = dg[i] + 1; k < rp[i + 1]; k++) s -= v[k] * z[cj[k]];
         w[i] = s / v[dg[i]];
      }
      memcpy(z, w, sizeof(double) * (size_t)n);
   }
}

/* hypre_ILUSolve as an iteration: x += M^{-1} (b - A x), iters times; tmp, cor length n */
static void
ilu_iterate(orc_ilu *F, const orc_csr *A, int iters, const double *b, double *x, double *tmp, double *cor)
{
   const int n = A->nrows;
   for (int it = 0; it < iters; it++)
   {
      memcpy(tmp, b, sizeof(double) * (size_t)n);
      orc_spmv(A, -1.0, x, 1.0, tmp);
      orc_ilu_apply(F, tmp, cor);
      for (int i = 0; i < n; i++) x[i] += cor[i];
   }
}

struct orc_mgr;
static void mgr_solve(struct orc_mgr *M, const double *b, double *x);
static void mgr_free(struct orc_mgr *M);
static int  mgr_rebind(struct orc_mgr *M, const orc_csr *A);

struct orc_amg {
   orc_amg_params p;
   int            nlev;
   orc_csr      **A, **P, **R;
   int          **cf;
   double       **l1d, **l1u; /* l1 vectors for down / up relax types */
   double       **f, **u, **tmp, **cor;
   double       **cheb_ds, **cheb_w2, **cheb_w3; /* relax type 16 */
   double        (*cheb_coef)[5];
   double        *dense;      /* coarsest dense copy */
   /* complex smoother (src/internal/amg.c:899-921): ILU on the first smooth_levels levels */
   orc_ilu      **ilu;
   int            smooth_levels, smooth_sweeps;
   int            nblk;       /* row blocks (orc_amg_params.blocks); 1 = none */
   int64_t      **bpart;      /* per level: nblk+1 row starts (coarse levels: through the C points) */
   struct orc_mgr *mgr;       /* handle made by orc_precond_mgr: the "hierarchy" is one MGR cycle */
   orc_ilu       *ilu_only;   /* handle made by orc_precond_ilu: the "hierarchy" is one ILU solve */
   const orc_csr *ilu_A;
   int            ilu_max_iter;
};

static int
l1_option_for(int relax_type)
{
   return (relax_type == 18) ? 1 : 4;
}

/* hypre_BoomerAMGSetup reached from src/internal/precon.c:107; level loop per SURVEY
 * App. A.2 (stop when rows <= max_coarse_size, at max_levels, or coarsening stalls). */
orc_amg *
orc_amg_setup(const orc_csr *A0, const orc_amg_params *p)
{
   return orc_amg_setup_dof(A0, p, NULL);
}

orc_amg *
orc_amg_setup_dof(const orc_csr *A0, const orc_amg_params *p, const int *dof0)
{
   orc_amg *h = (orc_amg *)calloc(1, sizeof(orc_amg));
   h->p       = *p;
   int maxl   = p->max_levels > 0 ? p->max_levels : 1;
   h->A   = (orc_csr **)calloc((size_t)maxl, sizeof(void *));
   h->P   = (orc_csr **)calloc((size_t)maxl, sizeof(void *));
   h->R   = (orc_csr **)calloc((size_t)maxl, sizeof(void *));
   h->cf  = (int **)calloc((size_t)maxl, sizeof(void *));
   h->l1d = (double **)calloc((size_t)maxl, sizeof(void *));
   h->l1u = (double **)calloc((size_t)maxl, sizeof(void *));
   h->f   = (double **)calloc((size_t)maxl, sizeof(void *));
   h->u   = (double **)calloc((size_t)maxl, sizeof(void *));
   h->tmp = (double **)calloc((size_t)maxl, sizeof(void *));
   h->cheb_ds   = (double **)calloc((size_t)maxl, sizeof(void *));
   h->cheb_w2   = (double **)calloc((size_t)maxl, sizeof(void *));
   h->cheb_w3   = (double **)calloc((size_t)maxl, sizeof(void *));
   h->cheb_coef = (double (*)[5])calloc((size_t)maxl, sizeof(double[5]));
   h->nblk  = p->blocks > 1 ? p->blocks : 1;
   h->bpart = (int64_t **)calloc((size_t)maxl, sizeof(void *));
   if (h->nblk > 1)
   { /* level 0: the caller's starts, or hypre's even split (hypre_GeneratePartitioning) */
      h->bpart[0] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(h->nblk + 1));
      for (int q = 0; q <= h->nblk; q++)
         h->bpart[0][q] = p->block_part ? p->block_part[q] : (int64_t)(((__int128)q * A0->nrows) / h->nblk);
   }
   h->p.block_part = NULL; /* (the caller's array is not kept) */
   /* own copy of level 0 */
   {
      int nnz = A0->rowptr[A0->nrows];
      h->A[0] = orc_csr_alloc(A0->nrows, A0->ncols, nnz);
      memcpy(h->A[0]->rowptr, A0->rowptr, sizeof(int) * (size_t)(A0->nrows + 1));
      memcpy(h->A[0]->col, A0->col, sizeof(int) * (size_t)nnz);
      memcpy(h->A[0]->val, A0->val, sizeof(double) * (size_t)nnz);
   }
   int lvl = 0;
   int not_finished = (h->A[0]->nrows > p->max_coarse_size) && (maxl > 1);
   /* function of every unknown on the current level (num_functions > 1), inherited by C points */
   int *dof = NULL;
   if (p->num_functions > 1)
   {
      dof = (int *)malloc(sizeof(int) * (size_t)(A0->nrows > 0 ? A0->nrows : 1));
      for (int i = 0; i < A0->nrows; i++) dof[i] = dof0 ? dof0[i] : i % p->num_functions;
   }
   while (not_finished)
   {
      const orc_csr *A   = h->A[lvl];
      int            n   = A->nrows;
      int            nnz = A->rowptr[n];
      unsigned char *sm  = (unsigned char *)malloc((size_t)(nnz > 0 ? nnz : 1));
      int           *cf  = (int *)malloc(sizeof(int) * (size_t)n);
      orc_strength_dof(A, p->strong_th, p->max_row_sum, dof, sm);
      if (p->pmis_rng == 1)
      { /* hypre's per-rank Park-Miller stream, the row blocks being the ranks */
         double *rnd = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
         orc_pmis_hypre_stream(n, h->nblk, h->bpart[lvl], rnd);
         if (p->coarsen_type == 8) pmis_core_ranks(A, sm, rnd, cf, p->pmis_rng == 1 ? h->nblk : 1, h->bpart[lvl]);
         else hmis_core(A, sm, h->nblk, h->bpart[lvl], rnd, cf);
         free(rnd);
      }
      else if (p->coarsen_type == 8)
         orc_pmis(A, sm, p->seed, lvl, 0, cf);
      else
         orc_hmis_blocks(A, sm, h->nblk, h->bpart[lvl], p->seed, lvl, cf);
      const int aggressive = lvl < p->agg_num_levels;
      if (aggressive) orc_coarsen_second_pass(A, sm, p->agg_num_paths > 0 ? p->agg_num_paths : 1, p->seed, lvl, cf);
      int nc = 0;
      for (int i = 0; i < n; i++) nc += (cf[i] == ORC_C_PT);
      if (nc == 0 || nc == n || nc < p->min_coarse_size)
      {
         free(sm); free(cf);
         break;
      }
      h->cf[lvl] = cf;
      h->P[lvl]  = aggressive ? orc_interp_multipass(A, sm, cf)  /* (truncated below) */
                   : (p->interp_type == 3)  ? orc_interp_direct_dof(A, sm, cf, p->pmax, p->trunc_factor, dof)
                   : (p->interp_type == 8)  ? orc_interp_standard_dof(A, sm, cf, p->pmax, p->trunc_factor, dof)
                   : (p->interp_type == 17) ? orc_interp_mm_extpi_dof(A, sm, cf, p->pmax, p->trunc_factor, dof)
                                            : orc_interp_extpi_dof(A, sm, cf, p->pmax, p->trunc_factor, dof);
      if (aggressive) orc_truncate_rows(h->P[lvl], p->agg_pmax, p->agg_trunc_factor);
      if (dof)
      { /* coarse unknowns keep the function of their fine C point */
         int q = 0;
         for (int i = 0; i < n; i++)
            if (cf[i] == ORC_C_PT) dof[q++] = dof[i];
      }
      h->R[lvl]  = orc_csr_transpose(h->P[lvl]);
      free(sm);
      if (h->nblk > 1)
      { /* coarse ids ascend with the fine ids of the C points: block q keeps a contiguous range (a rank's coarse rows) */
         int64_t *np_ = (int64_t *)malloc(sizeof(int64_t) * (size_t)(h->nblk + 1));
         int64_t  c = 0;
         int      q = 0;
         for (int64_t i = 0; i <= n; i++)
         {
            while (q <= h->nblk && h->bpart[lvl][q] == i) np_[q++] = c;
            if (i < n && cf[i] == ORC_C_PT) c++;
         }
         h->bpart[lvl + 1] = np_;
      }
      h->A[lvl + 1] = orc_rap(A, h->P[lvl]);
      lvl++;
      if (lvl >= maxl - 1 || nc <= p->max_coarse_size) not_finished = 0;
   }
   free(dof);
   h->nlev = lvl + 1;
   for (int l = 0; l < h->nlev; l++)
   {
      int n     = h->A[l]->nrows;
      h->l1d[l] = (double *)malloc(sizeof(double) * (size_t)n);
      h->l1u[l] = (double *)malloc(sizeof(double) * (size_t)n);
      orc_l1_norms_blocks(h->A[l], l1_option_for(p->relax_down), h->nblk, h->bpart[l], h->l1d[l]);
      orc_l1_norms_blocks(h->A[l], l1_option_for(p->relax_up), h->nblk, h->bpart[l], h->l1u[l]);
      h->f[l]   = (double *)calloc((size_t)n, sizeof(double));
      h->u[l]   = (double *)calloc((size_t)n, sizeof(double));
      h->tmp[l] = (double *)calloc((size_t)n, sizeof(double));
      if (p->relax_down == 16 || p->relax_up == 16 || p->relax_coarse == 16)
      {
         h->cheb_ds[l] = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
         h->cheb_w2[l] = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
         h->cheb_w3[l] = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
         orc_cheby_setup(h->A[l], p->cheby_order, p->cheby_eig_est, p->cheby_variant, p->cheby_scale, p->cheby_fraction, p->seed, l,
                         h->cheb_ds[l], h->cheb_coef[l], NULL, NULL);
      }
   }
   if (p->relax_coarse == 9)
   {
      int n    = h->A[h->nlev - 1]->nrows;
      h->dense = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
   }
   return h;
}

void
orc_amg_free(orc_amg *h)
{
   if (!h) return;
   if (h->mgr)
   {
      mgr_free(h->mgr);
      free(h);
      return;
   }
   if (h->ilu_only)
   {
      orc_ilu_free(h->ilu_only);
      free(h);
      return;
   }
   for (int l = 0; l < h->nlev; l++)
   {
      if (h->ilu && h->ilu[l]) orc_ilu_free(h->ilu[l]);
      if (h->cor && h->cor[l]) free(h->cor[l]);
      orc_csr_free(h->A[l]);
      orc_csr_free(h->P[l]);
      orc_csr_free(h->R[l]);
      free(h->cf[l]); free(h->l1d[l]); free(h->l1u[l]);
      free(h->f[l]); free(h->u[l]); free(h->tmp[l]);
      free(h->cheb_ds[l]); free(h->cheb_w2[l]); free(h->cheb_w3[l]);
      if (h->bpart) free(h->bpart[l]);
   }
   free(h->bpart);
   free(h->A); free(h->P); free(h->R); free(h->cf); free(h->l1d); free(h->l1u);
   free(h->f); free(h->u); free(h->tmp); free(h->dense);
   free(h->ilu); free(h->cor);
   free(h->cheb_ds); free(h->cheb_w2); free(h->cheb_w3); free(h->cheb_coef);
   free(h);
}

/* A hierarchy kept for a later system of a sequence (preconditioner.reuse, reference src/HYPREDRV.c:3010-3020):
 * hypre's BoomerAMGSolve takes level 0 from the matrix of the call and everything else -- smoother divisors,
 * transfer operators, coarse levels -- from the setup.  Returns 0 on success. */
int
orc_amg_rebind_level0(orc_amg *h, const orc_csr *A)
{
   if (h->ilu_only) { h->ilu_A = A; return 0; }
   if (h->mgr) return mgr_rebind(h->mgr, A);
   if (A->nrows != h->A[0]->nrows || A->ncols != h->A[0]->ncols) return 1;
   int      nnz = A->rowptr[A->nrows];
   orc_csr *C   = orc_csr_alloc(A->nrows, A->ncols, nnz);
   memcpy(C->rowptr, A->rowptr, sizeof(int) * (size_t)(A->nrows + 1));
   memcpy(C->col, A->col, sizeof(int) * (size_t)nnz);
   memcpy(C->val, A->val, sizeof(double) * (size_t)nnz);
   orc_csr_free(h->A[0]);
   h->A[0] = C;
   return 0;
}

int
orc_amg_set_ilu_smoother(orc_amg *h, int num_levels, int num_sweeps, int nparts, const int64_t *part, int tri_solve,
                         int lower_it, int upper_it)
{
   if (h->ilu_only) return 1;
   /* hypre smooths levels < smooth_num_levels that still have a coarser level below them */
   int K = num_levels < h->nlev - 1 ? num_levels : h->nlev - 1;
   if (K < 0) K = 0;
   h->ilu = (orc_ilu **)calloc((size_t)(h->nlev > 0 ? h->nlev : 1), sizeof(void *));
   h->cor = (double **)calloc((size_t)(h->nlev > 0 ? h->nlev : 1), sizeof(void *));
   int64_t  one[2] = {0, h->A[0]->nrows};
   if (nparts <= 0 || !part) { nparts = 1; part = one; }
   int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nparts + 1));
   memcpy(cur, part, sizeof(int64_t) * (size_t)(nparts + 1));
   for (int l = 0; l < K; l++)
   {
      h->ilu[l] = orc_ilu0_setup(h->A[l], nparts, cur, tri_solve, lower_it, upper_it);
      if (!h->ilu[l]) { free(cur); return 2; }
      h->cor[l] = (double *)calloc((size_t)h->A[l]->nrows, sizeof(double));
      /* coarse ids ascend with the fine ids of the C points: block p keeps a contiguous range */
      int64_t c = 0, p = 0;
      for (int64_t i = 0; i <= h->A[l]->nrows; i++)
      {
         while (p <= nparts && cur[p] == i) { cur[p] = c; p++; }
         if (i < h->A[l]->nrows && h->cf[l][i] > 0) c++;
      }
   }
   free(cur);
   h->smooth_levels = K;
   h->smooth_sweeps = num_sweeps < 1 ? 1 : num_sweeps;
   return 0;
}

orc_amg *
orc_precond_ilu(const orc_csr *A, int max_iter, int nparts, const int64_t *part, int tri_solve, int lower_it, int upper_it)
{
   orc_ilu *F = orc_ilu0_setup(A, nparts, part, tri_solve, lower_it, upper_it);
   if (!F) return NULL;
   orc_amg *h      = (orc_amg *)calloc(1, sizeof(orc_amg));
   h->ilu_only     = F;
   h->ilu_A        = A;
   h->ilu_max_iter = max_iter < 1 ? 1 : max_iter;
   return h;
}

int            orc_amg_num_levels(const orc_amg *h) { return h->nlev; }
const orc_csr *orc_amg_A(const orc_amg *h, int l) { return h->A[l]; }
const orc_csr *orc_amg_P(const orc_amg *h, int l) { return h->P[l]; }
const int     *orc_amg_cf(const orc_amg *h, int l) { return h->cf[l]; }
const double  *orc_amg_l1(const orc_amg *h, int l, int which) { return which ? h->l1u[l] : h->l1d[l]; }
const int64_t *orc_amg_block_part(const orc_amg *h, int l) { return h->bpart ? h->bpart[l] : NULL; } /* NULL: one block */

double
orc_amg_operator_complexity(const orc_amg *h)
{
   double s = 0.0;
   for (int l = 0; l < h->nlev; l++) s += (double)h->A[l]->rowptr[h->A[l]->nrows];
   return s / (double)h->A[0]->rowptr[h->A[0]->nrows];
}

double
orc_amg_grid_complexity(const orc_amg *h)
{
   double s = 0.0;
   for (int l = 0; l < h->nlev; l++) s += (double)h->A[l]->nrows;
   return s / (double)h->A[0]->nrows;
}

/* one relaxation sweep of the given type on level l (type 16: the Chebyshev polynomial smoother) */
static void
level_relax(orc_amg *h, int l, int type, const double *l1, const double *b, double *x)
{
   if (type == 16)
      orc_cheby_apply(h->A[l], h->p.cheby_order, h->p.cheby_scale, h->cheb_ds[l], h->cheb_coef[l], b, x, h->tmp[l], h->cheb_w2[l], h->cheb_w3[l]);
   else orc_relax_blocks(h->A[l], l1, type, h->p.relax_weight, b, x, h->tmp[l], h->nblk, h->bpart[l]);
}

static void
coarse_solve(orc_amg *h, int l, const double *b, double *x)
{
   const orc_csr *A = h->A[l];
   int            n = A->nrows;
   if (h->p.relax_coarse == 9)
   {
      memset(h->dense, 0, sizeof(double) * (size_t)n * (size_t)n);
      for (int i = 0; i < n; i++)
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
            h->dense[(size_t)i * n + A->col[k]] = A->val[k];
      memcpy(x, b, sizeof(double) * (size_t)n);
      orc_gselim(h->dense, x, n);
   }
   else
   {
      for (int s = 0; s < h->p.sweeps_coarse; s++)
         level_relax(h, l, h->p.relax_coarse, h->l1d[l], b, x);
   }
}

/* hypre_BoomerAMGCycle, cycle type 1, reached through
 * HYPRE_BoomerAMGSolve at src/internal/precon.c:108 with max_iter 1, tol 0
 * (src/internal/amg.c:224-226). */
void
orc_amg_vcycle(orc_amg *h, const double *b, double *x)
{
   if (h->mgr)
   {
      mgr_solve(h->mgr, b, x);
      return;
   }
   if (h->ilu_only)
   { /* preconditioner: ilu (precon.c op table): max_iter iterations of x += M^{-1}(b - A x) */
      const int n   = h->ilu_A->nrows;
      double   *tmp = (double *)malloc(sizeof(double) * (size_t)n), *cor = (double *)malloc(sizeof(double) * (size_t)n);
      ilu_iterate(h->ilu_only, h->ilu_A, h->ilu_max_iter, b, x, tmp, cor);
      free(tmp); free(cor);
      return;
   }
   int L = h->nlev;
   memcpy(h->f[0], b, sizeof(double) * (size_t)h->A[0]->nrows);
   memcpy(h->u[0], x, sizeof(double) * (size_t)h->A[0]->nrows);
   for (int l = 0; l < L - 1; l++)
   {
      const orc_csr *A = h->A[l];
      int            n = A->nrows;
      for (int s = 0; s < h->p.sweeps_down; s++)
      {
         if (l < h->smooth_levels) ilu_iterate(h->ilu[l], A, h->smooth_sweeps, h->f[l], h->u[l], h->tmp[l], h->cor[l]);
         else level_relax(h, l, h->p.relax_down, h->l1d[l], h->f[l], h->u[l]);
      }
      /* t = f - A u ; f_{l+1} = P^T t */
      memcpy(h->tmp[l], h->f[l], sizeof(double) * (size_t)n);
      orc_spmv(A, -1.0, h->u[l], 1.0, h->tmp[l]);
      orc_spmv(h->R[l], 1.0, h->tmp[l], 0.0, h->f[l + 1]);
      memset(h->u[l + 1], 0, sizeof(double) * (size_t)h->A[l + 1]->nrows);
   }
   if (L == 1)
   {
      coarse_solve(h, 0, h->f[0], h->u[0]);
   }
   else
   {
      coarse_solve(h, L - 1, h->f[L - 1], h->u[L - 1]);
      for (int l = L - 2; l >= 0; l--)
      {
         const orc_csr *A = h->A[l];
         orc_spmv(h->P[l], 1.0, h->u[l + 1], 1.0, h->u[l]);
         for (int s = 0; s < h->p.sweeps_up; s++)
         {
            if (l < h->smooth_levels) ilu_iterate(h->ilu[l], A, h->smooth_sweeps, h->f[l], h->u[l], h->tmp[l], h->cor[l]);
            else level_relax(h, l, h->p.relax_up, h->l1u[l], h->f[l], h->u[l]);
         }
      }
   }
   memcpy(x, h->u[0], sizeof(double) * (size_t)h->A[0]->nrows);
}


/* ------------------------------------------------------------------ MGR (multigrid reduction)
 * hypre's MGR as hypredrive configures it (reference src/internal/mgr.c: defaults :1226-1330, name maps
 * :1553-1721; arg tree include/internal/mgr.h:132-178).  hypre is not in the reference tree; this restates
 * the published method (Ries/Trottenberg/Winter; hypre reference manual, "MGR") for the option subset below.
 * PARITY UNPINNED: the reference's MGR outputs (refOutput/ex3..ex7) need data sets that are not in the tree.
 *
 * Per reduction level: unknowns whose label is in f_labels are F points, the rest C points (ascending order
 * keeps their relative numbering).  P = [W; I], R = [Z I], A_c = R A P (coarse_level_type rap).
 *   prolongation 0 injection W = 0 | 2 jacobi W = -D_FF^-1 A_FC | 1 l1-jacobi W = -diag(l1(A_FF))^-1 A_FC
 *   restriction  0 injection Z = 0 | 2 jacobi Z = -A_CF D_FF^-1  | 14 columped Z = -A_CF diag(colsum(A_FF))^-1
 * Cycle (cycle 1, smoothing position "pre"): global relaxation sweeps on all points, F-relaxation sweeps
 * (Jacobi on the F rows of the whole operator: u_F += D_FF^-1 (f - A u)_F), restrict the residual, recurse,
 * u += P e_c.  Coarsest level: one BoomerAMG V-cycle from a zero guess. */
typedef struct {
   orc_csr *A, *P, *R;
   int      n, nc;
   int     *labels, *cf, *cidx;
   double  *dinvF, *l1g;
   double  *f, *u, *t, *cor;
   orc_ilu *gilu; /* g_relaxation ilu (type 16): ILU(0) of the level operator, hypre's default ILU */
   /* f_relaxation amg (type 2): one BoomerAMG cycle on A_FF per sweep */
   orc_csr *Aff;
   orc_amg *famg;
   orc_ilu *filu; /* f_relaxation ilu (type 32): ILU(0) of A_FF */
   int     *fidx, nf, frelax_type;
   double  *rF, *eF;
   int      frelax_sweeps, grelax_type, grelax_sweeps;
   int      gnb;      /* row blocks of the global relaxation (orc_mgr_level_params.grelax_blocks), 1 = one block */
   int64_t *gpart;    /* gnb + 1 row starts */
   int      fkry, fkry_pre; /* nested Krylov F-relaxation: method + 1, preconditioned by famg */
   orc_krylov_params fkp;
} mgr_level;

struct orc_mgr {
   int        nlev; /* reduction levels */
   mgr_level *lv;
   orc_csr   *Ac;   /* coarsest operator */
   orc_amg   *camg;
   orc_ilu   *cilu; /* coarsest_level ilu: coarse_ilu_iters iterations of x += M^-1 (f - A_c x) */
   int        cilu_iters;
   double    *ct, *cc;
   double    *fc, *uc;
   int        max_iter;
   int        cycle, fpos, gpos; /* 1 V / 2 W; smoothing positions 1 pre, 2 post, 3 both */
   int        ckry, ckry_pre; /* nested Krylov coarsest solve */
   orc_krylov_params ckp;
};

/* A nested Krylov component: the solver runs to its own max_iter / tolerance from the guess it is given; missing the tolerance
 * is no error (reference hypredrv_NestedKrylovSolve, src/internal/krylov.c:557-603). */
static void
nested_krylov(int method1, const orc_krylov_params *kp, const orc_csr *A, orc_amg *pre, const double *b, double *x)
{
   int     conv = 0;
   double  rel  = 0.0;
   double *hist = (double *)malloc(sizeof(double) * (size_t)(kp->max_iter + 2));
   switch (method1)
   {
      case 1: orc_pcg(A, pre, kp, b, x, hist, &conv, &rel); break;
      case 2: orc_gmres(A, pre, kp, b, x, hist, &conv, &rel); break;
      case 3: orc_fgmres(A, pre, kp, b, x, hist, &conv, &rel); break;
      default: orc_bicgstab(A, pre, kp, b, x, hist, &conv, &rel); break;
   }
   free(hist);
}

static int
label_in(int lab, const int *set, int n)
{
   for (int i = 0; i < n; i++)
      if (set[i] == lab) return 1;
   return 0;
}

orc_amg *
orc_precond_mgr(const orc_csr *A0, const int *labels0, int nlevels, const orc_mgr_level_params *lp,
                const orc_amg_params *coarse_amg, int max_iter)
{
   struct orc_mgr *M = (struct orc_mgr *)calloc(1, sizeof(struct orc_mgr));
   M->nlev     = nlevels;
   M->lv       = (mgr_level *)calloc((size_t)(nlevels > 0 ? nlevels : 1), sizeof(mgr_level));
   M->max_iter = max_iter < 1 ? 1 : max_iter;
   /* level 0 works on a copy so that every level owns its operator */
   orc_csr *A = orc_csr_alloc(A0->nrows, A0->ncols, A0->rowptr[A0->nrows]);
   memcpy(A->rowptr, A0->rowptr, sizeof(int) * (size_t)(A0->nrows + 1));
   memcpy(A->col, A0->col, sizeof(int) * (size_t)A0->rowptr[A0->nrows]);
   memcpy(A->val, A0->val, sizeof(double) * (size_t)A0->rowptr[A0->nrows]);
   int *labels = (int *)malloc(sizeof(int) * (size_t)(A->nrows > 0 ? A->nrows : 1));
   memcpy(labels, labels0, sizeof(int) * (size_t)A->nrows);
   for (int l = 0; l < nlevels; l++)
   {
      mgr_level *L = &M->lv[l];
      const int  n = A->nrows;
      L->A = A; L->n = n; L->labels = labels;
      L->frelax_type   = lp[l].frelax_type;
      L->fkry = lp[l].frelax_krylov; L->fkry_pre = lp[l].frelax_krylov_precond; L->fkp = lp[l].frelax_kp;
      L->frelax_sweeps = lp[l].frelax_sweeps; L->grelax_type = lp[l].grelax_type; L->grelax_sweeps = lp[l].grelax_sweeps;
      L->cf    = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
      L->cidx  = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
      L->dinvF = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
      int nc = 0;
      for (int i = 0; i < n; i++)
      {
         L->cf[i]   = label_in(labels[i], lp[l].f_labels, lp[l].n_f_labels) ? -1 : 1;
         L->cidx[i] = (L->cf[i] > 0) ? nc++ : -1;
      }
      L->nc = nc;
      /* F-point divisors: a_ii (frelax 7) or the l1 norm of the whole row (18) */
      double *dF = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));   /* a_ii on F rows */
      double *l1F = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));  /* sum_j in F |a_ij| on F rows */
      double *csum = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); /* column sums of A_FF */
      for (int i = 0; i < n; i++)
      {
         if (L->cf[i] > 0) continue;
         double l1 = 0.0;
         for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
         {
            const int j = A->col[k];
            l1 += fabs(A->val[k]);
            if (j == i) dF[i] = A->val[k];
            if (L->cf[j] < 0) { l1F[i] += fabs(A->val[k]); csum[j] += A->val[k]; } /* rows ascending: fixed order per column */
         }
         const double d = (lp[l].frelax_type == 18) ? l1 : dF[i];
         L->dinvF[i]    = (d != 0.0) ? 1.0 / d : 0.0;
      }
      /* P */
      {
         int nnz = 0;
         for (int i = 0; i < n; i++)
         {
            if (L->cf[i] > 0) { nnz++; continue; }
            if (lp[l].interp_type == 0) continue;
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) nnz += (L->cf[A->col[k]] > 0);
         }
         orc_csr *P = orc_csr_alloc(n, nc, nnz);
         int q = 0;
         for (int i = 0; i < n; i++)
         {
            P->rowptr[i] = q;
            if (L->cf[i] > 0) { P->col[q] = L->cidx[i]; P->val[q++] = 1.0; continue; }
            if (lp[l].interp_type == 0) continue;
            const double d = (lp[l].interp_type == 1) ? l1F[i] : dF[i];
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
               if (L->cf[A->col[k]] > 0) { P->col[q] = L->cidx[A->col[k]]; P->val[q++] = -A->val[k] / d; }
         }
         P->rowptr[n] = q;
         L->P = P;
      }
      /* R */
      {
         int nnz = 0;
         for (int i = 0; i < n; i++)
         {
            if (L->cf[i] < 0) continue;
            nnz++;
            if (lp[l].restrict_type == 0) continue;
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) nnz += (L->cf[A->col[k]] < 0);
         }
         orc_csr *R = orc_csr_alloc(nc, n, nnz);
         int q = 0, c = 0;
         for (int i = 0; i < n; i++)
         {
            if (L->cf[i] < 0) continue;
            R->rowptr[c++] = q;
            int placed = 0; /* keep columns ascending: the identity entry goes where i sorts */
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1] && lp[l].restrict_type != 0; k++)
            {
               const int j = A->col[k];
               if (L->cf[j] > 0) continue;
               if (!placed && j > i) { R->col[q] = i; R->val[q++] = 1.0; placed = 1; }
               const double d = (lp[l].restrict_type == 14) ? csum[j] : dF[j];
               R->col[q] = j; R->val[q++] = -A->val[k] / d;
            }
            if (!placed) { R->col[q] = i; R->val[q++] = 1.0; }
         }
         R->rowptr[nc] = q;
         L->R = R;
      }
      free(dF); free(l1F); free(csum);
      if (lp[l].frelax_type == 2 || lp[l].frelax_type == 32)
      { /* A_FF in the relative order of the F points, BoomerAMG (2) or ILU(0) (32) on it */
         L->fidx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
         int nf = 0, nnzf = 0;
         for (int i = 0; i < n; i++) L->fidx[i] = (L->cf[i] < 0) ? nf++ : -1;
         for (int i = 0; i < n; i++)
            if (L->cf[i] < 0)
               for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) nnzf += (L->cf[A->col[k]] < 0);
         L->nf  = nf;
         L->Aff = orc_csr_alloc(nf, nf, nnzf);
         int q = 0;
         for (int i = 0; i < n; i++)
         {
            if (L->cf[i] > 0) continue;
            L->Aff->rowptr[L->fidx[i]] = q;
            for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++)
               if (L->cf[A->col[k]] < 0) { L->Aff->col[q] = L->fidx[A->col[k]]; L->Aff->val[q++] = A->val[k]; }
         }
         L->Aff->rowptr[nf] = q;
         if (lp[l].frelax_type == 2)
         {
            orc_amg_params fp;
            if (lp[l].frelax_amg) fp = *lp[l].frelax_amg;
            else orc_amg_default_params(&fp, 1);
            L->famg = orc_amg_setup(L->Aff, &fp);
         }
         else L->filu = orc_ilu0_setup(L->Aff, 0, NULL, lp[l].ilu_tri_solve, lp[l].ilu_lower_it, lp[l].ilu_upper_it);
         L->rF   = (double *)calloc((size_t)(nf > 0 ? nf : 1), sizeof(double));
         L->eF   = (double *)calloc((size_t)(nf > 0 ? nf : 1), sizeof(double));
      }
      /* global relaxation divisors */
      if (L->grelax_type == 16)
      {
         L->gilu = orc_ilu0_setup(A, 0, NULL, lp[l].ilu_tri_solve, lp[l].ilu_lower_it, lp[l].ilu_upper_it);
         L->cor  = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
      }
      else if (L->grelax_type >= 0)
      {
         L->l1g = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
         L->gnb = lp[l].grelax_blocks > 1 ? lp[l].grelax_blocks : 1;
         if (L->gnb > 1)
         { /* hypre_GeneratePartitioning of this level's rows */
            L->gpart = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L->gnb + 1));
            for (int q = 0; q <= L->gnb; q++) L->gpart[q] = (int64_t)(((__int128)q * n) / L->gnb);
            orc_l1_norms_blocks(A, (L->grelax_type == 18) ? 1 : 4, L->gnb, L->gpart, L->l1g);
         }
         else orc_l1_norms(A, (L->grelax_type == 18) ? 1 : 4, L->l1g);
      }
      L->f = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
      L->u = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
      L->t = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
      /* coarse operator and its labels */
      orc_csr *AP = spgemm(A, L->P);
      orc_csr *Ac = spgemm(L->R, AP);
      orc_csr_free(AP);
      int *lc = (int *)malloc(sizeof(int) * (size_t)(nc > 0 ? nc : 1));
      for (int i = 0; i < n; i++)
         if (L->cf[i] > 0) lc[L->cidx[i]] = labels[i];
      A = Ac;
      labels = lc;
   }
   M->Ac   = A;
   free(labels);
   M->cycle = 1; M->fpos = 1; M->gpos = 1;
   if (nlevels > 0)
   {
      const orc_mgr_level_params *q = &lp[nlevels - 1];
      if (q->mgr_cycle > 0) M->cycle = q->mgr_cycle;
      if (q->mgr_frelax_pos > 0) M->fpos = q->mgr_frelax_pos;
      if (q->mgr_gsmooth_pos > 0) M->gpos = q->mgr_gsmooth_pos;
   }
   if (nlevels > 0) { M->ckry = lp[nlevels - 1].coarse_krylov; M->ckry_pre = lp[nlevels - 1].coarse_krylov_precond; M->ckp = lp[nlevels - 1].coarse_kp; }
   if (coarse_amg) M->camg = orc_amg_setup(A, coarse_amg);
   else
   { /* coarsest_level: ilu (the ILU arguments ride in the last level's slots) */
      const orc_mgr_level_params *q = &lp[nlevels - 1];
      M->cilu       = orc_ilu0_setup(A, 0, NULL, q->coarse_ilu_tri_solve, q->coarse_ilu_lower_it, q->coarse_ilu_upper_it);
      M->cilu_iters = q->coarse_ilu_max_iter < 1 ? 1 : q->coarse_ilu_max_iter;
      M->ct         = (double *)calloc((size_t)(A->nrows > 0 ? A->nrows : 1), sizeof(double));
      M->cc         = (double *)calloc((size_t)(A->nrows > 0 ? A->nrows : 1), sizeof(double));
   }
   M->fc   = (double *)calloc((size_t)(A->nrows > 0 ? A->nrows : 1), sizeof(double));
   M->uc   = (double *)calloc((size_t)(A->nrows > 0 ? A->nrows : 1), sizeof(double));
   orc_amg *h = (orc_amg *)calloc(1, sizeof(orc_amg));
   h->mgr     = M;
   return h;
}

static void
mgr_global_relax(mgr_level *L, const double *f, double *u)
{
   if (L->grelax_type == 16 && L->gilu) ilu_iterate(L->gilu, L->A, L->grelax_sweeps, f, u, L->t, L->cor);
   else if (L->grelax_type >= 0)
      for (int s = 0; s < L->grelax_sweeps; s++)
      {
         if (L->gnb > 1) orc_relax_blocks(L->A, L->l1g, L->grelax_type == 88 ? 8 : L->grelax_type, 1.0, f, u, L->t, L->gnb, L->gpart);
         else orc_relax(L->A, L->l1g, L->grelax_type == 88 ? 8 : L->grelax_type, 1.0, f, u, L->t);
      }
}

static void
mgr_f_relax(mgr_level *L, const double *f, double *u)
{
   const int n = L->n;
   for (int s = 0; s < L->frelax_sweeps; s++)
   {
      memcpy(L->t, f, sizeof(double) * (size_t)n);
      orc_spmv(L->A, -1.0, u, 1.0, L->t);
      if (L->frelax_type == 2 || L->frelax_type == 32)
      { /* e_F = M_FF^-1 r_F (one BoomerAMG cycle from a zero guess, or one ILU(0) solve), u_F += e_F */
         for (int i = 0; i < n; i++)
            if (L->cf[i] < 0) L->rF[L->fidx[i]] = L->t[i];
         memset(L->eF, 0, sizeof(double) * (size_t)L->nf);
         if (L->fkry && L->frelax_type == 2) nested_krylov(L->fkry, &L->fkp, L->Aff, L->fkry_pre ? L->famg : NULL, L->rF, L->eF);
         else if (L->frelax_type == 2) orc_amg_vcycle(L->famg, L->rF, L->eF);
         else orc_ilu_apply(L->filu, L->rF, L->eF);
         for (int i = 0; i < n; i++)
            if (L->cf[i] < 0) u[i] += L->eF[L->fidx[i]];
      }
      else
         for (int i = 0; i < n; i++) u[i] += L->dinvF[i] * L->t[i];
   }
}

static void
mgr_cycle(struct orc_mgr *M, int l, const double *f, double *u)
{
   if (l == M->nlev)
   {
      memset(u, 0, sizeof(double) * (size_t)M->Ac->nrows);
      if (M->ckry && M->camg) nested_krylov(M->ckry, &M->ckp, M->Ac, M->ckry_pre ? M->camg : NULL, f, u);
      else if (M->camg) orc_amg_vcycle(M->camg, f, u);
      else ilu_iterate(M->cilu, M->Ac, M->cilu_iters, f, u, M->ct, M->cc);
      return;
   }
   mgr_level *L = &M->lv[l];
   /* smoothing positions (reference mgr.c:614-675): before the coarse correction global relaxation then F-relaxation, after it
    * the mirror image; W-cycle: the coarser level is visited twice */
   if (M->gpos & 1) mgr_global_relax(L, f, u);
   if (M->fpos & 1) mgr_f_relax(L, f, u);
   const int n   = L->n;
   double   *fc = (l + 1 < M->nlev) ? M->lv[l + 1].f : M->fc;
   double   *uc = (l + 1 < M->nlev) ? M->lv[l + 1].u : M->uc;
   for (int visit = 0; visit < (M->cycle == 2 ? 2 : 1); visit++)
   {
      memcpy(L->t, f, sizeof(double) * (size_t)n);
      orc_spmv(L->A, -1.0, u, 1.0, L->t);
      orc_spmv(L->R, 1.0, L->t, 0.0, fc);
      memset(uc, 0, sizeof(double) * (size_t)L->nc);
      mgr_cycle(M, l + 1, fc, uc);
      orc_spmv(L->P, 1.0, uc, 1.0, u);
   }
   if (M->fpos & 2) mgr_f_relax(L, f, u);
   if (M->gpos & 2) mgr_global_relax(L, f, u);
}

static void
mgr_solve(struct orc_mgr *M, const double *b, double *x)
{
   for (int it = 0; it < M->max_iter; it++) mgr_cycle(M, 0, b, x);
}

static void
mgr_free(struct orc_mgr *M)
{
   for (int l = 0; l < M->nlev; l++)
   {
      mgr_level *L = &M->lv[l];
      orc_csr_free(L->A); orc_csr_free(L->P); orc_csr_free(L->R);
      free(L->labels); free(L->cf); free(L->cidx); free(L->dinvF); free(L->l1g); free(L->f); free(L->u); free(L->t); free(L->cor);
      if (L->gilu) orc_ilu_free(L->gilu);
      if (L->famg) orc_amg_free(L->famg);
      if (L->filu) orc_ilu_free(L->filu);
      if (L->Aff) orc_csr_free(L->Aff);
      free(L->fidx); free(L->rF); free(L->eF); free(L->gpart);
   }
   orc_csr_free(M->Ac);
   orc_amg_free(M->camg);
   if (M->cilu) orc_ilu_free(M->cilu);
   free(M->ct); free(M->cc);
   free(M->fc); free(M->uc); free(M->lv);
   free(M);
}

static int
mgr_rebind(struct orc_mgr *M, const orc_csr *A)
{
   if (M->nlev < 1 || A->nrows != M->lv[0].A->nrows || A->ncols != M->lv[0].A->ncols) return 1;
   int      nnz = A->rowptr[A->nrows];
   orc_csr *C   = orc_csr_alloc(A->nrows, A->ncols, nnz);
   memcpy(C->rowptr, A->rowptr, sizeof(int) * (size_t)(A->nrows + 1));
   memcpy(C->col, A->col, sizeof(int) * (size_t)nnz);
   memcpy(C->val, A->val, sizeof(double) * (size_t)nnz);
   orc_csr_free(M->lv[0].A);
   M->lv[0].A = C;
   return 0;
}

const orc_csr *
orc_mgr_matrix(const orc_amg *h, int level, int which) /* which: 0 A_level (level == nlev: coarsest), 1 P, 2 R */
{
   const struct orc_mgr *M = h->mgr;
   if (!M || level < 0 || level > M->nlev) return NULL;
   if (level == M->nlev) return which == 0 ? M->Ac : NULL;
   return which == 0 ? M->lv[level].A : which == 1 ? M->lv[level].P : M->lv[level].R;
}

/* ------------------------------------------------------------------ Krylov */

static void
apply_precond(const orc_csr *A, orc_amg *h, const double *r, double *z)
{
   int n = A->nrows;
   if (!h)
   {
      memcpy(z, r, sizeof(double) * (size_t)n);
      return;
   }
   memset(z, 0, sizeof(double) * (size_t)n); /* hypre clears the vector first */
   orc_amg_vcycle(h, r, z);
}

/* hypre_PCGSolve as driven by src/internal/solver.c:561-620 with the settings of
 * src/internal/pcg.c:55-72 (two_norm 1, rel_change 0, stop_crit 0); SURVEY App. A.1. */
int
orc_pcg(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b, double *x,
        double *resid_hist, int *converged, double *final_rel)
{
   int     n = A->nrows, it = 0;
   double *r = (double *)malloc(sizeof(double) * (size_t)n);
   double *p = (double *)malloc(sizeof(double) * (size_t)n);
   double *s = (double *)malloc(sizeof(double) * (size_t)n);
   *converged = 0;
   double bi_prod;
   if (kp->two_norm)
      bi_prod = orc_dot(n, b, b);
   else
   {
      apply_precond(A, h, b, p);
      bi_prod = orc_dot(n, p, b);
   }
   if (bi_prod == 0.0)
   {
      memcpy(x, b, sizeof(double) * (size_t)n);
      *final_rel = 0.0;
      if (resid_hist) resid_hist[0] = 0.0;
      free(r); free(p); free(s);
      return 0;
   }
   double eps = kp->rtol * kp->rtol;
   {
      double a2 = kp->atol * kp->atol / bi_prod;
      if (a2 > eps) eps = a2;
   }
   memcpy(r, b, sizeof(double) * (size_t)n);
   orc_spmv(A, -1.0, x, 1.0, r);
   apply_precond(A, h, r, p);
   double gamma  = orc_dot(n, r, p);
   double i_prod = kp->two_norm ? orc_dot(n, r, r) : gamma;
   if (resid_hist) resid_hist[0] = sqrt(i_prod);
   while (it + 1 <= kp->max_iter)
   {
      it++;
      orc_spmv(A, 1.0, p, 0.0, s);
      double sdotp = orc_dot(n, s, p);
      if (sdotp == 0.0) { it--; break; }
      double alpha = gamma / sdotp;
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; i++)
      {
         x[i] += alpha * p[i];
         r[i] -= alpha * s[i];
      }
      apply_precond(A, h, r, s);
      double gamma_new = orc_dot(n, r, s);
      i_prod           = kp->two_norm ? orc_dot(n, r, r) : gamma_new;
      if (resid_hist) resid_hist[it] = sqrt(i_prod);
      if (i_prod / bi_prod < eps)
      {
         *converged = 1;
         break;
      }
      if (gamma_new < 1.0e-292 && -gamma_new < 1.0e-292) break;
      double beta = gamma_new / gamma;
      gamma       = gamma_new;
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; i++) p[i] = s[i] + beta * p[i];
   }
   *final_rel = sqrt(i_prod / bi_prod);
   free(r); free(p); free(s);
   return it;
}

/* hypre_GMRESSolve (right preconditioning, restart k, modified Gram-Schmidt, Givens),
 * reached through solver_ops[SOLVER_GMRES] src/internal/solver.c:217-228 with args
 * src/internal/gmres.c:16-27; SURVEY App. A.8. */
static int
gmres_core(int flexible, const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b, double *x,
           double *resid_hist, int *converged, double *final_rel)
{
   int      n = A->nrows, k = kp->krylov_dim, iter = 0;
   double **V  = (double **)malloc(sizeof(double *) * (size_t)(k + 1));
   for (int i = 0; i <= k; i++) V[i] = (double *)malloc(sizeof(double) * (size_t)n);
   /* FlexGMRES keeps the preconditioned directions z_j = M^-1 v_j and updates x with them */
   double **Z = NULL;
   if (flexible)
   {
      Z = (double **)malloc(sizeof(double *) * (size_t)k);
      for (int i = 0; i < k; i++) Z[i] = (double *)malloc(sizeof(double) * (size_t)n);
   }
   double *w  = (double *)malloc(sizeof(double) * (size_t)n);
   double *r  = (double *)malloc(sizeof(double) * (size_t)n);
   double *H  = (double *)calloc((size_t)(k + 1) * (size_t)k, sizeof(double)); /* H[i*k + j] */
   double *cs = (double *)calloc((size_t)k, sizeof(double));
   double *sn = (double *)calloc((size_t)k, sizeof(double));
   double *rs = (double *)calloc((size_t)k + 1, sizeof(double));
   *converged    = 0;
   double b_norm = sqrt(orc_dot(n, b, b));
   memcpy(V[0], b, sizeof(double) * (size_t)n);
   orc_spmv(A, -1.0, x, 1.0, V[0]);
   double r_norm   = sqrt(orc_dot(n, V[0], V[0]));
   double den_norm = (b_norm > 0.0) ? b_norm : r_norm;
   double epsilon  = kp->rtol * den_norm;
   if (kp->atol > epsilon) epsilon = kp->atol;
   if (resid_hist) resid_hist[0] = r_norm;
   if (r_norm == 0.0)
   {
      *converged = 1;
      *final_rel = 0.0;
      goto done;
   }
   while (iter < kp->max_iter)
   {
      rs[0] = r_norm;
      if (r_norm <= epsilon)
      { /* (also before the first iteration: hypre_GMRESSolve accepts an initial guess that already meets the tolerance with 0
         * iterations -- what the reference's tests/test_init_guess.c:170-199,247-270 assert) */
         /* true residual check */
         memcpy(r, b, sizeof(double) * (size_t)n);
         orc_spmv(A, -1.0, x, 1.0, r);
         r_norm = sqrt(orc_dot(n, r, r));
         if (r_norm <= epsilon) { *converged = 1; break; }
         memcpy(V[0], r, sizeof(double) * (size_t)n);
         rs[0] = r_norm;
      }
      double t = 1.0 / r_norm;
      for (int i = 0; i < n; i++) V[0][i] *= t;
      int i = 0;
      while (i < k && iter < kp->max_iter)
      {
         i++;
         iter++;
         double *zi = flexible ? Z[i - 1] : r;
         apply_precond(A, h, V[i - 1], zi);
         orc_spmv(A, 1.0, zi, 0.0, V[i]);
         for (int j = 0; j < i; j++)
         {
            double hji         = orc_dot(n, V[j], V[i]);
            H[j * k + (i - 1)] = hji;
            for (int q = 0; q < n; q++) V[i][q] -= hji * V[j][q];
         }
         double tn          = sqrt(orc_dot(n, V[i], V[i]));
         H[i * k + (i - 1)] = tn;
         if (tn != 0.0)
         {
            double ti = 1.0 / tn;
            for (int q = 0; q < n; q++) V[i][q] *= ti;
         }
         for (int j = 1; j < i; j++)
         {
            double hv                = H[(j - 1) * k + (i - 1)];
            H[(j - 1) * k + (i - 1)] = cs[j - 1] * hv + sn[j - 1] * H[j * k + (i - 1)];
            H[j * k + (i - 1)]       = -sn[j - 1] * hv + cs[j - 1] * H[j * k + (i - 1)];
         }
         double hh = H[(i - 1) * k + (i - 1)], hn = H[i * k + (i - 1)];
         double gm = sqrt(hh * hh + hn * hn);
         if (gm == 0.0) gm = 1.0e-16;
         cs[i - 1] = hh / gm;
         sn[i - 1] = hn / gm;
         rs[i]     = -sn[i - 1] * rs[i - 1];
         rs[i - 1] = cs[i - 1] * rs[i - 1];
         H[(i - 1) * k + (i - 1)] = cs[i - 1] * hh + sn[i - 1] * hn;
         r_norm                   = fabs(rs[i]);
         if (resid_hist) resid_hist[iter] = r_norm;
         if (r_norm <= epsilon) break;
      }
      /* solve the upper triangular system, update x through the preconditioner */
      rs[i - 1] = rs[i - 1] / H[(i - 1) * k + (i - 1)];
      for (int q = i - 2; q >= 0; q--)
      {
         double tt = rs[q];
         for (int j = q + 1; j < i; j++) tt -= H[q * k + j] * rs[j];
         rs[q] = tt / H[q * k + q];
      }
      if (flexible)
      {
         for (int j = i - 1; j >= 0; j--)
            for (int q = 0; q < n; q++) x[q] += rs[j] * Z[j][q];
      }
      else
      {
         for (int q = 0; q < n; q++) w[q] = rs[i - 1] * V[i - 1][q];
         for (int j = i - 2; j >= 0; j--)
            for (int q = 0; q < n; q++) w[q] += rs[j] * V[j][q];
         apply_precond(A, h, w, r);
         for (int q = 0; q < n; q++) x[q] += r[q];
      }
      /* restart residual */
      memcpy(V[0], b, sizeof(double) * (size_t)n);
      orc_spmv(A, -1.0, x, 1.0, V[0]);
      double true_norm = sqrt(orc_dot(n, V[0], V[0]));
      if (r_norm <= epsilon)
      {
         r_norm = true_norm;
         if (true_norm <= epsilon) { *converged = 1; break; }
      }
      else
         r_norm = true_norm;
   }
   *final_rel = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
done:
   for (int i = 0; i <= k; i++) free(V[i]);
   if (Z)
   {
      for (int i = 0; i < k; i++) free(Z[i]);
      free(Z);
   }
   free(V); free(w); free(r); free(H); free(cs); free(sn); free(rs);
   return iter;
}

int
orc_gmres(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b, double *x,
          double *resid_hist, int *converged, double *final_rel)
{
   return gmres_core(0, A, h, kp, b, x, resid_hist, converged, final_rel);
}

/* hypre_FlexGMRESSolve (solver_ops[SOLVER_FGMRES], src/internal/solver.c:229-240; args fgmres.c:15-22) */
int
orc_fgmres(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b, double *x,
           double *resid_hist, int *converged, double *final_rel)
{
   return gmres_core(1, A, h, kp, b, x, resid_hist, converged, final_rel);
}

/* hypre_BiCGSTABSolve (solver_ops[SOLVER_BICGSTAB], src/internal/solver.c:241-252; args bicgstab.c:15-23):
 * right-preconditioned BiCGSTAB (van der Vorst 1992), r0* = r0, stop on ||r|| <= max(atol, rtol ||b||) with
 * the true residual recomputed before accepting.  PARITY UNPINNED: no reference output uses it. */
int
orc_bicgstab(const orc_csr *A, orc_amg *h, const orc_krylov_params *kp, const double *b, double *x,
             double *resid_hist, int *converged, double *final_rel)
{
   int     n  = A->nrows, iter = 0;
   double *r0 = (double *)malloc(sizeof(double) * (size_t)n), *r = (double *)malloc(sizeof(double) * (size_t)n);
   double *p  = (double *)malloc(sizeof(double) * (size_t)n), *v = (double *)malloc(sizeof(double) * (size_t)n);
   double *q  = (double *)malloc(sizeof(double) * (size_t)n), *s = (double *)malloc(sizeof(double) * (size_t)n);
   *converged = 0;
   memcpy(r0, b, sizeof(double) * (size_t)n);
   orc_spmv(A, -1.0, x, 1.0, r0);
   memcpy(r, r0, sizeof(double) * (size_t)n);
   memcpy(p, r0, sizeof(double) * (size_t)n);
   double b_norm = sqrt(orc_dot(n, b, b));
   double res    = orc_dot(n, r0, r0);
   double r_norm = sqrt(res);
   double den    = (b_norm > 0.0) ? b_norm : r_norm;
   double eps    = kp->rtol * den;
   if (kp->atol > eps) eps = kp->atol;
   if (resid_hist) resid_hist[0] = r_norm;
   if (r_norm == 0.0)
   {
      *converged = 1;
      *final_rel = 0.0;
      goto done;
   }
   while (iter < kp->max_iter)
   {
      iter++;
      apply_precond(A, h, p, v);
      orc_spmv(A, 1.0, v, 0.0, q);
      double temp = orc_dot(n, r0, q);
      if (temp == 0.0) break; /* breakdown */
      double alpha = res / temp;
      for (int i = 0; i < n; i++) x[i] += alpha * v[i];
      for (int i = 0; i < n; i++) r[i] -= alpha * q[i];
      apply_precond(A, h, r, v);
      orc_spmv(A, 1.0, v, 0.0, s);
      double gn = orc_dot(n, r, s), gd = orc_dot(n, s, s);
      double gamma = (gn == 0.0 && gd == 0.0) ? 0.0 : gn / gd;
      for (int i = 0; i < n; i++) x[i] += gamma * v[i];
      for (int i = 0; i < n; i++) r[i] -= gamma * s[i];
      r_norm = sqrt(orc_dot(n, r, r));
      if (resid_hist) resid_hist[iter] = r_norm;
      if (r_norm <= eps)
      { /* accept only on the true residual */
         memcpy(r, b, sizeof(double) * (size_t)n);
         orc_spmv(A, -1.0, x, 1.0, r);
         r_norm = sqrt(orc_dot(n, r, r));
         if (r_norm <= eps) { *converged = 1; break; }
      }
      if (res == 0.0 || gamma == 0.0) break; /* breakdown */
      double beta = 1.0 / res;
      res         = orc_dot(n, r0, r);
      beta *= res;
      double c = beta * alpha / gamma;
      for (int i = 0; i < n; i++) p[i] = r[i] + c * (p[i] - gamma * q[i]);
   }
   *final_rel = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
done:
   free(r0); free(r); free(p); free(v); free(q); free(s);
   return iter;
}
