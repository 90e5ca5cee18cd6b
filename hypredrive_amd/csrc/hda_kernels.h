// hda_kernels.h -- launch wrappers of the solve-phase HIP kernels (gfx950).
// Each wrapper names the hypre entry point it replaces on the reference's hot path
// (SURVEY.md 2.4, K1/K2/K7/K8/K9).
#pragma once

#include "hda_common.h"

namespace hda {

struct HaloPlan; // hda_dist.h

// Every product below takes an optional halo plan of its input vector (row-partitioned runs): the ghost tail of x
// is then refreshed by the call itself and the transfer runs UNDER the product -- the rows' owned-column part is
// computed while the ghost values travel on the communication stream, the ghost-column part is added afterwards
// (SURVEY 2.4 C1).  With a plan, x must be writable up to [owned | ghosts].  halo == nullptr: x is used as it is.

// ---- K1: CSR SpMV family (replaces HYPRE_ParCSRMatrixMatvec, reached from
// src/internal/linsys.c:3031 and every PCG iteration via src/internal/solver.c:614).
// y_out = alpha*A*x + beta*y_in   (y_in may alias y_out; beta==0 never reads y_in)
void spmv(const DCsr &A, double alpha, const double *x, double beta, const double *y_in,
          double *y_out, const HaloPlan *halo = nullptr);
// y = A*x and block partials of <y, w> into slot (fused dot, K9)
// y = A x and, when the operator's kernel can do it in the same pass (returns true), y2 = dinv2 .* y: the zero-guess Jacobi sweep
// of the next coarser level fused into the restriction.  false: only y was written.
bool spmv_with_scaled_copy(const DCsr &A, const double *x, double *y, const double *dinv2, double *y2, const HaloPlan *halo = nullptr);
void spmv_dot(const DCsr &A, const double *x, double *y, const double *w, int slot, const HaloPlan *halo = nullptr);
// out = b - A*x
void residual(const DCsr &A, const double *x, const double *b, double *out, const HaloPlan *halo = nullptr);

// ---- K2: l1-Jacobi / weighted Jacobi sweep (hypre_BoomerAMGRelax types 18 / 0,7;
// selected by src/internal/amg.c:183-186,360-375).  dinv = weight / l1 (or / a_ii).
// x_out = x_in + dinv .* (b - A*x_in).  dot_slot >= 0 also emits partials of <b, x_out>.
void jacobi(const DCsr &A, const double *dinv, const double *b, const double *x_in,
            double *x_out, int dot_slot, const HaloPlan *halo = nullptr);
// first sweep from a zero guess: x = dinv .* b
void jacobi_zero_guess(int n, const double *dinv, const double *b, double *x);

// ---- K9: fused BLAS-1 of the PCG recurrences (hypre_PCGSolve inner loop, reached from
// solver_ops[SOLVER_PCG].solve src/internal/solver.c:211).
enum Scalar : int {
   // PCG keeps <r,z> and <r,r> of an iteration side by side (pairs 0/1 and 2/3, alternating by iteration parity) so
   // that one finalize_n + ONE two-double all-reduce serves both (C2 of SURVEY 2.4)
   S_GAMMA0 = 0, S_RR0 = 1, S_GAMMA1 = 2, S_RR1 = 3, S_SP = 4, S_BB = 5, S_TMP = 6, S_TMP2 = 7,
   S_GMRES = 8 /* .. S_GMRES + krylov_dim + 1 */
};
void dot(int n, const double *x, const double *y, int slot);              // partials only
void finalize(int slot, int scalar_idx);                                  // scalars[idx] = sum(slot)
void finalize_n(int first_slot, int nslots, int first_scalar);            // several at once
// alpha = gamma/S_SP; x += alpha p; r -= alpha s; partials <r,r>.  sp_slot >= 0 (one rank): <s,p> is finished from that slot of block
// partials inside the kernel (no finalize launch; same bits).  z0 != null: also z0 = dinv0 .* r (the zero-guess first sweep of the cycle)
void cg_update(int n, int gamma_idx, const double *p, const double *s, double *x, double *r, int rr_slot, int sp_slot = -1,
               const double *dinv0 = nullptr, double *z0 = nullptr);
// p = z + (gamma_new / gamma_old) p.  first_slot >= 0 (one rank): gamma_new and the scalar after it are finished here from the block
// partials of slots first_slot, first_slot + 1 (no finalize launch; same bits)
void cg_direction(int n, int gamma_old_idx, int gamma_new_idx, const double *z, double *p, int first_slot = -1);
// one step of the single-reduction (Chronopoulos-Gear) PCG: p = u + beta p; s = w + beta s; x += alpha p; r -= alpha s; partials <r,r>
// with beta, alpha formed on the device from scalars[t_new] = <r,u>, [t_new + 2] = <w,u>, [t_old] = the previous <r,u>, [alpha_idx]
void cg_single_step(int n, int t_new, int t_old, int alpha_idx, bool first, const double *u, const double *w, double *p, double *s, double *x,
                    double *r, int rr_slot);
void axpy(int n, double a, const double *x, double *y);                   // y += a x
void axpy_dev(int n, int scalar_idx, double sign, const double *x, double *y); // y += sign*scalars[idx]*x
void scale(int n, double a, double *x);
void scale_inv_sqrt_dev(int n, int scalar_idx, double *x);                // x /= sqrt(scalars[idx])
void copy(int n, const double *x, double *y);
void fill(int n, double v, double *x);
void mul(int n, const double *a, const double *b, double *out);           // out = a .* b
double read_scalar(int idx);                                              // sync read-back
void   read_scalars_async(int first, int count);                          // -> ctx.host_scalars, records ctx.ev

// ---- K8: coarsest-level dense solve (hypre relax type 9, src/internal/amg.c:190).
// inv is the n x n row-major inverse built at setup; x = inv * b with one workgroup.
void dense_apply(int n, const double *inv, const double *b, double *x);
// In-place Gauss-Jordan inverse of an n x n row-major matrix (no pivoting, like gselim).
void dense_invert(int n, double *a, double *inv);
void csr_to_dense(const DCsr &A, double *dense);

// ---- utilities
void exclusive_scan(int n, const int *in, int *out, int *total_out_dev); // out[i] = sum_{j<i} in[j]; out[n] = total
void require_int32_total(long n, const int *counts, const char *what);  // throws when sum(counts) >= 2^31
// hypre_ParCSRComputeL1Norms option 1 / 4 (0: the plain diagonal); part (device, nb + 1 row starts): row blocks whose outside counts as off-rank
void l1_row_norms(const DCsr &A, int option, double *l1, const int *part = nullptr, int nb = 0);
void extract_diag(const DCsr &A, double *d);
void make_dinv(int n, const double *d, double weight, double *dinv);
// bytes of (col, val) one product streams: plain = 12 nnz; stencil-coded = 1 nnz (+ 12 per escape);
// format = true asks for what the kernels really read, false for the CSR figure of SURVEY 8(d)
double matrix_stream_bytes(const DCsr &A, bool format);
// bytes of row pointers one product reads: 4 (n + 1), or -- format = true on a row-class coded operator -- only those of its CSR rows
double rowptr_stream_bytes(const DCsr &A, bool format);
// build the launch plans of A now (chunk plan, stencil coding attempt) instead of at its first product
void spmv_prepare(const DCsr &A);
// timing probe: bracket every launch of mode `mode` (0 plain, 1 residual, 2 Jacobi) on matrix A
// with HIP events on the library stream; read returns the average launch duration
void spmv_probe_set(const DCsr *A, int mode);   // clear, then arm one probe (A == nullptr: just clear)
void spmv_probe_read(double *avg_ms, int *count); // probe 0
// several probes at once (bench.py: dominant sweep, level-0 product, level-0 transfers)
void spmv_probe_clear();
int  spmv_probe_add(const DCsr *A, int mode);
void spmv_probe_read(int id, double *avg_ms, int *count);
void sort_rows(DCsr &A);                                    // column-sort every row in place
void sort_rows_segmented(DCsr &A);                          // the same by one segmented radix sort (any row length)
void transpose(const DCsr &A, DCsr &T);                     // rows of T sorted
void transpose_pattern_unsorted(const DCsr &A, DArray<int> &trp, DArray<int> &tcj); // pattern of A^T, rows unordered
// 7-pt Laplacian generator on device (examples/src/C_laplacian/laplacian.c:719-921),
// rows [ilower, iupper] of the block-partitioned numbering; cols are GLOBAL ids (int64).
void lap7_generate(const int n[3], const int P[3], const int pc[3], const double c[3],
                   int rowptr_out[], long long cols_out[], double vals_out[], double rhs_out[],
                   int local_n);

// products of a row partition overlap their ghost refresh (1), run it first (0), or decide by the transport / HDA_OVERLAP (-1: default)
void set_overlap_mode(int mode);

} // namespace hda
