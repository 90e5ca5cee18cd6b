// hda_amg_agg.hip -- aggressive coarsening of the first levels of a BoomerAMG-style hierarchy (single rank block).
//
// Reference surface: AMGagg_args (src/internal/amg.c:160-173: num_levels 0, num_paths 1, prolongation_type 4 = multipass,
// max_nnz_row 0, trunc factors 0) forwarded by hypredrv_AMGCreate through HYPRE_BoomerAMGSetAggNumLevels / SetNumPaths /
// SetAggInterpType / SetAgg*Trunc* (amg.c:938-944).  hypre itself is not in the reference tree and no checked-in output uses these
// options: the algorithm is the published one (Stueben 1999: A1 / A2 aggressive coarsening and multipass interpolation; Yang 2010,
// "On long-range interpolation operators for aggressive coarsening"), restated in the CPU checker (orc_second_strength,
// orc_coarsen_second_pass, orc_interp_multipass) -- PARITY UNPINNED -- and the kernels here reproduce the oracle bit for bit:
// every floating-point sum runs in the oracle's order (rows sequentially in column order; the pass products on the deterministic
// SpGEMM of the Galerkin product).
//
//   second strength graph   C points i, j of the first coarsening are connected when at least num_paths paths of length <= 2 lead
//                           from i to j along strong connections: S2 = [ S_CC + S_C: * S_:C >= num_paths ], no diagonal
//   second coarsening       PMIS over S2; first-pass C points that become F there are F points of the level
//   multipass interpolation pass 1: direct interpolation from strong C neighbours; pass p: through the rows of the strong
//                           neighbours of pass p - 1; alfa_i = -(sum of all off-diagonals) / (a_ii * sum over the neighbours used)
#include "hda_amg.h"

namespace hda {

#define STREAM (Context::get().stream)

namespace {

constexpr int kC = 1, kF = -1; // C/F marker values (as amg_pmis writes them; -3 = special F, never interpolated)

__global__ __launch_bounds__(256) void k_agg_cmark(int n, const int *__restrict__ cf, int *__restrict__ m)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) m[i] = (cf[i] == kC);
}
__global__ __launch_bounds__(256) void k_agg_fill_u8(long n, unsigned char v, unsigned char *__restrict__ out)
{
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = v;
}

// ---- second strength graph: the three factors.  cr / cc / cd: entries of row i in S_C: (rows = C points), S_:C (columns = C points,
// renumbered), S_CC; written at c1[i] for the two matrices with C rows
__global__ __launch_bounds__(256) void k_ss_count(int n, const int *__restrict__ rp, const int *__restrict__ cj, const unsigned char *__restrict__ sm,
                                                  const int *__restrict__ cf, const int *__restrict__ c1, int *__restrict__ cr, int *__restrict__ cc,
                                                  int *__restrict__ cd)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int s = 0, c = 0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (sm[k]) { s++; c += (cf[cj[k]] == kC); }
   cc[i] = c;
   if (cf[i] == kC) { cr[c1[i]] = s; cd[c1[i]] = c; }
}
__global__ __launch_bounds__(256) void k_ss_fill(int n, const int *__restrict__ rp, const int *__restrict__ cj, const unsigned char *__restrict__ sm,
                                                 const int *__restrict__ cf, const int *__restrict__ c1, const int *__restrict__ rpr,
                                                 int *__restrict__ cjr, double *__restrict__ vr, const int *__restrict__ rpc, int *__restrict__ cjc,
                                                 double *__restrict__ vc, const int *__restrict__ rpd, int *__restrict__ cjd, double *__restrict__ vd)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const bool crow = cf[i] == kC;
   int        a = crow ? rpr[c1[i]] : 0, b = rpc[i], d = crow ? rpd[c1[i]] : 0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      if (!sm[k]) continue;
      const int  j    = cj[k];
      const bool ccol = cf[j] == kC;
      if (crow) { cjr[a] = j; vr[a++] = 1.0; }
      if (ccol) { cjc[b] = c1[j]; vc[b++] = 1.0; }
      if (crow && ccol) { cjd[d] = c1[j]; vd[d++] = 1.0; }
   }
}
// S2 row i = entries of (T + D) row i with at least num_paths paths, diagonal dropped (both operands column-sorted)
template <bool FILL>
__global__ __launch_bounds__(256) void k_ss_merge(int n1, const int *__restrict__ trp, const int *__restrict__ tcj, const double *__restrict__ tv,
                                                  const int *__restrict__ drp, const int *__restrict__ dcj, const double *__restrict__ dv, double num_paths,
                                                  int *__restrict__ cnt, const int *__restrict__ orp, int *__restrict__ ocj, double *__restrict__ ov)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n1) return;
   int t = trp[i], te = trp[i + 1], e = drp[i], ee = drp[i + 1], q = FILL ? orp[i] : 0;
   while (t < te || e < ee)
   {
      const int jt = (t < te) ? tcj[t] : 0x7fffffff, jd = (e < ee) ? dcj[e] : 0x7fffffff, j = min(jt, jd);
      double    c  = 0.0;
      if (jt == j) c += tv[t++];
      if (jd == j) c += dv[e++];
      if (j != i && c >= num_paths)
      {
         if (FILL) { ocj[q] = j; ov[q] = c; }
         q++;
      }
   }
   if (!FILL) cnt[i] = q;
}
__global__ __launch_bounds__(256) void k_agg_merge_cf(int n, const int *__restrict__ c1, const int *__restrict__ cf2, int *__restrict__ cf)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n && cf[i] == kC && cf2[c1[i]] == kF) cf[i] = kF;
}

// ---- multipass interpolation
__global__ __launch_bounds__(256) void k_mp_init(int n, const int *__restrict__ cf, int *__restrict__ pass)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) pass[i] = (cf[i] == kC) ? 0 : -1;
}
// synchronous round p: an unassigned F point with a strong neighbour of pass p - 1 is marked (-2); k_mp_commit turns marks into p
__global__ __launch_bounds__(256) void k_mp_mark(int n, int p, const int *__restrict__ rp, const int *__restrict__ cj, const unsigned char *__restrict__ sm,
                                                 const int *__restrict__ cf, int *pass, int *counter)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n || pass[i] != -1 || cf[i] != kF) return;
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (sm[k])
      {
         const int pj = pass[cj[k]]; // (a neighbour marked -2 in this round reads as "unassigned": -2 != p - 1 for p >= 1)
         if (pj == p - 1) { pass[i] = -2; atomicAdd(counter, 1); return; }
      }
}
__global__ __launch_bounds__(256) void k_mp_commit(int n, int p, int *__restrict__ pass)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n && pass[i] == -2) pass[i] = p;
}
// alfa_i and the entry count of row i in W (after pass 1) -- sums run sequentially in column order, as in the oracle
__global__ __launch_bounds__(256) void k_mp_alfa(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                 const unsigned char *__restrict__ sm, const int *__restrict__ pass, double *__restrict__ alfa,
                                                 int *__restrict__ cnt1)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int pi = pass[i];
   double    a  = 0.0;
   int       c  = (pi == 0) ? 1 : 0;
   if (pi >= 1)
   {
      double diag = 0.0, sum_n = 0.0, sum_c = 0.0;
      for (int k = rp[i]; k < rp[i + 1]; k++)
      {
         const int j = cj[k];
         if (j == i) { diag = v[k]; continue; }
         sum_n += v[k];
         if (sm[k] && pass[j] == pi - 1) { sum_c += v[k]; c += (pi == 1); }
      }
      a = (sum_c * diag != 0.0) ? -sum_n / (sum_c * diag) : 0.0;
   }
   alfa[i] = a;
   cnt1[i] = c;
}
__global__ __launch_bounds__(256) void k_mp_fill1(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                  const unsigned char *__restrict__ sm, const int *__restrict__ pass, const int *__restrict__ cidx,
                                                  const double *__restrict__ alfa, const int *__restrict__ wrp, int *__restrict__ wcj,
                                                  double *__restrict__ wv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int q = wrp[i];
   if (pass[i] == 0) { wcj[q] = cidx[i]; wv[q] = 1.0; }
   else if (pass[i] == 1)
   {
      const double a = alfa[i];
      for (int k = rp[i]; k < rp[i + 1]; k++)
         if (sm[k] && pass[cj[k]] == 0) { wcj[q] = cidx[cj[k]]; wv[q++] = a * v[k]; }
   }
}
// M_p: the scaled strong pass-(p-1) entries of the pass-p rows (n x n, other rows empty)
template <bool FILL>
__global__ __launch_bounds__(256) void k_mp_m(int n, int p, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                              const unsigned char *__restrict__ sm, const int *__restrict__ pass, const double *__restrict__ alfa,
                                              int *__restrict__ cnt, const int *__restrict__ mrp, int *__restrict__ mcj, double *__restrict__ mv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int q = FILL ? mrp[i] : 0;
   if (pass[i] == p)
   {
      const double a = alfa[i];
      for (int k = rp[i]; k < rp[i + 1]; k++)
         if (sm[k] && pass[cj[k]] == p - 1)
         {
            if (FILL) { mcj[q] = cj[k]; mv[q] = a * v[k]; }
            q++;
         }
   }
   if (!FILL) cnt[i] = q;
}
// W := rows of T where pass == p, rows of W elsewhere
template <bool FILL>
__global__ __launch_bounds__(256) void k_mp_take(int n, int p, const int *__restrict__ pass, const int *__restrict__ wrp, const int *__restrict__ wcj,
                                                 const double *__restrict__ wv, const int *__restrict__ trp, const int *__restrict__ tcj,
                                                 const double *__restrict__ tv, int *__restrict__ cnt, const int *__restrict__ orp, int *__restrict__ ocj,
                                                 double *__restrict__ ov)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const bool    fromT = pass[i] == p;
   const int    *srp = fromT ? trp : wrp, *scj = fromT ? tcj : wcj;
   const double *sv  = fromT ? tv : wv;
   const int     s = srp[i], e = srp[i + 1];
   if (!FILL) { cnt[i] = e - s; return; }
   int q = orp[i];
   for (int k = s; k < e; k++, q++) { ocj[q] = scj[k]; ov[q] = sv[k]; }
}

// hypre_BoomerAMGInterpTruncation on a finished row, in place in the CSR arrays (thread per row; the same steps, sums and tie order
// as the truncation inside the extended+i kernels and as the oracle's orc_truncate_row): relative threshold, the pmax largest,
// row sum kept; the survivors are left column-sorted at the front of the row's slice.  cnt[i] = entries kept.
__device__ void agg_qsort_abs(int *L, double *W, int n, int *stack)
{ // descending |w|, K&R form: pivot = middle element swapped to the front, strict '>' partition, smaller partition first
   int sp = 0;
   stack[sp++] = 0;
   stack[sp++] = n - 1;
   while (sp > 0)
   {
      int right = stack[--sp], left = stack[--sp];
      while (left < right)
      {
         const int mid = (left + right) / 2;
         int       last = left, ti;
         double    td;
         ti = L[left]; L[left] = L[mid]; L[mid] = ti;
         td = W[left]; W[left] = W[mid]; W[mid] = td;
         const double piv = fabs(W[left]);
         for (int i = left + 1; i <= right; i++)
            if (fabs(W[i]) > piv)
            {
               ++last;
               ti = L[last]; L[last] = L[i]; L[i] = ti;
               td = W[last]; W[last] = W[i]; W[i] = td;
            }
         ti = L[left]; L[left] = L[last]; L[last] = ti;
         td = W[left]; W[left] = W[last]; W[last] = td;
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
__global__ __launch_bounds__(256) void k_agg_truncate(int n, const int *__restrict__ rp, int *cj, double *v, int pmax, double trunc_factor,
                                                      int *__restrict__ cnt_out)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int    *L   = cj + rp[i];
   double *W   = v + rp[i];
   int     cnt = rp[i + 1] - rp[i];
   if (trunc_factor > 0.0 && cnt > 0)
   {
      double mx = 0.0, tot = 0.0, kept = 0.0;
      for (int q = 0; q < cnt; q++)
      {
         if (fabs(W[q]) > mx) mx = fabs(W[q]);
         tot += W[q];
      }
      int c2 = 0;
      for (int q = 0; q < cnt; q++)
         if (fabs(W[q]) >= trunc_factor * mx)
         {
            L[c2] = L[q];
            W[c2] = W[q];
            kept += W[c2];
            c2++;
         }
      cnt = c2;
      if (kept != 0.0)
      {
         const double sc = tot / kept;
         for (int q = 0; q < cnt; q++) W[q] *= sc;
      }
   }
   if (pmax > 0 && cnt > pmax)
   {
      double tot = 0.0, kept = 0.0;
      for (int q = 0; q < cnt; q++) tot += W[q];
      int stk[64];
      agg_qsort_abs(L, W, cnt, stk);
      cnt = pmax;
      for (int a = 1; a < cnt; a++) // kept set -> column order before summing
      {
         const int    cc = L[a];
         const double ww = W[a];
         int          b  = a - 1;
         while (b >= 0 && L[b] > cc) { L[b + 1] = L[b]; W[b + 1] = W[b]; b--; }
         L[b + 1] = cc;
         W[b + 1] = ww;
      }
      for (int q = 0; q < cnt; q++) kept += W[q];
      if (kept != 0.0)
      {
         const double sc = tot / kept;
         for (int q = 0; q < cnt; q++) W[q] *= sc;
      }
   }
   cnt_out[i] = cnt;
}
__global__ __launch_bounds__(256) void k_agg_compact(int n, const int *__restrict__ srp, const int *__restrict__ scj, const double *__restrict__ sv,
                                                     const int *__restrict__ drp, int *__restrict__ dcj, double *__restrict__ dv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int s = srp[i], d = drp[i], c = drp[i + 1] - d;
   for (int k = 0; k < c; k++) { dcj[d + k] = scj[s + k]; dv[d + k] = sv[s + k]; }
}

// rowptr = exclusive scan of cnt; allocates col / val; returns nnz
int finish_rows(int nrows, int ncols, DArray<int> &cnt, DCsr &M)
{
   M.nrows = nrows;
   M.ncols = ncols;
   M.rowptr.alloc((size_t)nrows + 1);
   require_int32_total(nrows, cnt.data(), "aggressive coarsening");
   exclusive_scan(nrows, cnt.data(), M.rowptr.data(), nullptr);
   HDA_HIP(hipMemcpyAsync(&M.nnz, M.rowptr.data() + nrows, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   M.col.alloc((size_t)std::max(M.nnz, 1));
   M.val.alloc((size_t)std::max(M.nnz, 1));
   return M.nnz;
}

// ---- mm-ext+i (interpolation type 17): W = -D^-1 (I + B) A^s_FC as sparse products (oracle: orc_interp_mm_extpi_dof)
// q_k = sum of the strong C entries of row k (F rows; column order)
__global__ __launch_bounds__(256) void k_mm_q(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                              const unsigned char *__restrict__ sm, const int *__restrict__ cf, double *__restrict__ q,
                                              int *__restrict__ nsc)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   double s = 0.0;
   int    c = 0;
   if (cf[i] == kF)
      for (int k = rp[i]; k < rp[i + 1]; k++)
         if (sm[k] && cf[cj[k]] == kC) { s += v[k]; c++; }
   q[i]   = s;
   nsc[i] = c; // entries of row i of A^s_FC
}
// one walk over row i in column order (FILL = false: counts only): the entries of row i of I + B -- (i, 1) and (k, b_ik) for the
// strong F neighbours with a non-zero denominator -- and d_i
template <bool FILL>
__global__ __launch_bounds__(256) void k_mm_rows(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                 const unsigned char *__restrict__ sm, const int *__restrict__ cf, const int *__restrict__ dof,
                                                 const double *__restrict__ q, int *__restrict__ cnt, const int *__restrict__ brp,
                                                 int *__restrict__ bcj, double *__restrict__ bv, double *__restrict__ dd)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   if (cf[i] != kF)
   {
      if (!FILL) cnt[i] = 0;
      return;
   }
   double d = 0.0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (cj[k] == i) d = v[k];
   int o = FILL ? brp[i] : 0, c = 0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      const int    j   = cj[k];
      const double aij = v[k];
      if (j == i)
      {
         if (FILL) { bcj[o] = i; bv[o] = 1.0; o++; }
         c++;
      }
      else if (sm[k] && cf[j] == kF)
      {
         double ski = 0.0;
         for (int kk = rp[j]; kk < rp[j + 1]; kk++)
            if (cj[kk] == i && sm[kk]) ski = v[kk];
         const double den = q[j] + ski;
         if (den != 0.0)
         {
            const double coef = aij / den;
            d += coef * ski;
            if (FILL) { bcj[o] = j; bv[o] = coef; o++; }
            c++;
         }
         else d += aij;
      }
      else if (sm[k] && cf[j] == kC) {}
      else if (cf[j] != -3 && !(dof && dof[j] != dof[i])) d += aij; // weak (or strong towards a point that is neither C nor F): lumped
   }
   if (FILL) dd[i] = d;
   else cnt[i] = c;
}
// A^s_FC: row k = the strong C entries of row k (F rows), columns renumbered to coarse ids
__global__ __launch_bounds__(256) void k_mm_fc_fill(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                    const unsigned char *__restrict__ sm, const int *__restrict__ cf, const int *__restrict__ cidx,
                                                    const int *__restrict__ frp, int *__restrict__ fcj, double *__restrict__ fv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n || cf[i] != kF) return;
   int o = frp[i];
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (sm[k] && cf[cj[k]] == kC) { fcj[o] = cidx[cj[k]]; fv[o] = v[k]; o++; }
}
__global__ __launch_bounds__(256) void k_mm_p_count(int n, const int *__restrict__ cf, const int *__restrict__ trp, int *__restrict__ cnt)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) cnt[i] = (cf[i] == kC) ? 1 : (cf[i] == kF ? trp[i + 1] - trp[i] : 0);
}
__global__ __launch_bounds__(256) void k_mm_p_fill(int n, const int *__restrict__ cf, const int *__restrict__ cidx, const int *__restrict__ trp,
                                                   const int *__restrict__ tcj, const double *__restrict__ tv, const double *__restrict__ dd,
                                                   const int *__restrict__ prp, int *__restrict__ pcj, double *__restrict__ pv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int o = prp[i];
   if (cf[i] == kC) { pcj[o] = cidx[i]; pv[o] = 1.0; return; }
   if (cf[i] != kF) return;
   const double d = dd[i];
   for (int k = trp[i]; k < trp[i + 1]; k++, o++)
   {
      pcj[o] = tcj[k];
      pv[o]  = (d != 0.0) ? tv[k] / (-d) : tv[k];
   }
}

} // namespace

void amg_truncate_rows(DCsr &P, int pmax, double trunc_factor)
{
   if ((pmax <= 0 && trunc_factor <= 0.0) || P.nrows == 0) return;
   const int   n = P.nrows, g = ceil_div(n, 256);
   DArray<int> cnt((size_t)n + 1);
   cnt.zero();
   k_agg_truncate<<<g, 256, 0, STREAM>>>(n, P.rowptr.data(), P.col.data(), P.val.data(), pmax, trunc_factor, cnt.data());
   DCsr Q;
   finish_rows(n, P.ncols, cnt, Q);
   k_agg_compact<<<g, 256, 0, STREAM>>>(n, P.rowptr.data(), P.col.data(), P.val.data(), Q.rowptr.data(), Q.col.data(), Q.val.data());
   P = std::move(Q);
   P.reset_plan();
}

void amg_second_strength(const DCsr &A, const unsigned char *smask, const int *cf, int num_paths, DCsr &S2, DArray<int> &c1)
{
   const int n = A.nrows, g = ceil_div(std::max(n, 1), 256);
   DArray<int> m((size_t)n + 1);
   c1.alloc((size_t)n + 1);
   k_agg_cmark<<<g, 256, 0, STREAM>>>(n, cf, m.data());
   exclusive_scan(n, m.data(), c1.data(), nullptr);
   int n1 = 0;
   HDA_HIP(hipMemcpyAsync(&n1, c1.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   DArray<int> cr((size_t)n1 + 1), cc((size_t)n + 1), cd((size_t)n1 + 1);
   cr.zero(); cc.zero(); cd.zero();
   k_ss_count<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, c1.data(), cr.data(), cc.data(), cd.data());
   DCsr Sr, Sc, D, T;
   finish_rows(n1, n, cr, Sr);
   finish_rows(n, n1, cc, Sc);
   finish_rows(n1, n1, cd, D);
   k_ss_fill<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, c1.data(), Sr.rowptr.data(), Sr.col.data(), Sr.val.data(),
                                   Sc.rowptr.data(), Sc.col.data(), Sc.val.data(), D.rowptr.data(), D.col.data(), D.val.data());
   spgemm(Sr, Sc, T); // number of two-step paths between C points
   const int   g1 = ceil_div(std::max(n1, 1), 256);
   DArray<int> cnt((size_t)n1 + 1);
   cnt.zero();
   k_ss_merge<false><<<g1, 256, 0, STREAM>>>(n1, T.rowptr.data(), T.col.data(), T.val.data(), D.rowptr.data(), D.col.data(), D.val.data(),
                                            (double)num_paths, cnt.data(), nullptr, nullptr, nullptr);
   finish_rows(n1, n1, cnt, S2);
   k_ss_merge<true><<<g1, 256, 0, STREAM>>>(n1, T.rowptr.data(), T.col.data(), T.val.data(), D.rowptr.data(), D.col.data(), D.val.data(),
                                           (double)num_paths, nullptr, S2.rowptr.data(), S2.col.data(), S2.val.data());
}

void amg_coarsen_second_pass(const DCsr &A, const unsigned char *smask, int num_paths, uint64_t seed, int level, int *cf)
{
   DCsr        S2;
   DArray<int> c1;
   amg_second_strength(A, smask, cf, std::max(num_paths, 1), S2, c1);
   if (S2.nrows == 0) return;
   DArray<unsigned char> all((size_t)std::max(S2.nnz, 1));
   k_agg_fill_u8<<<std::min(ceil_div(std::max(S2.nnz, 1), 256), 1 << 16), 256, 0, STREAM>>>(std::max(S2.nnz, 1), 1, all.data());
   DArray<int> cf2((size_t)S2.nrows);
   amg_pmis(S2, all.data(), seed, level + 64, 0, cf2.data());
   k_agg_merge_cf<<<ceil_div(A.nrows, 256), 256, 0, STREAM>>>(A.nrows, c1.data(), cf2.data(), cf);
}

void amg_interp_multipass(const DCsr &A, const unsigned char *smask, const int *cf, DCsr &P)
{
   const int n = A.nrows, g = ceil_div(std::max(n, 1), 256);
   DArray<int> m((size_t)n + 1), cidx((size_t)n + 1), pass((size_t)std::max(n, 1)), counter(1);
   k_agg_cmark<<<g, 256, 0, STREAM>>>(n, cf, m.data());
   exclusive_scan(n, m.data(), cidx.data(), nullptr);
   int nc = 0;
   HDA_HIP(hipMemcpyAsync(&nc, cidx.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   k_mp_init<<<g, 256, 0, STREAM>>>(n, cf, pass.data());
   int npass = 0;
   for (int p = 1;; p++)
   {
      counter.zero();
      k_mp_mark<<<g, 256, 0, STREAM>>>(n, p, A.rowptr.data(), A.col.data(), smask, cf, pass.data(), counter.data());
      k_mp_commit<<<g, 256, 0, STREAM>>>(n, p, pass.data());
      int found = 0;
      counter.download(&found, 1);
      if (!found) break;
      npass = p;
      HDA_REQUIRE(p < 1000, "multipass interpolation: pass numbering did not terminate");
   }
   DArray<double> alfa((size_t)std::max(n, 1));
   DArray<int>    cnt((size_t)n + 1);
   cnt.zero();
   k_mp_alfa<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, pass.data(), alfa.data(), cnt.data());
   DCsr W;
   finish_rows(n, nc, cnt, W);
   k_mp_fill1<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, pass.data(), cidx.data(), alfa.data(), W.rowptr.data(),
                                    W.col.data(), W.val.data());
   for (int p = 2; p <= npass; p++)
   {
      DCsr M, T, W2;
      cnt.zero();
      k_mp_m<false><<<g, 256, 0, STREAM>>>(n, p, A.rowptr.data(), A.col.data(), A.val.data(), smask, pass.data(), alfa.data(), cnt.data(), nullptr,
                                          nullptr, nullptr);
      finish_rows(n, n, cnt, M);
      k_mp_m<true><<<g, 256, 0, STREAM>>>(n, p, A.rowptr.data(), A.col.data(), A.val.data(), smask, pass.data(), alfa.data(), nullptr, M.rowptr.data(),
                                         M.col.data(), M.val.data());
      spgemm(M, W, T);
      cnt.zero();
      k_mp_take<false><<<g, 256, 0, STREAM>>>(n, p, pass.data(), W.rowptr.data(), W.col.data(), W.val.data(), T.rowptr.data(), T.col.data(),
                                             T.val.data(), cnt.data(), nullptr, nullptr, nullptr);
      finish_rows(n, nc, cnt, W2);
      k_mp_take<true><<<g, 256, 0, STREAM>>>(n, p, pass.data(), W.rowptr.data(), W.col.data(), W.val.data(), T.rowptr.data(), T.col.data(),
                                            T.val.data(), nullptr, W2.rowptr.data(), W2.col.data(), W2.val.data());
      W = std::move(W2);
   }
   HDA_TRACE("  multipass interpolation: %d passes, %d x %d, %d entries", npass, n, nc, W.nnz);
   P = std::move(W);
   P.reset_plan();
}

// hypre's mm-ext+i (interpolation type 17, reference src/internal/amg.c:267-268): W = -D^-1 (I + B) A^s_FC, the product on the
// deterministic SpGEMM of the Galerkin operator, then InterpTruncation on the finished rows.  Bit-identical to
// orc_interp_mm_extpi_dof (same order of every sum).
void amg_interp_mm_extpi(const DCsr &A, const unsigned char *smask, const int *cf, int pmax, double trunc_factor, DCsr &P, const int *dof)
{
   const int n = A.nrows, g = ceil_div(std::max(n, 1), 256);
   DArray<int> m((size_t)n + 1), cidx((size_t)n + 1), nsc((size_t)n + 1), cb((size_t)n + 1);
   DArray<double> q((size_t)std::max(n, 1)), dd((size_t)std::max(n, 1));
   k_agg_cmark<<<g, 256, 0, STREAM>>>(n, cf, m.data());
   exclusive_scan(n, m.data(), cidx.data(), nullptr);
   int nc = 0;
   HDA_HIP(hipMemcpyAsync(&nc, cidx.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   nsc.zero();
   cb.zero();
   k_mm_q<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, cf, q.data(), nsc.data());
   k_mm_rows<false><<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, cf, dof, q.data(), cb.data(), nullptr, nullptr,
                                          nullptr, nullptr);
   Context::get().sync();
   DCsr B, FC, T;
   finish_rows(n, n, cb, B);
   finish_rows(n, nc, nsc, FC);
   k_mm_rows<true><<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, cf, dof, q.data(), nullptr, B.rowptr.data(),
                                         B.col.data(), B.val.data(), dd.data());
   k_mm_fc_fill<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, cf, cidx.data(), FC.rowptr.data(), FC.col.data(),
                                      FC.val.data());
   spgemm(B, FC, T); // rows column-sorted, every output entry summed in the order the product enumerates its terms (k ascending)
   DArray<int> pc((size_t)n + 1);
   pc.zero();
   k_mm_p_count<<<g, 256, 0, STREAM>>>(n, cf, T.rowptr.data(), pc.data());
   finish_rows(n, nc, pc, P);
   k_mm_p_fill<<<g, 256, 0, STREAM>>>(n, cf, cidx.data(), T.rowptr.data(), T.col.data(), T.val.data(), dd.data(), P.rowptr.data(), P.col.data(),
                                     P.val.data());
   P.reset_plan();
   amg_truncate_rows(P, pmax, trunc_factor);
}

} // namespace hda
