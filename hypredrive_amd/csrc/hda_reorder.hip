// hda_reorder.hip -- solve-phase renumbering of the coarse levels.
//
// A Galerkin coarse operator inherits the lexicographic order of its fine C points: on a 3-D
// grid the 30-odd neighbours of a coarse unknown then sit tens of thousands of indices apart
// (one coarse "plane" per fine plane) and every x gather of its SpMV touches a dozen cache
// lines (measured: the level-1 product of the 256^3 benchmark ran 23 % faster with the gather
// removed).  The hierarchy itself knows which unknowns are neighbours in space: points that
// interpolate from the same coarse point form a blob.  So after the setup -- which stays in
// natural order, bit-identical to the oracle -- every coarse level is renumbered by the key
//     rank_l(i) = order of ( rank_{l+1}( parent_l(i) ), i ),   parent = strongest interpolation source,
// recursively from a coarse level kept in natural order: a nested clustering, the algebraic
// cousin of a space-filling curve.  A_l, P_l, P_{l-1}, R_l, R_{l-1} are permuted once; vectors
// never need permuting because level 0 keeps the user's order.  The preconditioner is the same
// operator (a permutation similarity), only the order of the row sums changes (rounding).
// Measured on the level-1 operator of 160^3: 3.79 -> 4.5 TB/s.
#include "hda_amg.h"

#include <cstring> // rocprim's texture iterator calls memset on the host
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>

namespace hda {

#define STREAM (Context::get().stream)

namespace {

// strongest interpolation source of every row of P (first maximum in column order); rows
// without entries keep their proportional position
// (row blocks: only columns < nc, the coarse points this rank owns, can be parents)
__global__ __launch_bounds__(256) void k_parent(int n, int nc, const int *__restrict__ rp, const int *__restrict__ cj,
                                                const double *__restrict__ v, int *__restrict__ parent)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int    best = -1;
   double bv   = -1.0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      const double a = fabs(v[k]);
      if (cj[k] < nc && a > bv) { bv = a; best = cj[k]; }
   }
   parent[i] = (best >= 0) ? best : (int)(((long long)i * nc) / max(n, 1));
}
__global__ __launch_bounds__(256) void k_keys(int n, const int *__restrict__ parent, const int *__restrict__ rank_next,
                                              unsigned long long *__restrict__ keys)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const unsigned r = rank_next ? (unsigned)rank_next[parent[i]] : (unsigned)parent[i];
   keys[i]          = ((unsigned long long)r << 32) | (unsigned)i;
}
__global__ __launch_bounds__(256) void k_unpack(int n, const unsigned long long *__restrict__ keys, int *__restrict__ perm, int *__restrict__ rank)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   const int i = (int)(keys[q] & 0xffffffffu);
   perm[q]     = i; // new position q holds old unknown i
   rank[i]     = q;
}
__global__ __launch_bounds__(256) void k_perm_len(int n, const int *__restrict__ perm, const int *__restrict__ rp, int *__restrict__ len)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) { const int i = perm ? perm[q] : q; len[q] = rp[i + 1] - rp[i]; }
}
// LPR lanes copy one row: the source rows are scattered, a lane group reads each of them coalesced
template <int LPR>
__global__ __launch_bounds__(256) void k_perm_copy(int n, const int *__restrict__ perm, const int *__restrict__ col_rank, int nown,
                                                   const int *__restrict__ srp, const int *__restrict__ scj, const double *__restrict__ sv,
                                                   const int *__restrict__ drp, int *__restrict__ dcj, double *__restrict__ dv)
{
   const int lane = threadIdx.x & (LPR - 1);
   for (long q = ((long)blockIdx.x * 256 + threadIdx.x) / LPR; q < n; q += (long)gridDim.x * 256 / LPR)
   {
      const int i = perm ? perm[q] : (int)q;
      const int s = srp[i], e = srp[i + 1], d = drp[q];
      for (int k = s + lane; k < e; k += LPR)
      {
         const int c      = scj[k];
         dcj[d + (k - s)] = (col_rank && c < nown) ? col_rank[c] : c; // ghost columns keep their slot
         dv[d + (k - s)]  = sv[k];
      }
   }
}

__global__ __launch_bounds__(256) void k_gather_i(int n, const int *__restrict__ idx, const int *__restrict__ src, int *__restrict__ dst)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) dst[q] = src[idx[q]];
}

// out = rows of M taken in the order perm (new -> old; null = unchanged), owned columns (< nown)
// renamed by col_rank (old -> new; null = unchanged)
void permute_csr(DCsr &M, const int *perm, const int *col_rank, int nown)
{
   const int n = M.nrows;
   DCsr      out;
   out.nrows = n;
   out.ncols = M.ncols;
   out.nnz   = M.nnz;
   out.rowptr.alloc((size_t)n + 1);
   out.col.alloc((size_t)std::max(M.nnz, 1));
   out.val.alloc((size_t)std::max(M.nnz, 1));
   DArray<int> len((size_t)n + 1);
   len.zero();
   if (n) k_perm_len<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, perm, M.rowptr.data(), len.data());
   exclusive_scan(n, len.data(), out.rowptr.data(), nullptr);
   if (n)
   {
      const int grid = std::min(ceil_div((long long)n * 8, 256), 1 << 16);
      if (M.avg_row() > 12.0)
         k_perm_copy<32><<<std::min(ceil_div((long long)n * 32, 256), 1 << 16), 256, 0, STREAM>>>(n, perm, col_rank, nown, M.rowptr.data(), M.col.data(),
                                                                                            M.val.data(), out.rowptr.data(), out.col.data(),
                                                                                            out.val.data());
      else
         k_perm_copy<8><<<grid, 256, 0, STREAM>>>(n, perm, col_rank, nown, M.rowptr.data(), M.col.data(), M.val.data(), out.rowptr.data(),
                                                  out.col.data(), out.val.data());
   }
   // Column-sorting the renamed rows again costs 50-80 ms of setup at 256^3 and buys 1.6 % of solve time
   // (39.4 vs 40.0 ms): off unless asked for.  Nothing in the solve phase needs sorted rows.
   constexpr bool resort = false;
   if (col_rank && resort) sort_rows(out);
   M = std::move(out);
   M.reset_plan();
}

__global__ __launch_bounds__(256) void k_remap(int n, const int *__restrict__ rank, int *__restrict__ idx)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) idx[q] = rank[idx[q]];
}
// the owned unknowns a halo plan packs are renamed with their level
void remap_halo(HaloPlan &h, const int *rank)
{
   if (h.send_total) k_remap<<<ceil_div(h.send_total, 256), 256, 0, STREAM>>>(h.send_total, rank, h.send_idx.data());
}

} // namespace

void Amg::reorder_levels()
{
   static const long long min_rows = getenv("HDA_REORDER") ? atoll(getenv("HDA_REORDER")) : 50000; // 0 disables
   const int L = num_levels();
   if (min_rows <= 0 || L < 3) return;
   if (dist)
   { // the level-by-level self check compares two setups entry by entry: keep both in natural order
      const char *chk = getenv("HDA_DIST_CHECK");
      if (chk && *chk && *chk != '0') return;
   }
   // Gauss-Seidel sweeps depend on the numbering: hypre's semantics are the natural order
   auto gs = [](int t) { return t == 3 || t == 4 || t == 6 || t == 8 || t == 13 || t == 14; };
   if (gs(prm.relax_down) || gs(prm.relax_up) || gs(prm.relax_coarse)) return;
   if (prm.smooth_num_levels > 1) return; // so does an ILU factorisation (level 0 is never renumbered)
   // Row blocks (dist): only the unknowns this rank owns are renumbered -- ghost slots are numbered
   // by global id and stay -- and the level handed over to the replicated tail keeps its order.
   // nown(l) = owned unknowns of level l; the last level of `levels` is the coarsest / hand-over one.
   auto nown = [&](int l) { return (l == L - 1) ? (dist ? coarse_nloc : level_A(l).nrows) : level_A(l).nrows; };
   // levels 1 .. last with at least min_rows rows are renumbered
   int last = 0;
   for (int l = 1; l <= L - 2; l++)
      if (level_A(l).nrows >= min_rows) last = l;
   if (last < 1) return;
   // ranks are computed all the way up from the coarsest level (natural order there), so that
   // clusters are nested at every scale; only levels 1..last are actually permuted
   const int top = L - 2; // deepest level that has a P (recursing only 1-2 levels below `last` measured 1 % slower)
   std::vector<DArray<int>> perm((size_t)top + 2), rank((size_t)top + 2); // index = level
   for (int l = top; l >= 1; l--)
   {
      const DCsr &P = levels[(size_t)l].P; // level l -> l+1
      const int   n = P.nrows, nc = dist ? nown(l + 1) : P.ncols;
      DArray<int> parent((size_t)std::max(n, 1));
      DArray<unsigned long long> keys((size_t)std::max(n, 1)), sorted((size_t)std::max(n, 1));
      k_parent<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, nc, P.rowptr.data(), P.col.data(), P.val.data(), parent.data());
      k_keys<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, parent.data(), (l + 1 <= top) ? rank[(size_t)l + 1].data() : nullptr, keys.data());
      size_t tmp_bytes = 0;
      HDA_ROCPRIM(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys.data(), sorted.data(), (size_t)n, 0, 64, STREAM));
      DArray<char> tmp(std::max<size_t>(tmp_bytes, 1));
      HDA_ROCPRIM(rocprim::radix_sort_keys(tmp.data(), tmp_bytes, keys.data(), sorted.data(), (size_t)n, 0, 64, STREAM));
      perm[(size_t)l].alloc((size_t)std::max(n, 1));
      rank[(size_t)l].alloc((size_t)std::max(n, 1));
      k_unpack<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, sorted.data(), perm[(size_t)l].data(), rank[(size_t)l].data());
   }
   for (int l = 1; l <= last; l++)
   {
      AmgLevel  &lv = levels[(size_t)l];
      const int *pl = perm[(size_t)l].data(), *rl = rank[(size_t)l].data();
      const int *rn = (l + 1 <= last) ? rank[(size_t)l + 1].data() : nullptr, *pn = (l + 1 <= last) ? perm[(size_t)l + 1].data() : nullptr;
      const int nl = nown(l), nn = nown(l + 1);
      permute_csr(lv.A, pl, rl, nl);                  // A_l: rows and owned columns
      permute_csr(lv.P, pl, rn, nn);                  // P_l: rows level l, columns level l+1
      if (lv.pg_ready && lv.Pg.nrows && rn) permute_csr(lv.Pg, nullptr, rn, nn); // its ghost rows keep their slots, owned columns move
      permute_csr(lv.R, pn, rl, nl);                  // R_l: rows level l+1, columns level l
      if (dist)
      { // the owned unknowns the halo plans pack: inputs of A_l and R_l are level-l vectors, of P_l level l+1
         remap_halo(lv.hA, rl);
         remap_halo(lv.hR, rl);
         if (rn) remap_halo(lv.hP, rn);
      }
      if (l == 1)
      { // level 0 keeps the user's numbering: only the level-1 side of its transfer operators moves
         permute_csr(levels[0].P, nullptr, rl, nl);
         if (levels[0].pg_ready && levels[0].Pg.nrows) permute_csr(levels[0].Pg, nullptr, rl, nl);
         permute_csr(levels[0].R, pl, nullptr, 0);
         if (dist) remap_halo(levels[0].hP, rl);
      }
      // the C/F marker of the level follows its rows (kept consistent for level_cf())
      if (lv.cf.size() >= (size_t)lv.A.nrows && lv.A.nrows)
      {
         DArray<int> cf2((size_t)lv.A.nrows);
         k_gather_i<<<ceil_div(lv.A.nrows, 256), 256, 0, STREAM>>>(lv.A.nrows, pl, lv.cf.data(), cf2.data());
         lv.cf = std::move(cf2);
      }
   }
   reordered_levels = last;
   HDA_TRACE("solve-phase renumbering of levels 1..%d", last);
}

} // namespace hda
