// hda_gs.hip -- hybrid Gauss-Seidel sweeps (hypre relax types 3/4/6 and the l1 variants
// 13/14/8) with exactly hypre's semantics: Gauss-Seidel over the rank's
// own rows in ascending (forward) or descending (backward) order, ghost values frozen for
// the sweep ("hybrid" = Jacobi across ranks).
//
// A sequential sweep is a DAG: row i must see the new values of its lower neighbours and the
// old values of its upper ones.  Rows are therefore grouped into dependency levels of the
// symmetrised local pattern (level(i) = 1 + max level of lower neighbours); rows of one level
// are mutually non-adjacent, so a level is a data-parallel launch and the result equals the
// sequential sweep.  A backward sweep walks the same levels in reverse.  Runs of small
// levels are fused into one single-workgroup launch separated by workgroup barriers, so the
// ramp-up/ramp-down of the wavefront (and whole coarse grids) do not pay one launch per level.
#include "hda_amg.h"

#include <cstring> // rocprim's texture iterator calls memset on the host
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <mutex>

namespace hda {

#define STREAM (Context::get().stream)

// distinct neighbours of row i in pattern(A) U pattern(A^T), restricted to owned columns,
// visited in ascending order; f(j) is called once per neighbour j != i
// (lo, hi: only neighbours in [lo, hi) count -- the row's own block in the row-block form, [0, n) otherwise)
// Workgroups w, w + 8, w + 16, ... run on the same XCD (MI355X_MICROARCH: round-robin dispatch over the 8 XCDs, one L2 each).  The two
// kernels below gather / scatter 8-byte values whose 128-byte lines are shared by the positions of ONE row block, which are contiguous
// in q: with workgroup w on positions [256 w, 256 w + 256) every XCD touched every line of x (8 fetches of each line from memory,
// 340 us for the 16.7 M unknowns of 256^3 = 1 TB/s of useful bytes, round-5 trace).  Dealt in chunks instead -- XCD k takes the k-th
// eighth of the positions -- a line is fetched by one L2 only.  The launch is rounded up to a multiple of 8 workgroups.
__device__ __forceinline__ int xcd_chunk_block() { return (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3); }
static inline int xcd_chunk_grid(int n) { return 8 * ceil_div(ceil_div(n, 256), 8); }
template <class F>
__device__ __forceinline__ void for_each_sym_neighbour(int i, int n, const int *__restrict__ rp, const int *__restrict__ cj,
                                                       const int *__restrict__ trp, const int *__restrict__ tcj, F f, int lo = 0,
                                                       int hi = 0x7fffffff)
{
   int a = rp[i], ae = rp[i + 1], t = trp[i], te = trp[i + 1];
   while (a < ae || t < te)
   {
      const int ja = (a < ae) ? cj[a] : 0x7fffffff, jt = (t < te) ? tcj[t] : 0x7fffffff;
      const int j  = min(ja, jt);
      if (ja == j) a++;
      if (jt == j) t++;
      if (j >= n || j >= hi) break; // ghost columns sort last
      if (j != i && j >= lo) f(j);
   }
}

// first row and past-the-end row of the block that holds row i (part: nb + 1 row starts; nullptr = one block)
__device__ __forceinline__ void gs_block_of(const int *__restrict__ part, int nb, int i, int n, int &lo, int &hi)
{
   lo = 0;
   hi = n;
   if (!part) return;
   int a = 0, b = nb;
   while (b - a > 1)
   {
      const int m = (a + b) >> 1;
      if (part[m] <= i) a = m;
      else b = m;
   }
   lo = part[a];
   hi = part[a + 1];
}

__global__ __launch_bounds__(256) void k_gs_indeg(int n, const int *__restrict__ rp, const int *__restrict__ cj,
                                                  const int *__restrict__ trp, const int *__restrict__ tcj, int *__restrict__ indeg,
                                                  int *__restrict__ perm, int *counter, const int *__restrict__ part, int nb)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int d = 0, lo, hi;
   gs_block_of(part, nb, i, n, lo, hi);
   for_each_sym_neighbour(i, n, rp, cj, trp, tcj, [&](int j) { d += (j < i); }, lo, hi);
   indeg[i] = d;
   if (d == 0) perm[atomicAdd(counter, 1)] = i;
}

__global__ __launch_bounds__(256) void k_gs_expand(int nf, const int *__restrict__ frontier, int n, const int *__restrict__ rp,
                                                   const int *__restrict__ cj, const int *__restrict__ trp,
                                                   const int *__restrict__ tcj, int *indeg, int *__restrict__ next, int *counter,
                                                   const int *__restrict__ part, int nb)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= nf) return;
   const int i = frontier[q];
   int       lo, hi;
   gs_block_of(part, nb, i, n, lo, hi);
   for_each_sym_neighbour(i, n, rp, cj, trp, tcj, [&](int j) {
      if (j > i && atomicSub(&indeg[j], 1) == 1) next[atomicAdd(counter, 1)] = j;
   }, lo, hi);
}

// ascending row ids inside every level: deterministic launch contents and better locality
// (the level of a frontier position = the interval of lvl_ptr it lies in: nlev + 1 offsets, found by bisection)
__global__ __launch_bounds__(256) void k_gs_mark(int n, const int *__restrict__ perm, const int *__restrict__ lvl_ptr, int nlev,
                                                 int *__restrict__ level_of_row)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   int a = 0, b = nlev;
   while (b - a > 1)
   {
      const int m = (a + b) >> 1;
      if (lvl_ptr[m] <= q) a = m;
      else b = m;
   }
   level_of_row[perm[q]] = a;
}
// rows in the order (block, level, row): 64-bit keys for a stable radix sort, then the boundaries of the (block, level) groups
__global__ __launch_bounds__(256) void k_gs_keys(int n, const int *__restrict__ level_of_row, const int *__restrict__ part, int nb, int nlev,
                                                 unsigned long long *__restrict__ keys, int *__restrict__ rows)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int q = 0;
   if (part)
   {
      int a = 0, b = nb;
      while (b - a > 1)
      {
         const int m = (a + b) >> 1;
         if (part[m] <= i) a = m;
         else b = m;
      }
      q = a;
   }
   keys[i] = (unsigned long long)q * (unsigned long long)nlev + (unsigned long long)level_of_row[i];
   rows[i] = i;
}
__global__ __launch_bounds__(256) void k_gs_group_flags(int n, const unsigned long long *__restrict__ keys, int *__restrict__ flag)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) flag[q] = (q == 0 || keys[q] != keys[q - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_gs_group_starts(int n, const int *__restrict__ flag, const int *__restrict__ gidx,
                                                         const unsigned long long *__restrict__ keys, int *__restrict__ start,
                                                         unsigned long long *__restrict__ gkey)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n && flag[q]) { start[gidx[q]] = q; gkey[gidx[q]] = keys[q]; }
}
// perm <- the rows sorted by (block, level, row) (part == nullptr: by (level, row)); gstart / gkey: first position and key of every
// non-empty (block, level) group, on the host (their number is levels x blocks, small); gidx (device, n): group of every position
static void gs_order_rows(int n, const DArray<int> &level_of_row, const int *d_part, int nb, int nlev, DArray<int> &perm, std::vector<int> &gstart,
                          std::vector<unsigned long long> &gkey, DArray<int> *gidx_out = nullptr)
{
   DArray<unsigned long long> keys((size_t)n), skeys((size_t)n);
   DArray<int>                rows((size_t)n), flag((size_t)n + 1), gidx((size_t)n + 1);
   k_gs_keys<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, level_of_row.data(), d_part, nb, std::max(nlev, 1), keys.data(), rows.data());
   int bits = 1;
   while (bits < 64 && ((unsigned long long)std::max(nb, 1) * (unsigned long long)std::max(nlev, 1)) >> bits) bits++;
   size_t tmp_bytes = 0;
   HDA_ROCPRIM(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.data(), skeys.data(), rows.data(), perm.data(), (size_t)n, 0, bits, STREAM));
   DArray<char> tmp(std::max<size_t>(tmp_bytes, 1));
   HDA_ROCPRIM(rocprim::radix_sort_pairs(tmp.data(), tmp_bytes, keys.data(), skeys.data(), rows.data(), perm.data(), (size_t)n, 0, bits, STREAM));
   flag.zero();
   k_gs_group_flags<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, skeys.data(), flag.data());
   exclusive_scan(n, flag.data(), gidx.data(), nullptr);
   int ng = 0;
   HDA_HIP(hipMemcpyAsync(&ng, gidx.data() + n, sizeof(int), hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   DArray<int>                dstart((size_t)std::max(ng, 1));
   DArray<unsigned long long> dkey((size_t)std::max(ng, 1));
   k_gs_group_starts<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, flag.data(), gidx.data(), skeys.data(), dstart.data(), dkey.data());
   gstart.resize((size_t)ng);
   gkey.resize((size_t)ng);
   if (ng)
   {
      dstart.download(gstart.data(), (size_t)ng);
      dkey.download(gkey.data(), (size_t)ng);
   }
   if (gidx_out) *gidx_out = std::move(gidx); // (exclusive scan of the flags: position q belongs to group gidx[q + 1] - 1)
}

__global__ __launch_bounds__(256) void k_gs_rowspan(int n, const int *__restrict__ perm, const int *__restrict__ rp, int *__restrict__ rbeg,
                                                    int *__restrict__ rend)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) { rbeg[q] = rp[perm[q]]; rend[q] = rp[perm[q] + 1]; }
}

static void gs_row_spans(const DCsr &A, const GsPlan &plan)
{
   const int n = A.nrows;
   if (plan.rbeg.size() != (size_t)n) { plan.rbeg.alloc((size_t)n); plan.rend.alloc((size_t)n); }
   if (n) k_gs_rowspan<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.perm.data(), A.rowptr.data(), plan.rbeg.data(), plan.rend.data());
   plan.span_rp  = A.rowptr.data();
   plan.span_nnz = A.nnz;
   plan.span_gen = A.gen;
}

// ---- sweep-order copy of the operator for the row-block sweeps of big levels.  A row is stored as whole CHUNKS of four entries
// (padded with zero-valued entries that name the row itself), 16-byte aligned: a lane reads a chunk with one 16-byte load of columns
// and two of values, and the lanes of a wave read neighbouring chunks -- where one entry per lane per load makes every load
// instruction touch every cache line of the wave's rows (18 line look-ups per load instruction and the address unit 60 % busy in
// the first form of this kernel, profiles/r04_gs_blocks.md).
__global__ __launch_bounds__(256) void k_gs_inverse(int n, const int *__restrict__ perm, const int *__restrict__ rp, int *__restrict__ pos_of,
                                                    int *__restrict__ len4)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   const int i = perm[q];
   pos_of[i]   = q;
   len4[q]     = (rp[i + 1] - rp[i] + 3) >> 2;
}
// w[g] = max over the positions q of group g of len[q]; gscan = exclusive scan of the group-start flags (position q is in group gscan[q + 1] - 1)
__global__ __launch_bounds__(256) void k_gs_group_max(int n, const int *__restrict__ gscan, const int *__restrict__ len, int *__restrict__ w)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) atomicMax(&w[gscan[q + 1] - 1], len[q]);
}
// the sweep-order copy keeps the diagonal entry apart (s_aii): a 17-entry row of a coarse level is then four chunks, not five, and the
// barrier-free kernel gives it four lanes instead of eight
__global__ __launch_bounds__(256) void k_gs_inverse_nodiag(int n, const int *__restrict__ perm, const int *__restrict__ rp, const int *__restrict__ cj,
                                                           const double *__restrict__ v, int *__restrict__ pos_of, int *__restrict__ len4,
                                                           double *__restrict__ aii)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   const int i = perm[q];
   pos_of[i]   = q;
   int    len = rp[i + 1] - rp[i];
   double d   = 0.0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (cj[k] == i) { d = v[k]; len--; break; } // (the first diagonal entry; a second one stays an ordinary entry)
   aii[q]  = d;
   len4[q] = (len + 3) >> 2;
}
__global__ __launch_bounds__(256) void k_gs_sorted_fill(int n, int nb, const int *__restrict__ part, const int *__restrict__ perm,
                                                        const int *__restrict__ pos_of, const int *__restrict__ rp, const int *__restrict__ cj,
                                                        const double *__restrict__ v, const int *__restrict__ srp4, int *__restrict__ scj,
                                                        double *__restrict__ sv)
{
   // eight lanes to a row (its entries are read and written side by side), 32 rows to a workgroup; the rows of a block are gathered
   // by one XCD (xcd_chunk_block: their lines are fetched once)
   constexpr int G = 8;
   const int     q = xcd_chunk_block() * (256 / G) + (int)threadIdx.x / G, gl = (int)threadIdx.x % G;
   if (q >= n) return;
   const int i = perm[q];
   int       lo, hi;
   gs_block_of(part, nb, i, n, lo, hi);
   const int r0 = rp[i], r1 = rp[i + 1];
   int       kd = 0x7fffffff; // the row's first diagonal entry: that one is in s_aii, the others keep their order
   for (int k = r0 + gl; k < r1; k += G)
      if (cj[k] == i) { kd = k; break; }
   for (int o = 1; o < G; o <<= 1) kd = min(kd, __shfl_xor(kd, o));
   const int d0 = 4 * srp4[q], e = 4 * srp4[q + 1];
   for (int k = r0 + gl; k < r1; k += G)
   {
      if (k == kd) continue;
      const int c = cj[k], d = d0 + (k - r0) - (k > kd ? 1 : 0);
      scj[d]      = (c >= lo && c < hi) ? pos_of[c] : ~pos_of[c]; // (another block's column: its position in the sweep-START copy)
      sv[d]       = v[k];
   }
   for (int d = d0 + (r1 - r0) - (kd != 0x7fffffff ? 1 : 0) + gl; d < e; d += G) { scj[d] = q; sv[d] = 0.0; } // padding: 0 * (the row's own value)
}
static void gs_sorted_copy(const DCsr &A, const GsPlan &plan)
{
   const int n = A.nrows;
   DArray<int> pos_of((size_t)n), len4((size_t)n + 1);
   if (plan.s_aii.size() != (size_t)n) plan.s_aii.alloc((size_t)n);
   k_gs_inverse_nodiag<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.perm.data(), A.rowptr.data(), A.col.data(), A.val.data(), pos_of.data(), len4.data(),
                                                           plan.s_aii.data());
   plan.s_rowptr.alloc((size_t)n + 1);
   exclusive_scan(n, len4.data(), plan.s_rowptr.data(), nullptr);
   int chunks = 0;
   HDA_HIP(hipMemcpyAsync(&chunks, plan.s_rowptr.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   HDA_REQUIRE(chunks < (1 << 29), "sweep-order copy: too many entries for 32-bit offsets");
   plan.s_col.alloc((size_t)4 * chunks + 4); // (one spare chunk: the kernel reads a row's first chunk unconditionally)
   plan.s_val.alloc((size_t)4 * chunks + 4);
   plan.s_col.zero();
   plan.s_val.zero();
   k_gs_sorted_fill<<<8 * ceil_div(ceil_div(n, 32), 8), 256, 0, STREAM>>>(n, plan.nblk, plan.blk_part.data(), plan.perm.data(), pos_of.data(), A.rowptr.data(),
                                                         A.col.data(), A.val.data(), plan.s_rowptr.data(), plan.s_col.data(), plan.s_val.data());
   if (plan.s_x.size() != (size_t)n) { plan.s_x.alloc((size_t)n); plan.s_b.alloc((size_t)n); plan.s_d.alloc((size_t)n); plan.sd_src = plan.sb_src = nullptr; }
   if (plan.s_x0.size() != (size_t)n) plan.s_x0.alloc((size_t)n);
   plan.sorted = true;
}

// ---- level-wise sweep-order copy (GsPlan::r_*) for k_gs_blocks_ring
constexpr int kGsRing = 4096; // doubles of LDS that hold the iterate of the level just swept (and of the one being swept)
__global__ __launch_bounds__(256) void k_gs_len4(int n, const int *__restrict__ perm, const int *__restrict__ rp, int *__restrict__ len4)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) len4[q] = (rp[perm[q] + 1] - rp[perm[q]] + 3) >> 2;
}
__global__ __launch_bounds__(256) void k_gs_ring_fill(int n, int nb, int nlev_total, const int *__restrict__ part, const int *__restrict__ perm,
                                                      const int *__restrict__ pos_of, const int *__restrict__ rp, const int *__restrict__ cj,
                                                      const double *__restrict__ v, const int *__restrict__ lvl_first,
                                                      const int *__restrict__ lvl_cb, const int *__restrict__ lvl_w, int *__restrict__ rcj,
                                                      double *__restrict__ rv)
{
   const int q = xcd_chunk_block() * 256 + threadIdx.x;
   if (q >= n) return;
   int a = 0, b = nlev_total; // the (block, level) that holds position q: lvl_first[a] <= q < lvl_first[b]
   while (b - a > 1)
   {
      const int m = (a + b) >> 1;
      if (lvl_first[m] <= q) a = m;
      else b = m;
   }
   const int i = perm[q], w = lvl_w[a];
   int       lo, hi;
   gs_block_of(part, nb, i, n, lo, hi);
   long      d = 4L * (lvl_cb[a] + (long)(q - lvl_first[a]) * w);
   const long e = d + 4L * w;
   for (int k = rp[i]; k < rp[i + 1]; k++, d++)
   {
      const int c = cj[k];
      rcj[d]      = (c >= lo && c < hi) ? pos_of[c] : ~pos_of[c]; // (another block's column: its position in the sweep-START copy)
      rv[d]       = v[k];
   }
   for (; d < e; d++) { rcj[d] = q; rv[d] = 0.0; }
}
static void gs_ring_shape(const DCsr &A, const GsPlan &plan, int &lpr, int &nt)
{
   const double a     = A.avg_row();
   lpr                = (a <= 10.0) ? 2 : (a <= 40.0) ? 8 : 16;
   const double lanes = plan.blk_mean_rows_per_level * lpr;
   nt                 = lanes <= 192.0 ? 256 : 512; // (1024 threads leave a wavefront 128 registers: 35.0 against 32.7 ms per 128^3 solve)
}
// passes of the level-wise (ring) kernel, mean over the blocks: every dependency level of a block takes ceil(rows / rows per pass)
static double gs_ring_mean_passes(const GsPlan &plan, int rows_per_pass)
{
   long long np = 0;
   for (size_t g = 0; g + 1 < plan.h_blk_lvl.size(); g++)
      np += std::max((plan.h_blk_lvl[g + 1] - plan.h_blk_lvl[g] + rows_per_pass - 1) / rows_per_pass, 1);
   return (double)np / std::max(plan.nblk, 1);
}
// barrier-free kernel or ring kernel: both cost about a microsecond per step (round of 512 / LPR rows, pass of a dependency level);
// the one with fewer steps runs (measured: 1.06 us per round, 1.42 us per pass on the 128^3 series-B levels)
static bool gs_free_beats_ring(const GsPlan &plan, double ring_passes)
{
   int maxblock = 0;
   for (size_t q = 0; q + 1 < plan.h_part.size(); q++) maxblock = std::max(maxblock, plan.h_part[q + 1] - plan.h_part[q]);
   const double rounds = std::ceil((double)maxblock / (512 / std::max(plan.free_lpr, 1)));
   return rounds <= 1.3 * ring_passes;
}
static void gs_ring_copy(const DCsr &A, const GsPlan &plan)
{
   plan.ring = false;
   const int n = A.nrows, ng = (int)plan.h_blk_lvl.size() - 1; // (block, level) pairs
   if (n == 0 || ng <= 0) return;
   // the ring must hold two consecutive levels of a block, the three per-level tables and the ring must fit the LDS budget
   for (int q = 0; q < plan.nblk; q++)
      for (int g = plan.h_blk_lvl_ptr[(size_t)q]; g + 1 < plan.h_blk_lvl_ptr[(size_t)q + 1]; g++)
      { // three consecutive levels of a block: the two just swept and the one being swept
         const int g3 = std::min(g + 3, plan.h_blk_lvl_ptr[(size_t)q + 1]);
         if (plan.h_blk_lvl[(size_t)g3] - plan.h_blk_lvl[(size_t)g] > kGsRing) return;
      }
   if (plan.blk_max_levels + 1 > 2700) return; // (three tables of that length + the ring within 64 KB of LDS)
   DArray<int> len4((size_t)n), pos_of((size_t)n), dummy((size_t)n);
   k_gs_inverse<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.perm.data(), A.rowptr.data(), pos_of.data(), len4.data());
   // widest row of every (block, level) group: a segmented maximum on the device (the groups of the positions come with the plan)
   std::vector<int> w((size_t)ng, 0);
   {
      HDA_REQUIRE(plan.group_of_pos.size() >= (size_t)n + 1, "Gauss-Seidel plan: group table missing");
      DArray<int> dw((size_t)ng);
      dw.zero();
      k_gs_group_max<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.group_of_pos.data(), len4.data(), dw.data());
      dw.download(w.data(), (size_t)ng);
   }
   std::vector<int> cb((size_t)ng + 1, 0);
   long long        total = 0;
   for (int g = 0; g < ng; g++)
   {
      const int mx = w[(size_t)g];
      cb[(size_t)g] = (int)total;
      total += (long long)mx * (plan.h_blk_lvl[(size_t)g + 1] - plan.h_blk_lvl[(size_t)g]);
      if (total >= (1LL << 29)) return; // 32-bit chunk offsets
   }
   cb[(size_t)ng] = (int)total;
   plan.r_cb.upload(cb.data(), cb.size());
   plan.r_w.upload(w.data(), w.size());
   plan.r_col.alloc((size_t)4 * total + 4); // (one spare chunk: a lane without a chunk of its own reads the level's first one)
   plan.r_val.alloc((size_t)4 * total + 4);
   plan.r_col.zero();
   plan.r_val.zero();
   k_gs_ring_fill<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, plan.nblk, ng, plan.blk_part.data(), plan.perm.data(), pos_of.data(), A.rowptr.data(),
                                                       A.col.data(), A.val.data(), plan.blk_lvl.data(), plan.r_cb.data(), plan.r_w.data(),
                                                       plan.r_col.data(), plan.r_val.data());
   if (plan.s_x.size() != (size_t)n) { plan.s_x.alloc((size_t)n); plan.s_b.alloc((size_t)n); plan.s_d.alloc((size_t)n); plan.sd_src = plan.sb_src = nullptr; }
   plan.s_x0.alloc((size_t)n);
   // lanes per row and workgroup size of this level's sweeps: what a typical level of a block keeps busy (every wavefront of the
   // workgroup runs every pass and meets every barrier, rows or not)
   gs_ring_shape(A, plan, plan.ring_lpr, plan.ring_nt);
   // the pass lists
   {
      const int        RP = plan.ring_nt / plan.ring_lpr;
      std::vector<int> pp((size_t)plan.nblk + 1, 0), rec[2];
      for (int dir = 0; dir < 2; dir++)
      {
         rec[dir].clear();
         for (int q = 0; q < plan.nblk; q++)
         {
            const int g0 = plan.h_blk_lvl_ptr[(size_t)q], g1 = plan.h_blk_lvl_ptr[(size_t)q + 1];
            if (dir == 0) pp[(size_t)q] = (int)(rec[0].size() / 8);
            // positions of the TWO levels swept before a level: what its rows read from the ring (contiguous: levels follow each other)
            int r1lo = 0, r1hi = 0, r2lo = 0, r2hi = 0; // the level before (r1) and the one before that (r2)
            for (int s2 = 0; s2 < g1 - g0; s2++)
            {
               const int g = dir == 0 ? g0 + s2 : g1 - 1 - s2;
               const int first = plan.h_blk_lvl[(size_t)g], rows = plan.h_blk_lvl[(size_t)g + 1] - first, wg = w[(size_t)g];
               const int np = std::max((rows + RP - 1) / RP, 1);
               int       ulo = 0, uhi = 0;
               if (r1hi > r1lo) { ulo = r1lo; uhi = r1hi; }
               if (r2hi > r2lo) { ulo = std::min(ulo, r2lo); uhi = std::max(uhi, r2hi); }
               for (int p2 = 0; p2 < np; p2++)
               {
                  const int j0 = p2 * RP, r = std::min(rows - j0, RP);
                  const int last = (p2 == np - 1);
                  const int v8[8] = {first + j0, std::max(r, 0), cb[(size_t)g] + j0 * wg, wg, ulo, uhi, last, 0};
                  rec[dir].insert(rec[dir].end(), v8, v8 + 8);
               }
               r2lo = r1lo; r2hi = r1hi;
               r1lo = first; r1hi = first + rows;
            }
            // three spare passes without rows behind every block: the pipeline reads three passes ahead
            const int safe_pos = std::min(plan.h_blk_lvl[(size_t)g0 < plan.h_blk_lvl.size() ? (size_t)g0 : 0], std::max(n - 1, 0));
            const int v8[8]    = {safe_pos, 0, (int)total, 0, 0, 0, 0, 0};
            for (int k2 = 0; k2 < 3; k2++) rec[dir].insert(rec[dir].end(), v8, v8 + 8);
         }
         if (dir == 0) pp[(size_t)plan.nblk] = (int)(rec[0].size() / 8);
         plan.r_pass[dir].upload(rec[dir].data(), rec[dir].size());
      }
      plan.r_pass_ptr.upload(pp.data(), pp.size());
   }
   plan.ring = true;
   if (getenv("HDA_VERBOSE"))
      fprintf(stderr, "[hda] block Gauss-Seidel plan: level-wise copy, %.2f padded entries per entry, %d lanes per row, workgroups of %d\n",
              4.0 * (double)total / std::max(A.nnz, 1), plan.ring_lpr, plan.ring_nt);
}

// Row blocks: the dependency levels of a block concern that block alone, so ONE workgroup runs Kahn's algorithm for its block from
// start to end -- frontier after frontier with workgroup barriers, no launch and no host read-back per level (round 5: the global
// level loop below made ~900 launch + read-back round trips over the series-B hierarchy at 256^3, 30-70 ms per operator level
// whatever its size).  lvl[i] = dependency level of row i inside its block; nlev[b] = levels of block b.  fr0 / fr1: frontier
// ping-pong, a block uses the slice [part[b], part[b + 1]) of each.
// Kahn's algorithm on one row block per workgroup.  A row's predecessors are its in-block neighbours of smaller index, in A's row or
// in A^T's: each ENTRY counts (a pair present both ways counts twice and is taken off twice), so the two rows are walked one after
// the other with independent loads -- the round-4 form merged them into one sorted walk, two dependent loads per step, and kept the
// counters in memory (returned global atomics, one round trip each): 122 us per level on the 256^3 level 0.  W > 0: the counters of a
// block as W-bit fields of LDS words -- 16 bits for blocks of up to 32768 rows (a count is below twice the block size), 8 or 4 bits for
// larger blocks whose rows are short enough (a count is at most a row's entries plus its column's: gs_levels measures that); a field
// never borrows: every decrement takes off a unit that was counted.
template <int W> // W: bits of a row's counter in LDS (16, 8 or 4 -- a count must stay below 2^W), 0: counters in memory
__global__ __launch_bounds__(1024) void k_gs_levels_blocks(int n, const int *__restrict__ part, const int *__restrict__ rp, const int *__restrict__ cj,
                                                           const int *__restrict__ trp, const int *__restrict__ tcj, int *indeg, int *fr0, int *fr1,
                                                           int *__restrict__ lvl, int *__restrict__ nlev)
{
   __shared__ int               cnt[2];
   extern __shared__ unsigned int sdeg[];
   const int b = blockIdx.x, lo = part[b], hi = part[b + 1], tid = threadIdx.x;
   constexpr bool LDSDEG = W > 0;
   constexpr int  PW = LDSDEG ? 32 / W : 1; // counters per LDS word
   if (tid < 2) cnt[tid] = 0;
   if constexpr (LDSDEG)
      for (int w = tid; w < (hi - lo + PW - 1) / PW; w += 1024) sdeg[w] = 0u;
   __syncthreads();
   int *cur = fr0 + lo, *nxt = fr1 + lo;
   // G lanes to a row, one entry each per step: a level of a block is a few hundred rows, and what a level costs is the chain
   // row number -> row bounds -> entries -> counter, not the entries themselves
   constexpr int G = 8, NG = 1024 / G;
   const int     grp = tid / G, gl = tid % G;
   for (int i = lo + grp; i < hi; i += NG)
   {
      int       d  = 0;
      const int a0 = rp[i], a1 = rp[i + 1], t0 = trp[i], t1 = trp[i + 1];
      for (int k = a0 + gl; k < a1; k += G) d += (cj[k] >= lo && cj[k] < i);
      for (int k = t0 + gl; k < t1; k += G) d += (tcj[k] >= lo && tcj[k] < i);
      for (int o = 1; o < G; o <<= 1) d += __shfl_xor(d, o);
      if (gl != 0) continue;
      if constexpr (LDSDEG) { if (d) atomicAdd(&sdeg[(i - lo) / PW], (unsigned)d << (W * ((i - lo) % PW))); }
      else indeg[i] = d;
      if (d == 0) cur[atomicAdd(&cnt[0], 1)] = i;
   }
   __syncthreads();
   int level = 0, c = 0;
   for (;;)
   {
      const int nf = cnt[c];
      if (nf == 0) break;
      auto take = [&](int j) { // one predecessor of j is done
         bool last;
         if constexpr (LDSDEG)
         {
            const int      sh  = W * ((j - lo) % PW);
            const unsigned old = atomicSub(&sdeg[(j - lo) / PW], 1u << sh);
            last               = ((old >> sh) & ((1u << W) - 1u)) == 1u;
         }
         else last = atomicSub(&indeg[j], 1) == 1;
         if (last) nxt[atomicAdd(&cnt[c ^ 1], 1)] = j;
      };
      for (int q = grp; q < nf; q += NG)
      {
         const int i = cur[q];
         if (gl == 0) lvl[i] = level;
         const int a0 = rp[i], a1 = rp[i + 1], t0 = trp[i], t1 = trp[i + 1];
         for (int k = a0 + gl; k < a1; k += G)
         {
            const int j = cj[k];
            if (j > i && j < hi) take(j); // (ghost columns are >= n >= hi)
         }
         for (int k = t0 + gl; k < t1; k += G)
         {
            const int j = tcj[k];
            if (j > i && j < hi) take(j);
         }
      }
      __syncthreads();           // the next frontier is complete (and visible: one workgroup, one CU)
      if (tid == 0) cnt[c] = 0;  // (this counter is the one the level after next fills)
      c ^= 1;
      int *t = cur; cur = nxt; nxt = t;
      level++;
      __syncthreads();
   }
   if (tid == 0) nlev[b] = level;
}

// dependency levels of the symmetrised pattern (restricted to the row blocks of part when given): dependency level of every row
// (device array); one block: plan.lvl_ptr gets the level sizes (Kahn's algorithm with one launch and one counter read-back per level);
// row blocks: one workgroup per block (k_gs_levels_blocks), plan.nlev = the most levels any block has.  Everything of size n stays on
// the device since round 5.
__global__ __launch_bounds__(256) void k_gs_max_degree(int n, const int *__restrict__ rp, const int *__restrict__ trp, int *mx)
{ // the most entries any row has in its row and its column together (a bound on its predecessor count)
   int m = 0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = max(m, rp[i + 1] - rp[i] + trp[i + 1] - trp[i]);
   for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
   if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(mx, m);
}
static DArray<int> gs_levels(const DCsr &A, GsPlan &plan, const int *d_part, int nb)
{
   const int n = A.nrows;
   if (d_part && nb > 0)
   { // (the block kernel only enumerates a column's rows: the pattern of A^T, unsorted, without values)
      struct { DArray<int> rowptr, col; } T;
      transpose_pattern_unsorted(A, T.rowptr, T.col);
      DArray<int> indeg((size_t)n), fr0((size_t)n), fr1((size_t)n), lrow((size_t)n), dnl((size_t)nb);
      int maxblock = 0;
      for (size_t q = 0; q + 1 < plan.h_part.size(); q++) maxblock = std::max(maxblock, plan.h_part[q + 1] - plan.h_part[q]);
      HDA_REQUIRE((int)plan.h_part.size() == nb + 1, "Gauss-Seidel plan: row blocks of the host and of the device differ");
      // counter width: 16 bits hold any count of a block of <= 32768 rows; larger blocks get 8 or 4 bits where no row has that many
      // entries in its row and its column together (64 KB of LDS for 16 bits at 32768 rows, 128 KB for 8 bits at 131072 and 4 bits at 262144)
      int W = maxblock <= 32768 ? 16 : 0;
      if (W == 0 && maxblock <= 262144)
      {
         DArray<int> mx(1);
         mx.zero();
         k_gs_max_degree<<<std::min(ceil_div(n, 256), 2048), 256, 0, STREAM>>>(n, A.rowptr.data(), T.rowptr.data(), mx.data());
         int md = 0;
         mx.download(&md, 1);
         if (maxblock <= 131072 && md < 256) W = 8;
         else if (md < 16) W = 4;
      }
#define HDA_GS_LEVELS(WW)                                                                                                                          \
   do                                                                                                                                              \
   {                                                                                                                                               \
      const size_t lds = WW ? sizeof(unsigned) * (size_t)((maxblock + 32 / (WW ? WW : 32) - 1) / (32 / (WW ? WW : 32))) : 0;                       \
      if (lds > 48 * 1024) HDA_HIP(hipFuncSetAttribute((const void *)k_gs_levels_blocks<WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      k_gs_levels_blocks<WW><<<nb, 1024, lds, STREAM>>>(n, d_part, A.rowptr.data(), A.col.data(), T.rowptr.data(), T.col.data(), indeg.data(),     \
                                                        fr0.data(), fr1.data(), lrow.data(), dnl.data());                                          \
   } while (0)
      if (W == 16) HDA_GS_LEVELS(16);
      else if (W == 8) HDA_GS_LEVELS(8);
      else if (W == 4) HDA_GS_LEVELS(4);
      else HDA_GS_LEVELS(0);
#undef HDA_GS_LEVELS
      const std::vector<int> hn = dnl.to_host();
      plan.nlev = 0;
      for (int v : hn) plan.nlev = std::max(plan.nlev, v);
      return lrow;
   }
   DCsr T;
   transpose(A, T); // rows of T = columns of A (sorted: the kernels below walk a row and its column together); only rows < n are consulted
   DArray<int> indeg((size_t)n), counter(1);
   counter.zero();
   k_gs_indeg<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), T.rowptr.data(), T.col.data(), indeg.data(),
                                                   plan.perm.data(), counter.data(), d_part, nb);
   int done = 0, nf = 0;
   counter.download(&nf, 1);
   while (nf > 0)
   {
      plan.lvl_ptr.push_back(done + nf);
      const int *cur = plan.perm.data() + done;
      done += nf;
      if (done >= n) break;
      counter.zero();
      k_gs_expand<<<ceil_div(nf, 256), 256, 0, STREAM>>>(nf, cur, n, A.rowptr.data(), A.col.data(), T.rowptr.data(), T.col.data(),
                                                        indeg.data(), plan.perm.data() + done, counter.data(), d_part, nb);
      counter.download(&nf, 1);
   }
   HDA_REQUIRE(done == n, "Gauss-Seidel level scheduling did not reach every row");
   plan.nlev = (int)plan.lvl_ptr.size() - 1;
   DArray<int> dptr, lrow((size_t)n);
   dptr.upload(plan.lvl_ptr.data(), plan.lvl_ptr.size());
   k_gs_mark<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.perm.data(), dptr.data(), plan.nlev, lrow.data());
   return lrow;
}

void build_gs_plan(const DCsr &A, GsPlan &plan)
{
   const int n = A.nrows;
   plan        = GsPlan();
   plan.built  = true;
   plan.perm.alloc((size_t)std::max(n, 1));
   plan.lvl_ptr.assign(1, 0);
   if (n == 0) return;
   // level of every row -> rows sorted by (level, row): ascending rows inside each level (stable radix sort on the device)
   {
      const DArray<int>               lrow = gs_levels(A, plan, nullptr, 0);
      std::vector<int>                gstart;
      std::vector<unsigned long long> gkey;
      gs_order_rows(n, lrow, nullptr, 0, plan.nlev, plan.perm, gstart, gkey);
      HDA_REQUIRE((int)gstart.size() == plan.nlev, "Gauss-Seidel plan: a dependency level without rows");
   }
   // launch segments: big levels alone, runs of small levels fused into one workgroup
   const int small = 512;
   int       L     = 0;
   while (L < plan.nlev)
   {
      const int sz = plan.lvl_ptr[(size_t)L + 1] - plan.lvl_ptr[(size_t)L];
      if (sz > small) { plan.segments.push_back({L, L + 1}); L++; continue; }
      int E = L;
      while (E < plan.nlev && plan.lvl_ptr[(size_t)E + 1] - plan.lvl_ptr[(size_t)E] <= small) E++;
      plan.segments.push_back({L, E});
      L = E;
   }
   plan.d_lvl_ptr.upload(plan.lvl_ptr.data(), plan.lvl_ptr.size());
   gs_row_spans(A, plan);
   Context::get().sync();
}

// Row-block form (hypre's hybrid sweep at np = V on one GPU): the dependency levels of every block's OWN pattern -- connections
// that leave a block carry the values of the sweep's start, so they order nothing -- and the rows sorted by (block, level, row).
static void gs_free_plan(const GsPlan &plan, int n);
void build_gs_plan_blocks(const DCsr &A, const std::vector<int> &part, GsPlan &plan)
{
   const int n = A.nrows, nb = (int)part.size() - 1;
   HDA_REQUIRE(nb >= 1 && part.front() == 0 && part.back() == n, "row blocks must cover the rows of the operator");
   plan       = GsPlan();
   plan.built = true;
   plan.nblk  = nb;
   plan.perm.alloc((size_t)std::max(n, 1));
   plan.lvl_ptr.assign(1, 0);
   plan.blk_part.upload(part.data(), part.size());
   plan.h_part = part;
   std::vector<int> bl_ptr((size_t)nb + 1, 0), bl;
   if (n == 0)
   {
      bl.push_back(0);
      plan.blk_lvl_ptr.upload(bl_ptr.data(), bl_ptr.size());
      plan.blk_lvl.upload(bl.data(), bl.size());
      return;
   }
   for (int q = 0; q < nb; q++) HDA_REQUIRE(part[(size_t)q] <= part[(size_t)q + 1], "row blocks must ascend");
   // rows sorted by (block, level, row) on the device (round 5: the host made this order from downloaded level numbers, O(n) loops at
   // 16.7 M rows); what comes back are the first positions of the (block, level) groups -- blocks x levels of them
   {
      HDA_TRACE("  gs plan: start n=%d", n);
      const DArray<int>               lrow = gs_levels(A, plan, plan.blk_part.data(), nb);
      HDA_TRACE("  gs plan: levels done");
      std::vector<int>                gstart;
      std::vector<unsigned long long> gkey;
      gs_order_rows(n, lrow, plan.blk_part.data(), nb, plan.nlev, plan.perm, gstart, gkey, &plan.group_of_pos);
      HDA_TRACE("  gs plan: rows ordered");
      const unsigned long long nl = (unsigned long long)std::max(plan.nlev, 1);
      size_t                   g  = 0;
      for (int q = 0; q < nb; q++)
      { // a block's levels 0 .. nl_q - 1 are all present (a row of level L has a neighbour of level L - 1 in its block)
         int nlq = 0;
         while (g < gstart.size() && (int)(gkey[g] / nl) == q)
         {
            HDA_REQUIRE((int)(gkey[g] % nl) == nlq, "Gauss-Seidel plan: a block skips a dependency level");
            bl.push_back(gstart[g]);
            nlq++;
            g++;
         }
         bl_ptr[(size_t)q + 1] = (int)bl.size();
         plan.blk_max_levels   = std::max(plan.blk_max_levels, nlq);
      }
      HDA_REQUIRE(g == gstart.size(), "Gauss-Seidel plan: (block, level) groups out of order");
   }
   bl.push_back(n);
   plan.blk_mean_rows_per_level = (double)n / (double)std::max<size_t>(bl.size() - 1, 1);
   plan.blk_lvl_ptr.upload(bl_ptr.data(), bl_ptr.size());
   plan.blk_lvl.upload(bl.data(), bl.size());
   plan.h_blk_lvl     = bl;
   plan.h_blk_lvl_ptr = bl_ptr;
   gs_row_spans(A, plan);
   HDA_TRACE("  gs plan: tables uploaded");
   const int sorted_min = getenv("HDA_GS_SORTED_MIN") ? atoi(getenv("HDA_GS_SORTED_MIN")) : 2000; // (read per plan: the tests move it)
   if (n >= sorted_min && plan.blk_max_levels + 1 <= 12 * 1024) gs_sorted_copy(A, plan); // (the kernel keeps a block's level offsets in LDS: 48 KB)
   HDA_TRACE("  gs plan: sweep-order copy");
   if (plan.sorted)
   {
      if (!plan.s_x0.size()) plan.s_x0.alloc((size_t)n);
      gs_free_plan(plan, n);
   }
   HDA_TRACE("  gs plan: barrier-free plan");
   // the level-wise copy (a second padded copy of the operator, 6 - 9 ms to fill at 256^3) only where its kernel will run: not where
   // the barrier-free kernel takes fewer steps -- the same rule gs_sweep_blocks applies (HDA_GS_FREE set: both, the tests force either)
   bool want_ring = plan.sorted && !(getenv("HDA_GS_RING") && atoi(getenv("HDA_GS_RING")) == 0);
   if (want_ring && plan.free_lpr > 0 && !getenv("HDA_GS_FREE"))
   {
      int lpr = 2, nt = 256;
      gs_ring_shape(A, plan, lpr, nt);
      if (gs_free_beats_ring(plan, gs_ring_mean_passes(plan, nt / lpr))) want_ring = false;
   }
   if (want_ring) gs_ring_copy(A, plan);
   HDA_TRACE("  gs plan: level-wise copy");
   if (getenv("HDA_VERBOSE"))
      fprintf(stderr, "[hda] block Gauss-Seidel plan: n=%d nnz=%d (%.1f per row), %d blocks, dependency levels per block: max %d, mean %.0f (%.0f rows per level)%s\n",
              n, A.nnz, A.avg_row(), nb, plan.blk_max_levels, (double)(bl.size() - 1) / nb, (double)n / std::max<size_t>(bl.size() - 1, 1),
              plan.sorted ? ", sweep-order copy" : "");
   Context::get().sync();
}

// x_i += dinv_i * (b_i - sum_j a_ij x_j) for the rows of one level; LPR lanes per row
template <int LPR>
__device__ __forceinline__ void gs_rows(int first, int count, int tid, int nthreads, const int *__restrict__ perm,
                                        const int *__restrict__ rbeg, const int *__restrict__ rend, const int *__restrict__ cj,
                                        const double *__restrict__ v, const double *__restrict__ dinv, const double *__restrict__ b, double *x)
{
   const int lane = tid & (LPR - 1);
   for (int q = tid / LPR; q < count; q += nthreads / LPR)
   {
      const int i = perm[first + q], k0 = rbeg[first + q], k1 = rend[first + q]; // three independent coalesced loads
      double    s = 0.0;
      int       k = k0 + lane;
      for (; k + 3 * LPR < k1; k += 4 * LPR)
      { // four entries in flight per lane (a level is a chain of dependent round trips: row -> entries -> x); same order of additions
         const int    c0 = cj[k], c1 = cj[k + LPR], c2 = cj[k + 2 * LPR], c3 = cj[k + 3 * LPR];
         const double a0 = v[k], a1 = v[k + LPR], a2 = v[k + 2 * LPR], a3 = v[k + 3 * LPR];
         const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
         s += a0 * x0;
         s += a1 * x1;
         s += a2 * x2;
         s += a3 * x3;
      }
      for (; k < k1; k += LPR) s += v[k] * x[cj[k]];
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) x[i] += dinv[i] * (b[i] - s);
   }
}

template <int LPR>
__global__ __launch_bounds__(256) void k_gs_level(int first, int count, const int *__restrict__ perm, const int *__restrict__ rbeg,
                                                  const int *__restrict__ rend, const int *__restrict__ cj, const double *__restrict__ v,
                                                  const double *__restrict__ dinv, const double *__restrict__ b, double *x)
{
   gs_rows<LPR>(first, count, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256, perm, rbeg, rend, cj, v, dinv, b, x);
}

// levels [l0, l1) by ONE workgroup of 16 wavefronts, forward or backward, a barrier between levels
// (requesting the next level's row ids, spans and first entries before the barrier was measured: no gain, 363 -> 372 ms at 128^3)
template <int LPR>
__global__ __launch_bounds__(1024) void k_gs_levels_fused(int l0, int l1, int backward, const int *__restrict__ lvl_ptr,
                                                         const int *__restrict__ perm, const int *__restrict__ rbeg,
                                                         const int *__restrict__ rend, const int *__restrict__ cj, const double *__restrict__ v,
                                                         const double *__restrict__ dinv, const double *__restrict__ b, double *x)
{
   for (int s = 0; s < l1 - l0; s++)
   {
      const int L = backward ? (l1 - 1 - s) : (l0 + s);
      gs_rows<LPR>(lvl_ptr[L], lvl_ptr[L + 1] - lvl_ptr[L], threadIdx.x, blockDim.x, perm, rbeg, rend, cj, v, dinv, b, x);
      __threadfence_block();
      __syncthreads();
   }
}

// The same run of small levels, software-pipelined.  A level is a chain of dependent round trips to memory (row ids and spans ->
// entries -> x), and a run is a thousand levels of a few dozen rows: what it costs is that chain, not the arithmetic.  Here a
// "pass" (the rows of a level the workgroup takes at once) goes through three stages, each one pass ahead of the next:
//   A  row id and span                         (independent of x)
//   B  the lane's first NPF entries, divisor, right-hand side   (independent of x)
//   C  gather x, sum in entry order, update x  (needs every earlier level: the barrier)
// so that between two barriers only the gather of x is waited for.  Same arithmetic, same order of additions.
template <int LPR, int NPF>
__global__ __launch_bounds__(1024) void k_gs_levels_pipe(int l0, int l1, int backward, const int *__restrict__ lvl_ptr,
                                                         const int *__restrict__ perm, const int *__restrict__ rbeg,
                                                         const int *__restrict__ rend, const int *__restrict__ cj, const double *__restrict__ v,
                                                         const double *__restrict__ dinv, const double *__restrict__ b, double *x)
{
   extern __shared__ int slp[]; // lvl_ptr[l0 .. l1]
   const int nl = l1 - l0, tid = threadIdx.x, lane = tid & (LPR - 1), q = tid / LPR;
   constexpr int RP = 1024 / LPR; // rows per pass
   for (int t = tid; t <= nl; t += 1024) slp[t] = lvl_ptr[l0 + t];
   __syncthreads();
   struct It { int s, p; }; // level of the run in sweep order, pass inside the level
   auto level = [&](const It &it) { return backward ? nl - 1 - it.s : it.s; };
   auto advance = [&](It it) {
      if (it.s >= nl) return it; // past the end: stays there
      const int L = level(it);
      it.p++;
      if (it.p * RP >= slp[L + 1] - slp[L]) { it.s++; it.p = 0; }
      return it;
   };
   struct RowA { int i, k0, k1; bool has; };
   struct RowB { int i, k0, k1; bool has; int c[NPF]; double a[NPF]; double d, rhs; };
   auto stage_a = [&](const It &it) {
      RowA r;
      r.has = false; r.i = 0; r.k0 = 0; r.k1 = 0;
      if (it.s < nl)
      {
         const int L = level(it), pos = slp[L] + it.p * RP + q;
         if (pos < slp[L + 1])
         {
            r.has = true;
            r.i   = perm[pos];
            r.k0  = rbeg[pos];
            r.k1  = rend[pos];
         }
      }
      return r;
   };
   auto stage_b = [&](const RowA &ra) {
      RowB r;
      r.i = ra.i; r.k0 = ra.k0; r.k1 = ra.k1; r.has = ra.has; r.d = 0.0; r.rhs = 0.0;
#pragma unroll
      for (int u = 0; u < NPF; u++) { r.c[u] = -1; r.a[u] = 0.0; }
      if (ra.has)
      {
#pragma unroll
         for (int u = 0; u < NPF; u++)
         {
            const int k = ra.k0 + lane + u * LPR;
            if (k < ra.k1) { r.c[u] = cj[k]; r.a[u] = v[k]; }
         }
         r.d   = dinv[ra.i];
         r.rhs = b[ra.i];
      }
      return r;
   };
   It   itC = {0, 0}, itB = advance(itC), itA = advance(itB);
   RowB cur = stage_b(stage_a(itC));
   RowA nxa = stage_a(itB);
   while (itC.s < nl)
   {
      const RowA nx2 = stage_a(itA); // two passes ahead
      const RowB nxb = stage_b(nxa); // one pass ahead
      if (cur.has)
      {
         double xs[NPF];
#pragma unroll
         for (int u = 0; u < NPF; u++) xs[u] = (cur.c[u] >= 0) ? x[cur.c[u]] : 0.0;
         double sum = 0.0;
#pragma unroll
         for (int u = 0; u < NPF; u++)
            if (cur.c[u] >= 0) sum += cur.a[u] * xs[u];
         for (int k = cur.k0 + lane + NPF * LPR; k < cur.k1; k += LPR) sum += v[k] * x[cj[k]]; // rows longer than the prefetch
#pragma unroll
         for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
         if (lane == 0) x[cur.i] += cur.d * (cur.rhs - sum);
      }
      // (the lanes of a row group always agree on cur.has: the shuffles above stay matched)
      const It nextC = itB;
      if (nextC.s != itC.s)
      { // the next pass belongs to another level: everything written so far must be visible
         __threadfence_block();
         __syncthreads();
      }
      itC = nextC;
      itB = itA;
      itA = advance(itA);
      cur = nxb;
      nxa = nx2;
   }
}

// Row-block form: workgroup q sweeps block q -- the levels of the block's own pattern one after the other, a barrier between two
// levels, the same three-stage pipeline as above -- while every other block is swept by its own workgroup at the same time.  A
// column inside the block is read from xout (the block's rows are copied there first and updated in place), a column outside from
// xin, which nobody writes during the sweep: the values the other blocks held when the sweep began (hypre's hybrid sweep with the
// blocks in the role of ranks).  zero_in: the input is the zero vector and is not read.
template <int LPR, int NPF>
__global__ __launch_bounds__(1024) void k_gs_blocks(int backward, int zero_in, int lds_levels, const int *__restrict__ part,
                                                    const int *__restrict__ blk_lvl_ptr, const int *__restrict__ blk_lvl,
                                                    const int *__restrict__ perm, const int *__restrict__ rbeg, const int *__restrict__ rend,
                                                    const int *__restrict__ cj, const double *__restrict__ v, const double *__restrict__ dinv,
                                                    const double *__restrict__ b, const double *xin, double *xout)
{
   extern __shared__ int slp_lds[];
   const int blk = blockIdx.x, lo = part[blk], hi = part[blk + 1];
   const int L0 = blk_lvl_ptr[blk], nl = blk_lvl_ptr[blk + 1] - L0;
   const int tid = threadIdx.x, lane = tid & (LPR - 1), q = tid / LPR;
   constexpr int RP = 1024 / LPR; // rows per pass
   const int *slp = blk_lvl + L0; // level offsets of this block: in LDS when they fit
   if (nl + 1 <= lds_levels)
   {
      for (int t = tid; t <= nl; t += 1024) slp_lds[t] = blk_lvl[L0 + t];
      slp = slp_lds;
   }
   if (zero_in) { for (int i = lo + tid; i < hi; i += 1024) xout[i] = 0.0; }
   else { for (int i = lo + tid; i < hi; i += 1024) xout[i] = xin[i]; }
   __threadfence_block();
   __syncthreads();
   struct It { int s, p; }; // level of the block in sweep order, pass inside the level
   auto level = [&](const It &it) { return backward ? nl - 1 - it.s : it.s; };
   auto advance = [&](It it) {
      if (it.s >= nl) return it; // past the end: stays there
      const int L = level(it);
      it.p++;
      if (it.p * RP >= slp[L + 1] - slp[L]) { it.s++; it.p = 0; }
      return it;
   };
   struct RowA { int i, k0, k1; bool has; };
   struct RowB { int i, k0, k1; bool has; int c[NPF]; double a[NPF]; double d, rhs; };
   auto stage_a = [&](const It &it) {
      RowA r;
      r.has = false; r.i = 0; r.k0 = 0; r.k1 = 0;
      if (it.s < nl)
      {
         const int L = level(it), pos = slp[L] + it.p * RP + q;
         if (pos < slp[L + 1])
         {
            r.has = true;
            r.i   = perm[pos];
            r.k0  = rbeg[pos];
            r.k1  = rend[pos];
         }
      }
      return r;
   };
   auto stage_b = [&](const RowA &ra) {
      RowB r;
      r.i = ra.i; r.k0 = ra.k0; r.k1 = ra.k1; r.has = ra.has; r.d = 0.0; r.rhs = 0.0;
#pragma unroll
      for (int u = 0; u < NPF; u++) { r.c[u] = -1; r.a[u] = 0.0; }
      if (ra.has)
      {
#pragma unroll
         for (int u = 0; u < NPF; u++)
         {
            const int k = ra.k0 + lane + u * LPR;
            if (k < ra.k1) { r.c[u] = cj[k]; r.a[u] = v[k]; }
         }
         r.d   = dinv[ra.i];
         r.rhs = b[ra.i];
      }
      return r;
   };
   auto value = [&](int c) { return (c >= lo && c < hi) ? xout[c] : (zero_in ? 0.0 : xin[c]); };
   It   itC = {0, 0}, itB = advance(itC), itA = advance(itB);
   RowB cur = stage_b(stage_a(itC));
   RowA nxa = stage_a(itB);
   while (itC.s < nl)
   {
      const RowA nx2 = stage_a(itA); // two passes ahead
      const RowB nxb = stage_b(nxa); // one pass ahead
      if (cur.has)
      {
         double xs[NPF];
#pragma unroll
         for (int u = 0; u < NPF; u++) xs[u] = (cur.c[u] >= 0) ? value(cur.c[u]) : 0.0;
         double sum = 0.0;
#pragma unroll
         for (int u = 0; u < NPF; u++)
            if (cur.c[u] >= 0) sum += cur.a[u] * xs[u];
         for (int k = cur.k0 + lane + NPF * LPR; k < cur.k1; k += LPR) sum += v[k] * value(cj[k]); // rows longer than the prefetch
#pragma unroll
         for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
         if (lane == 0) xout[cur.i] += cur.d * (cur.rhs - sum);
      }
      const It nextC = itB;
      if (nextC.s != itC.s)
      { // the next pass belongs to another level: everything written so far must be visible
         __threadfence_block();
         __syncthreads();
      }
      itC = nextC;
      itB = itA;
      itA = advance(itA);
      cur = nxb;
      nxa = nx2;
   }
}

// The same sweep on the sweep-order copy (GsPlan::s_*): before it, every unknown's iterate, right-hand side and divisor are brought
// into sweep order by a kernel of the whole chip; after it the iterate goes back.  Position q of the copy is the row perm[q].
__global__ __launch_bounds__(256) void k_gs_to_sweep_order(int n, int zero_in, const int *__restrict__ perm, const double *__restrict__ xin,
                                                           const double *__restrict__ b, const double *__restrict__ dinv,
                                                           double *__restrict__ sx, double *__restrict__ sb, double *__restrict__ sd,
                                                           double *__restrict__ sx0 = nullptr)
{
   const int q = xcd_chunk_block() * 256 + threadIdx.x;
   if (q >= n) return;
   const int i = perm[q];
   if (sx || sx0)
   { // (the barrier-free kernel reads the sweep-start copy alone -- not even that for a zero input, which it is told of --
     //  and writes its result in the caller's numbering: neither pointer, or sx0 only)
      const double x = zero_in ? 0.0 : xin[i];
      if (sx) sx[q] = x;
      if (sx0) sx0[q] = x; // (the ring kernel reads the other blocks' columns from this copy, which the sweep leaves alone)
   }
   if (b) sb[q] = b[i];       // (nullptr: the sweep-order copy is current -- GsPlan::sb_src / sd_src)
   if (dinv) sd[q] = dinv[i];
}
__global__ __launch_bounds__(256) void k_gs_from_sweep_order(int n, const int *__restrict__ perm, const double *__restrict__ sx, double *__restrict__ xout)
{
   const int q = xcd_chunk_block() * 256 + threadIdx.x;
   if (q < n) xout[perm[q]] = sx[q];
}
// LPR lanes per row, one 4-entry chunk per lane and pass (rows longer than 4 * LPR entries: further chunks in a loop).
// The loop body is STRAIGHT-LINE code: every load of a pass is issued unconditionally at a clamped, always valid address and its
// result masked afterwards.  The first form of this kernel wrapped each load in the branch of its condition, and the level offsets
// sat behind a generic pointer (LDS or memory): the compiler then cannot count the loads in flight and drains ALL of them
// (s_waitcnt vmcnt(0), flat loads on both counters) at every use -- four to six full memory round trips per pass where one is
// needed (disassembly in profiles/r04_gs_blocks.md).  Here the level offsets are always in LDS, the gathers of a pass go out first,
// the requests for the next two passes behind them, and the two software-pipeline stages alternate between two sets of registers
// (no copies at the end of a pass, which would wait for the loads they copy).
template <int LPR, int NT>
__global__ __launch_bounds__(NT) void k_gs_blocks_sorted(int backward, int zero_in, const int *__restrict__ blk_lvl_ptr,
                                                           const int *__restrict__ blk_lvl, const int *__restrict__ srp4,
                                                           const int4 *__restrict__ scj4, const double2 *__restrict__ sv2,
                                                           const double *__restrict__ sd, const double *__restrict__ sb,
                                                           const double *__restrict__ saii, const double *__restrict__ sx0, double *sx,
                                                           unsigned long long *diag)
{
   extern __shared__ int slp[]; // level offsets of this block (nl + 1 of them)
   unsigned long long tG = 0, tF = 0, tB = 0, tN = 0, t0 = 0, t1 = 0, tU = 0, tI = 0, tE = 0; // diag: shader-clock sums per phase, passes
   const int blk = blockIdx.x;
   const int L0 = blk_lvl_ptr[blk], nl = blk_lvl_ptr[blk + 1] - L0;
   const int tid = threadIdx.x, lane = tid & (LPR - 1), q = tid / LPR;
   constexpr int RP = NT / LPR; // rows per pass
   for (int t = tid; t <= nl; t += NT) slp[t] = blk_lvl[L0 + t];
   __syncthreads();
   if (nl == 0) return;
   struct It { int s, p; };
   auto level = [&](const It &it) { const int sc = min(it.s, nl - 1); return backward ? nl - 1 - sc : sc; }; // (clamped past the end)
   auto advance = [&](It it) {
      const int L = level(it);
      it.p++;
      if (it.s < nl && it.p * RP >= slp[L + 1] - slp[L]) { it.s++; it.p = 0; }
      return it;
   };
   struct RowA { int pos, c0, c1; bool has; };
   struct RowB { int pos, c0, c1; bool has, mine; int4 c; double2 a01, a23; double d, rhs, aii; };
   auto stage_a = [&](const It &it) {
      RowA r;
      const int L = level(it), pos = slp[L] + it.p * RP + q;
      r.has = (it.s < nl) && (pos < slp[L + 1]);
      r.pos = r.has ? pos : slp[L]; // a valid row either way: the loads below are unconditional
      r.c0  = srp4[r.pos];
      r.c1  = srp4[r.pos + 1];
      return r;
   };
   auto stage_b = [&](const RowA &ra) {
      RowB r;
      r.pos = ra.pos; r.c0 = ra.c0; r.c1 = ra.c1; r.has = ra.has;
      const int ch = ra.c0 + lane;
      r.mine       = ra.has && (ch < ra.c1);
      const int cc = r.mine ? ch : ra.c0; // (the copy ends with a spare chunk: c0 of an empty last row is still readable)
      r.c   = scj4[cc];
      r.a01 = sv2[2 * cc];
      r.a23 = sv2[2 * cc + 1];
      r.d   = sd[ra.pos];
      r.rhs = sb[ra.pos];
      r.aii = saii[ra.pos];
      return r;
   };
   // value of column c: inside the block sx[position], outside it the other blocks' value at the sweep's start, sx0[~c] (zeros from a
   // zero guess) -- ONE load
   auto address = [&](int c, int) -> const double * { return (c >= 0) ? sx + c : sx0 + ~c; };
   auto masked  = [&](int, double v) { return v; };
   auto pass = [&](const RowB &cur, const RowA &nxa, RowB &nxb, RowA &nx2, const It &itA) {
      // the gathers of this pass go out FIRST: loads return in issue order, so requests made before them would have to land before them
      const double x0 = *address(cur.c.x, cur.pos), x1 = *address(cur.c.y, cur.pos), x2 = *address(cur.c.z, cur.pos),
                   x3 = *address(cur.c.w, cur.pos);
      if (diag) { t0 = __builtin_amdgcn_s_memtime(); if (tE) tU += t0 - tE; }
      nx2 = stage_a(itA); // two passes ahead
      nxb = stage_b(nxa); // one pass ahead
      if (diag) { const unsigned long long ti = __builtin_amdgcn_s_memtime(); tI += ti - t0; }
      double sum = 0.0;
      sum += cur.a01.x * masked(cur.c.x, x0);
      sum += cur.a01.y * masked(cur.c.y, x1);
      sum += cur.a23.x * masked(cur.c.z, x2);
      sum += cur.a23.y * masked(cur.c.w, x3);
      if (!cur.mine) sum = 0.0;
      if (cur.has)
         for (int ch = cur.c0 + lane + LPR; ch < cur.c1; ch += LPR)
         { // rows longer than 4 * LPR entries
            const int4    c = scj4[ch];
            const double2 a = sv2[2 * ch], b2 = sv2[2 * ch + 1];
            const double  y0 = *address(c.x, cur.pos), y1 = *address(c.y, cur.pos), y2 = *address(c.z, cur.pos), y3 = *address(c.w, cur.pos);
            sum += a.x * masked(c.x, y0);
            sum += a.y * masked(c.y, y1);
            sum += b2.x * masked(c.z, y2);
            sum += b2.y * masked(c.w, y3);
         }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (lane == 0 && cur.has)
      { // (the diagonal entry is kept apart: s_aii)
         const double xo = sx[cur.pos];
         sx[cur.pos]     = xo + cur.d * (cur.rhs - (sum + cur.aii * xo));
      }
      if (diag)
      {
         t1 = __builtin_amdgcn_s_memtime();
         tG += t1 - t0 + (unsigned long long)(sum == 12345.678); // (the sum is needed here: the gathers have landed)
         tN++;
      }
   };
   It   itC = {0, 0}, itB = advance(itC), itA = advance(itB);
   RowB B0 = stage_b(stage_a(itC)), B1;
   RowA A1 = stage_a(itB), A0;
   auto step = [&]() { // after a pass: a barrier when the next one belongs to another level, then the iterators move on
      if (itB.s != itC.s)
      {
         __threadfence_block();
         if (diag) { t0 = __builtin_amdgcn_s_memtime(); tF += t0 - t1; }
         __syncthreads();
         if (diag) { t1 = __builtin_amdgcn_s_memtime(); tB += t1 - t0; }
      }
      if (diag) tE = t1;
      itC = itB;
      itB = itA;
      itA = advance(itA);
   };
   while (true)
   {
      pass(B0, A1, B1, A0, itA);
      step();
      if (itC.s >= nl) break;
      pass(B1, A0, B0, A1, itA);
      step();
      if (itC.s >= nl) break;
   }
   if (diag && blk == 0 && (tid & 63) == 0)
   {
      unsigned long long *o = diag + 4 * (tid >> 6);
      o[0] = tG; o[1] = tF; o[2] = tB; o[3] = tN; diag[64 + 2 * (tid >> 6)] = tU; diag[65 + 2 * (tid >> 6)] = tI;
   }
}

// The row-block sweep with the iterate of the level just swept in LDS.  What bounds the kernels above is not bandwidth but the chain
// of dependent steps per level -- barrier, gather of the values the previous level wrote (a trip to L2 behind the streaming loads of
// sixteen wavefronts: 1 700-2 000 cycles measured with the shader clock), arithmetic, store, fence -- and, once those are shortened,
// the number of INSTRUCTIONS a wavefront issues per pass (450 in the first form of this kernel: iterators over LDS tables, address
// selects for three kinds of columns; profiles/r04_gs_blocks.md).  Here a row goes through three stages in three consecutive passes:
//   B  its chunk of the level-wise copy (columns, values), divisor, right-hand side, own old value      (streams, two passes ahead)
//   X  the values of its columns that are NOT in the level being swept right now -- older levels' stores have landed before this
//      pass began, later levels still hold the values of the sweep's start, other blocks' columns name the sweep-start copy: all
//      safe to read one pass early
//   C  the values of the columns in the level just swept, from the LDS ring where that level's rows left them; sum; update; the new
//      value goes to the ring and to memory
// so between two barriers the dependent work is an LDS read, the arithmetic and the stores.  B runs three passes ahead of C and X two:
// a value requested in one pass is not needed before the pass after the next (a request made in the pass before its use waited out
// its whole latency at the top of that pass: 33.6 ms per 128^3 solve either way).  The ring therefore holds the TWO levels swept
// last, and the plan keeps three consecutive levels of a block within its kGsRing slots.  Four register sets rotate (no copies).
// The control flow is a LIST OF PASSES made at plan time (GsPlan::r_pass): one scalar load per pass.
struct GsPass { int pos0, rows, chunk0, w, ulo, uhi, last, pad; };
template <int LPR, int NT>
__global__ __launch_bounds__(NT) void k_gs_blocks_ring(const int *__restrict__ pass_ptr, const GsPass *__restrict__ passes,
                                                       const int4 *__restrict__ rcj4, const double2 *__restrict__ rv2,
                                                       const double *__restrict__ sd, const double *__restrict__ sb,
                                                       const double *__restrict__ sx0, double *sx, unsigned long long *diag)
{
   __shared__ double ring[kGsRing];
   unsigned long long tc = 0, ti = 0, tw = 0, tb = 0, tn = 0, ta = 0, tz = 0;
   const int      blk = blockIdx.x, np = pass_ptr[blk + 1] - pass_ptr[blk] - 3; // (three spare passes close every block's list)
   const GsPass  *pl  = passes + pass_ptr[blk];
   const int      tid = threadIdx.x, lane = tid & (LPR - 1), q = tid / LPR;
   if (np <= 0) return;
   struct Row {
      int     pos, chunk0, w, ulo, uhi;
      bool    has, mine, r0, r1, r2, r3; // (r_u: column u is read from the ring)
      int4    c;
      double2 a01, a23;
      double  d, rhs, xown, x0, x1, x2, x3;
   };
   auto stage_b = [&](Row &r, const GsPass &p) {
      r.has       = q < p.rows;
      const int j = r.has ? q : 0;
      r.pos       = p.pos0 + j;
      r.w         = p.w;
      r.ulo       = p.ulo;
      r.uhi       = p.uhi;
      r.chunk0    = p.chunk0 + j * p.w;
      r.mine      = r.has && (lane < p.w);
      const int cc = r.mine ? r.chunk0 + lane : p.chunk0; // (always readable: the copy ends with a spare chunk)
      r.c    = rcj4[cc];
      r.a01  = rv2[2 * cc];
      r.a23  = rv2[2 * cc + 1];
      r.d    = sd[r.pos];
      r.rhs  = sb[r.pos];
      r.xown = sx[r.pos];
   };
   // where column c is read early: its own position (sweep-order iterate), ~position of another block's column (sweep-start copy), or
   // -- when its value must wait for the ring -- the row's own slot, which is always valid
   auto early = [&](int c, int self, bool in_ring) -> const double * { return (c >= 0) ? sx + (in_ring ? self : c) : sx0 + ~c; };
   auto stage_x = [&](Row &r) {
      r.r0 = (r.c.x >= r.ulo) && (r.c.x < r.uhi);
      r.r1 = (r.c.y >= r.ulo) && (r.c.y < r.uhi);
      r.r2 = (r.c.z >= r.ulo) && (r.c.z < r.uhi);
      r.r3 = (r.c.w >= r.ulo) && (r.c.w < r.uhi);
      r.x0 = *early(r.c.x, r.pos, r.r0);
      r.x1 = *early(r.c.y, r.pos, r.r1);
      r.x2 = *early(r.c.z, r.pos, r.r2);
      r.x3 = *early(r.c.w, r.pos, r.r3);
   };
   auto stage_c = [&](const Row &r) {
      const double y0 = ring[r.c.x & (kGsRing - 1)], y1 = ring[r.c.y & (kGsRing - 1)], y2 = ring[r.c.z & (kGsRing - 1)],
                   y3 = ring[r.c.w & (kGsRing - 1)];
      double sum = 0.0;
      sum += r.a01.x * (r.r0 ? y0 : r.x0);
      sum += r.a01.y * (r.r1 ? y1 : r.x1);
      sum += r.a23.x * (r.r2 ? y2 : r.x2);
      sum += r.a23.y * (r.r3 ? y3 : r.x3);
      if (!r.mine) sum = 0.0;
      if (r.has)
         for (int ch = lane + LPR; ch < r.w; ch += LPR)
         { // rows of a level wider than 4 * LPR entries: the remaining chunks, read now
            const int4    c = rcj4[r.chunk0 + ch];
            const double2 a = rv2[2 * (r.chunk0 + ch)], b2 = rv2[2 * (r.chunk0 + ch) + 1];
            const bool    i0 = (c.x >= r.ulo) && (c.x < r.uhi), i1 = (c.y >= r.ulo) && (c.y < r.uhi), i2 = (c.z >= r.ulo) && (c.z < r.uhi),
                       i3 = (c.w >= r.ulo) && (c.w < r.uhi);
            const double  e0 = *early(c.x, r.pos, i0), e1 = *early(c.y, r.pos, i1), e2 = *early(c.z, r.pos, i2), e3 = *early(c.w, r.pos, i3);
            sum += a.x * (i0 ? ring[c.x & (kGsRing - 1)] : e0);
            sum += a.y * (i1 ? ring[c.y & (kGsRing - 1)] : e1);
            sum += b2.x * (i2 ? ring[c.z & (kGsRing - 1)] : e2);
            sum += b2.y * (i3 ? ring[c.w & (kGsRing - 1)] : e3);
         }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      if (lane == 0 && r.has)
      {
         const double nv             = r.xown + r.d * (r.rhs - sum);
         ring[r.pos & (kGsRing - 1)] = nv;
         sx[r.pos]                   = nv;
      }
   };
   Row S0, S1, S2, S3;
   stage_b(S0, pl[0]);
   stage_b(S1, pl[1]);
   stage_b(S2, pl[2]);
   stage_x(S0); // (the first pass's ring range is empty, the second's names the first level: its values come from the ring later)
   stage_x(S1);
   int  t = 0;
   bool last = pl[0].last != 0;
   // one pass: C of this pass's rows first (LDS reads, arithmetic, the stores), then X of the rows two passes on and B of the rows three
   // passes on go out -- ten loads per lane, unconditional, in program order behind the store.  Before the barrier a wavefront waits
   // until at most those ten are outstanding (memory operations retire in issue order: the store has then landed) and for its LDS
   // writes; the loads stay in flight across the barrier.  (__syncthreads' own fence would drain them.)
   auto pass = [&](const Row &c, Row &x, Row &b) {
      const GsPass pb    = pl[t + 3];
      const bool   lnext = pl[t + 1].last != 0;
      if (diag) { ta = __builtin_amdgcn_s_memtime(); if (tz) tn += ta - tz; }
      stage_c(c);
      asm volatile("" ::: "memory"); // (no memory operation of X / B may be scheduled above the store)
      if (diag) { const unsigned long long u = __builtin_amdgcn_s_memtime(); tc += u - ta; ta = u; }
      stage_x(x);
      stage_b(b, pb);
      if (diag) { const unsigned long long u = __builtin_amdgcn_s_memtime(); ti += u - ta; ta = u; }
      if (last)
      { // the next pass belongs to another level: this level's stores and ring writes must have landed
         asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
         if (diag) { const unsigned long long u = __builtin_amdgcn_s_memtime(); tw += u - ta; ta = u; }
         __builtin_amdgcn_s_barrier();
         asm volatile("" ::: "memory");
         if (diag) { const unsigned long long u = __builtin_amdgcn_s_memtime(); tb += u - ta; ta = u; }
      }
      tz = ta;
      last = lnext;
      t++;
   };
   while (true)
   {
      pass(S0, S2, S3);
      if (t >= np) break;
      pass(S1, S3, S0);
      if (t >= np) break;
      pass(S2, S0, S1);
      if (t >= np) break;
      pass(S3, S1, S2);
      if (t >= np) break;
   }
   if (diag && blk == 0 && (tid & 63) == 0)
   {
      unsigned long long *o = diag + 8 * (tid >> 6);
      o[0] = tc; o[1] = ti; o[2] = tw; o[3] = tb; o[4] = tn; o[5] = (unsigned long long)t;
   }
}

static unsigned long long *gs_diag_buffer();
static void                gs_ring_diag_report(int lpr, int nt, int n);
template <int LPR, int NT>
static void gs_blocks_ring_t(const DCsr &A, const GsPlan &p, const double *dinv, const double *b, const double *xin, double *xout, bool forward,
                             bool zero_in)
{
   const int n = A.nrows;
   k_gs_to_sweep_order<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, zero_in ? 1 : 0, p.perm.data(), xin, p.to_b, p.to_d, p.s_x.data(), p.s_b.data(), p.s_d.data(),
                                                            p.s_x0.data());
   k_gs_blocks_ring<LPR, NT><<<p.nblk, NT, 0, STREAM>>>(p.r_pass_ptr.data(), (const GsPass *)p.r_pass[forward ? 0 : 1].data(),
                                                      (const int4 *)p.r_col.data(), (const double2 *)p.r_val.data(), p.s_d.data(), p.s_b.data(),
                                                      p.s_x0.data(), p.s_x.data(), gs_diag_buffer());
   gs_ring_diag_report(LPR, NT, n);
   k_gs_from_sweep_order<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, p.perm.data(), p.s_x.data(), xout);
}

// (the shader-clock instrumentation of the ring / sorted kernels -- their `diag` argument -- is switched off in product builds: the
//  measurements it made are in profiles/r04_gs_blocks.md)
static unsigned long long *gs_diag_buffer() { return nullptr; }
static void gs_ring_diag_report(int, int, int) {}
static void gs_diag_report(int, int) {}

template <int LPR, int NT>
static void gs_blocks_sorted_t(const DCsr &A, const GsPlan &p, const double *dinv, const double *b, const double *xin, double *xout, bool forward,
                               bool zero_in)
{
   const int    n   = A.nrows;
   const size_t lds = sizeof(int) * (size_t)(p.blk_max_levels + 1);
   k_gs_to_sweep_order<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, zero_in ? 1 : 0, p.perm.data(), xin, p.to_b, p.to_d, p.s_x.data(), p.s_b.data(), p.s_d.data(),
                                                          p.s_x0.data());
   k_gs_blocks_sorted<LPR, NT><<<p.nblk, NT, lds, STREAM>>>(forward ? 0 : 1, zero_in ? 1 : 0, p.blk_lvl_ptr.data(), p.blk_lvl.data(),
                                                         p.s_rowptr.data(), (const int4 *)p.s_col.data(), (const double2 *)p.s_val.data(),
                                                         p.s_d.data(), p.s_b.data(), p.s_aii.data(), p.s_x0.data(), p.s_x.data(), gs_diag_buffer());
   gs_diag_report(LPR, n);
   k_gs_from_sweep_order<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, p.perm.data(), p.s_x.data(), xout);
}

// ---- the row-block sweep WITHOUT level barriers ("sync-free": every row waits for its own dependencies) ---------------------------------
// The kernels above pay a workgroup barrier and a full software-pipeline stage per dependency level (~2 800 cycles measured).  Here the
// rows of a block are dealt round-robin, in sweep order, to the G = NT / LPR lane groups of the block's workgroup; a group finishes
// sweep index t (row of position lo + t, or hi - 1 - t backwards), leaves the new value in an LDS ring (slot t mod RING) and then
// publishes t in done[group] (LDS; the two writes of one lane stay in order), and a row whose column is an EARLIER sweep index t' polls
// done[t' mod G] >= t' before it takes the value from the ring.  Columns later in the sweep, the row's own value and other blocks'
// columns all come from the sweep-start copy (s_x0): none depends on the sweep, all are requested a round before the wait.
// What a round costs turned out to be the instructions the workgroup's eight wavefronts issue for it (two wavefronts to a SIMD; about
// 1.06 us), not the LDS round trips, the memory latency or the bandwidth: profiles/r04_gs_blocks.md.  A group may run at most
// RING - W - G sweep indices ahead of the slowest wavefront (wdone[]), so that a ring slot is never overwritten while a row W = RING / 4
// indices back may still read it; the plan guarantees that no dependency reaches further back than W, or gives every row of the block a
// slot of its own (blocks of up to 16 384 rows: no guard at all), else this kernel is not used.  A lane takes MAXC 4-entry chunks of its
// row (one; two for rows that would otherwise need 32 or 64 lanes): LPR x MAXC >= chunks of the longest row, the diagonal entry is kept
// apart (s_aii).  Progress: the smallest unfinished sweep index never waits for the guard and all its dependencies are smaller, so some
// row can always finish; a spin limit turns a protocol error into an error flag (gs_free_check) instead of a hang.

// sum / conjunction over the LPR lanes of a row with every lane receiving the result: data-parallel-primitive moves inside a row of
// sixteen lanes (quad permutes, half-row and row mirrors: no LDS traffic, unlike ds_bpermute), shuffles beyond.  The pairing differs from
// the xor butterfly only in which lane adds which partner: a + b and b + a, the same bits.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double x)
{
   const int lo = dpp_i<CTRL>(__double2loint(x)), hi = dpp_i<CTRL>(__double2hiint(x));
   return __hiloint2double(hi, lo);
}
template <int LPR>
__device__ __forceinline__ double group_sum(double x)
{
   if (LPR >= 2) x += dpp_d<0xB1>(x);  // quad_perm [1,0,3,2]
   if (LPR >= 4) x += dpp_d<0x4E>(x);  // quad_perm [2,3,0,1]
   if (LPR >= 8) x += dpp_d<0x141>(x); // row_half_mirror
   if (LPR >= 16) x += dpp_d<0x140>(x); // row_mirror
   if (LPR >= 32) x += __shfl_xor(x, 16);
   if (LPR >= 64) x += __shfl_xor(x, 32);
   return x;
}
template <int LPR>
__device__ __forceinline__ int group_and(int x)
{
   if (LPR >= 2) x &= dpp_i<0xB1>(x);
   if (LPR >= 4) x &= dpp_i<0x4E>(x);
   if (LPR >= 8) x &= dpp_i<0x141>(x);
   if (LPR >= 16) x &= dpp_i<0x140>(x);
   if (LPR >= 32) x &= __shfl_xor(x, 16);
   if (LPR >= 64) x &= __shfl_xor(x, 32);
   return x;
}
template <int LPR, int MAXC, int NT, bool BACKWARD, bool LONG>
__global__ __launch_bounds__(NT) void k_gs_blocks_free(int ring_mask, int zero_in, const int *__restrict__ part, const int *__restrict__ srp4,
                                                       const int4 *__restrict__ scj4, const double2 *__restrict__ sv2,
                                                       const double *__restrict__ sd, const double *__restrict__ sb,
                                                       const double *__restrict__ saii, const double *__restrict__ sx0, double *sx, int *err)
{
   constexpr int G = NT / LPR, NW = NT / 64, GPW = 64 / LPR, NE = 4 * MAXC; // groups, wavefronts, groups per wavefront, entries per lane
   extern __shared__ long long free_ring[];   // ring_mask + 1 values (the bits of a double each)
   __shared__ int              done_s[G];     // last sweep index every group has finished
   __shared__ int              wdone_s[NW];   // last sweep index of the last round every wavefront has finished
   // (relaxed atomic accesses of the NAMED shared arrays: they stay LDS instructions and are re-read in every turn of a waiting loop;
   //  a volatile pointer to them becomes a generic pointer -- flat loads with system scope and a wait for every counter after each)
#define RING_LD(i) __longlong_as_double(__atomic_load_n(&free_ring[i], __ATOMIC_RELAXED))
#define RING_ST(i, x) __atomic_store_n(&free_ring[i], __double_as_longlong(x), __ATOMIC_RELAXED)
#define DONE_LD(i) __atomic_load_n(&done_s[i], __ATOMIC_RELAXED)
#define DONE_ST(i, x) __atomic_store_n(&done_s[i], x, __ATOMIC_RELAXED)
#define WDONE_LD(i) __atomic_load_n(&wdone_s[i], __ATOMIC_RELAXED)
#define WDONE_ST(i, x) __atomic_store_n(&wdone_s[i], x, __ATOMIC_RELAXED)
   constexpr bool backward = BACKWARD;
   const int blk = blockIdx.x, lo = part[blk], hi = part[blk + 1], nb = hi - lo;
   const int tid = threadIdx.x, lane = tid & (LPR - 1), g = tid / LPR, wave = tid >> 6;
   if (lane == 0) DONE_ST(g, -1);
   if ((tid & 63) == 0) WDONE_ST(wave, -1);
   __syncthreads();
   const int  rounds = (nb + G - 1) / G, W = (ring_mask + 1) >> 2;
   const bool whole  = (ring_mask + 1) >= nb; // every value of the block has a slot of its own: nothing is ever overwritten, no guard
   int        KG     = 8;
   while (KG > 1 && KG * G > (ring_mask + 1) - W - G - 1) KG >>= 1;
   if (!whole && KG * G > (ring_mask + 1) - W - G - 1)
   { // the plan broke the guard's invariant (gs_free_plan sizes the ring for it): say so instead of waiting for a progress nobody can make
      if (tid == 0) atomicMax(err, 3);
      return;
   }
   // A row goes through four stages in four consecutive rounds, so that nothing the sweep does not produce is waited for when the row's
   // turn comes: A its chunk range (three rounds ahead), B its chunks, divisor, right-hand side and own old value (two ahead), X the
   // values that do not depend on the sweep -- other blocks' columns, columns later in the sweep (one ahead) --, C the wait for its
   // dependencies, the sum and the update.  Four NAMED register sets per stage rotate and the loop is unrolled by four: plain values
   // the compiler keeps in registers (arrays of these structs went to scratch memory), never a copy of a value still in flight.
   // (Deeper pipelines -- leads of 5 / 3 / 1 and 6 / 4 / 2 rounds -- measured the same: a round costs the instructions its eight
   //  wavefronts issue, two to a SIMD, not a memory latency.)
   struct RowA { int t, p, c0, c1; bool has; };
   // (c0, c1: the row's chunk range; a row of more than LPR * MAXC chunks -- the plan sizes the lanes for all but a few per cent of the
   //  rows -- has its further chunks read in stage C itself: `long_row`)
   struct RowB { int t, p, c0, c1; bool has, mine[MAXC]; int4 c[MAXC]; double2 a01[MAXC], a23[MAXC]; double d, rhs, own, aii; };
   struct RowX { int t, p, c0, c1; bool has, mine[MAXC]; double a[NE], v[NE], d, rhs, own, aii; int dep[NE]; };
   auto stage_a = [&](int r) {
      RowA ra;
      ra.t   = r * G + g;
      ra.has = ra.t < nb;
      ra.p   = ra.has ? (backward ? hi - 1 - ra.t : lo + ra.t) : lo;
      ra.c0  = srp4[ra.p];
      ra.c1  = srp4[ra.p + 1];
      return ra;
   };
   auto stage_b = [&](const RowA &ra) {
      RowB rb;
      rb.t = ra.t; rb.p = ra.p; rb.has = ra.has; rb.c0 = ra.c0; rb.c1 = ra.has ? ra.c1 : ra.c0;
#pragma unroll
      for (int m = 0; m < MAXC; m++)
      {
         const int  ch   = ra.c0 + lane + m * LPR;
         const bool mine = ra.has && ch < ra.c1;
         const int  cc   = mine ? ch : ra.c0; // (always a readable chunk: the copy ends with a spare one)
         rb.c[m]    = scj4[cc];
         rb.a01[m]  = sv2[2 * cc];
         rb.a23[m]  = sv2[2 * cc + 1];
         rb.mine[m] = mine; // (a lane without a chunk of its own carries the row's first one: real dependencies of the row, a sum that is dropped)
      }
      rb.d   = sd[ra.p];
      rb.rhs = sb[ra.p];
      rb.own = zero_in ? 0.0 : sx0[ra.p]; // (zero_in, the same for every lane of the grid: a sweep from the zero guess reads no iterate)
      rb.aii = saii[ra.p];
      return rb;
   };
   auto stage_x = [&](const RowB &rb) {
      RowX rx;
      rx.t = rb.t; rx.p = rb.p; rx.has = rb.has; rx.c0 = rb.c0; rx.c1 = rb.c1; rx.d = rb.d; rx.rhs = rb.rhs; rx.own = rb.own; rx.aii = rb.aii;
#pragma unroll
      for (int m = 0; m < MAXC; m++)
      {
         rx.mine[m]           = rb.mine[m];
         const int    cols[4] = {rb.c[m].x, rb.c[m].y, rb.c[m].z, rb.c[m].w};
         const double as[4]   = {rb.a01[m].x, rb.a01[m].y, rb.a23[m].x, rb.a23[m].y};
#pragma unroll
         for (int e = 0; e < 4; e++)
         { // every load unconditional, at an always valid address; what it is worth is decided afterwards
            const int  col   = cols[e];
            const int  pos   = col ^ (col >> 31); // (another block's column is stored as ~position: the sweep-start copy serves both)
            const bool isdep = col >= 0 && (backward ? col > rb.p : col < rb.p);
            const double val = zero_in ? 0.0 : sx0[pos];
            rx.a[4 * m + e]   = as[e];
            rx.dep[4 * m + e] = isdep ? (backward ? hi - 1 - col : col - lo) : -1;
            rx.v[4 * m + e]   = isdep ? 0.0 : val;
         }
      }
      return rx;
   };
   int  bad = 0; // (a spin limit was hit: reported once, after the sweep -- no memory operation inside the waiting loops)
   auto stage_c = [&](RowX &rx, int r) {
      // guard of the ring (looked at every KG-th round, for the KG rounds ahead; KG G <= RING - W - G - 1, or round 0 could never start):
      // nobody more than RING - W - G sweep indices behind
      const int need = (r + KG) * G - (ring_mask + 1) + W + G;
      if (!whole && (r & (KG - 1)) == 0 && need >= 0)
      {
         int spins = 0;
         while (true)
         {
            int m = WDONE_LD(tid & (NW - 1));
#pragma unroll
            for (int o = NW / 2; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
            if (m >= need) break;
            if (++spins > (1 << 22)) { bad = 2; break; }
         }
      }
      double     xn       = 0.0;
      // (LONG: the plan left rows beyond the lanes' capacity -- few by its rule; without them the kernel is built without this path, which
      //  costs 34 registers: 154 against 120 with one chunk per lane, one workgroup to a CU instead of two)
      const bool long_row = LONG && rx.c1 - rx.c0 > LPR * MAXC;
      auto look = [&]() { // are this lane's dependencies done?  (every read unconditional: dep = -1 names the last slot, then ignored)
         int ok = 1;
#pragma unroll
         for (int e = 0; e < NE; e++)
         {
            const int dn = DONE_LD(rx.dep[e] & (G - 1));
            ok &= (rx.dep[e] < 0) | (dn >= rx.dep[e]);
         }
         if (long_row)
            for (int ch = rx.c0 + LPR * MAXC + lane; ch < rx.c1; ch += LPR)
            {
               const int4 c       = scj4[ch];
               const int  cols[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
               for (int e = 0; e < 4; e++)
               {
                  const int  col   = cols[e];
                  const bool isdep = col >= 0 && (backward ? col > rx.p : col < rx.p);
                  const int  dep   = isdep ? (backward ? hi - 1 - col : col - lo) : -1;
                  const int  dn    = DONE_LD(dep & (G - 1));
                  ok &= (dep < 0) | (dn >= dep);
               }
            }
         return group_and<LPR>(ok);
      };
      auto finish = [&]() { // the row's sum with the ring's values, the update, the new value into the ring and then the flag
         asm volatile("" ::: "memory");
         double rv[NE];
#pragma unroll
         for (int e = 0; e < NE; e++) rv[e] = RING_LD(rx.dep[e] & ring_mask);
         double sum = 0.0;
#pragma unroll
         for (int m = 0; m < MAXC; m++)
         {
            double sm = 0.0;
#pragma unroll
            for (int e = 4 * m; e < 4 * m + 4; e++) sm += rx.a[e] * (rx.dep[e] >= 0 ? rv[e] : rx.v[e]);
            sum += rx.mine[m] ? sm : 0.0;
         }
         if (long_row)
            for (int ch = rx.c0 + LPR * MAXC + lane; ch < rx.c1; ch += LPR)
            {
               const int4    c       = scj4[ch];
               const double2 a01 = sv2[2 * ch], a23 = sv2[2 * ch + 1];
               const int     cols[4] = {c.x, c.y, c.z, c.w};
               const double  as[4]   = {a01.x, a01.y, a23.x, a23.y};
               double        sm      = 0.0;
#pragma unroll
               for (int e = 0; e < 4; e++)
               {
                  const int  col   = cols[e];
                  const int  pos   = col ^ (col >> 31);
                  const bool isdep = col >= 0 && (backward ? col > rx.p : col < rx.p);
                  const int  dep   = isdep ? (backward ? hi - 1 - col : col - lo) : -1;
                  const double val = isdep ? RING_LD(dep & ring_mask) : (zero_in ? 0.0 : sx0[pos]);
                  sm += as[e] * val;
               }
               sum += sm;
            }
         sum = group_sum<LPR>(sum);
         xn  = rx.own + rx.d * (rx.rhs - (sum + rx.aii * rx.own)); // (the diagonal entry is kept apart: s_aii)
         if (lane == 0 && rx.has)
         {
            RING_ST(rx.t & ring_mask, xn);
            asm volatile("" ::: "memory");
            DONE_ST(g, rx.t);
         }
      };
      // the common case first, without a diverging branch: every row of this wavefront's round finds its dependencies done at the first
      // look (a round is a level's worth of rows; the level before it was finished a round ago)
      if (__builtin_amdgcn_ballot_w64(rx.has && !look()) == 0ull) finish();
      else
      { // rows of this round that wait for each other (a round across a level's end) or for a slower wavefront
         bool fin   = !rx.has;
         int  spins = 0;
         while (true)
         {
            if (!fin && look())
            {
               finish();
               fin = true;
            }
            if (__builtin_amdgcn_ballot_w64(!fin) == 0ull) break;
            if (++spins > (1 << 22)) { bad = 1; break; }
         }
      }
      if (lane == 0 && rx.has) sx[rx.p] = xn; // (after the loop: a memory operation inside it would make the compiler wait for every prefetch)
      if ((tid & 63) == 0) WDONE_ST(wave, r * G + wave * GPW + GPW - 1);
   };
   RowA A0 = stage_a(0), A1 = stage_a(1), A2 = stage_a(2), A3;
   RowB B0 = stage_b(A0), B1 = stage_b(A1), B2, B3;
   RowX X0 = stage_x(B0), X1, X2, X3;
   for (int r = 0; r < rounds && !bad; r += 4)
   {
      A3 = stage_a(r + 3); B2 = stage_b(A2); X1 = stage_x(B1); stage_c(X0, r);
      if (r + 1 >= rounds) break;
      A0 = stage_a(r + 4); B3 = stage_b(A3); X2 = stage_x(B2); stage_c(X1, r + 1);
      if (r + 2 >= rounds) break;
      A1 = stage_a(r + 5); B0 = stage_b(A0); X3 = stage_x(B3); stage_c(X2, r + 2);
      if (r + 3 >= rounds) break;
      A2 = stage_a(r + 6); B1 = stage_b(A1); X0 = stage_x(B0); stage_c(X3, r + 3);
   }
   if (bad) *err = bad;
}
#undef RING_LD
#undef RING_ST
#undef DONE_LD
#undef DONE_ST
#undef WDONE_LD
#undef WDONE_ST

// longest distance, in sweep positions, between a row and an in-block column of it (the reach of a dependency in either direction)
__global__ __launch_bounds__(256) void k_gs_reach(int n, const int *__restrict__ srp4, const int *__restrict__ scj, int *reach, int *maxchunks,
                                                  int *longer)
{
   int m = 0, k = 0; // (grid-stride, a small grid: the two atomics of a wavefront queue at the L2 behind everybody else's)
   int lg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // rows of more than 1, 2, 4, ..., 128 chunks
   for (long ql = (long)blockIdx.x * 256 + threadIdx.x; ql < n; ql += (long)gridDim.x * 256)
   {
      const int q = (int)ql, nc = srp4[q + 1] - srp4[q];
#pragma unroll
      for (int j = 0; j < 8; j++) lg[j] += (nc > (1 << j));
      k = max(k, nc);
      for (int e = 4 * srp4[q]; e < 4 * srp4[q + 1]; e++)
         if (scj[e] >= 0) m = max(m, abs(scj[e] - q));
   }
   for (int o = 32; o > 0; o >>= 1) { m = max(m, __shfl_xor(m, o)); k = max(k, __shfl_xor(k, o)); }
#pragma unroll
   for (int j = 0; j < 8; j++)
      for (int o = 32; o > 0; o >>= 1) lg[j] += __shfl_xor(lg[j], o);
   if ((threadIdx.x & 63) == 0)
   {
      if (m) atomicMax(reach, m);
      if (k) atomicMax(maxchunks, k);
#pragma unroll
      for (int j = 0; j < 8; j++)
         if (lg[j]) atomicAdd(&longer[j], lg[j]);
   }
}
__global__ void k_gs_chunk_hist(int n, const int *__restrict__ srp4, int *maxchunks, int *longer);
struct GsFreeShape { int ring = 0, lpr = 0, maxc = 1; bool lng = false; };
// can the barrier-free kernel take this sweep-order copy (chunk offsets rp4, columns cj)?  ring size, lanes per row and chunks per lane if so
static GsFreeShape gs_free_shape(const GsPlan &plan, int n, const int *rp4, const int *cj, const char *what, int known_reach = -1, int *reach_out = nullptr)
{
   GsFreeShape out;
   DArray<int> two(10);
   two.zero();
   if (known_reach >= 0) k_gs_chunk_hist<<<std::min(ceil_div(n, 256), 2048), 256, 0, STREAM>>>(n, rp4, two.data() + 1, two.data() + 2);
   else k_gs_reach<<<std::min(ceil_div(n, 256), 2048), 256, 0, STREAM>>>(n, rp4, cj, two.data(), two.data() + 1, two.data() + 2);
   int h[10] = {0};
   two.download(h, 10);
   if (known_reach >= 0) h[0] = known_reach; // (a bound: the copy holds a subset of the entries the reach was taken over)
   if (reach_out) *reach_out = h[0];
   const int *longer = h + 2; // rows of more than 1, 2, 4, ..., 128 chunks
   int maxblock = 0;
   for (size_t q = 0; q + 1 < plan.h_part.size(); q++) maxblock = std::max(maxblock, plan.h_part[q + 1] - plan.h_part[q]);
   int ring = 1024;
   if (maxblock > 0 && maxblock <= 16384)
      while (ring < maxblock) ring <<= 1; // the whole block in the ring: nothing is overwritten, any reach
   else
      while (ring < 4 * (h[0] + 1)) ring <<= 1; // W = ring / 4 must cover the longest reach
   // one chunk per lane for short rows (two chunks per lane and half the rounds measured slower there: 0.41 against 0.29 ms on the
   // 128^3 level 0, 0.80 against 0.58 on level 1 -- the sweep costs instructions, and eight entries per lane issue more of them per row
   // than two lanes with four each); two where a row would otherwise take 32 or 64 lanes and the blocks are long enough for the halved
   // number of rounds to count (128^3 level 2, rows of up to 69 entries, 7 000 rows per block: 0.47 -> 0.41 ms; level 3, 730 rows per
   // block: 0.079 -> 0.083, left at one)
   // (round 5: from 9 chunks in the longest row -- 8 lanes x 2 chunks instead of 16 x 1, and with half the rounds the barrier-free kernel
   //  beats the ring kernel on the level-1 operators of rank-block hierarchies: 69.9 -> 68.4 ms per 256^3 solve, 19.5 -> 18.5 ms at 128^3 on
   //  -P 4 4 4 blocks, nothing lost on lexicographic slabs; from 2 or 5 chunks -- one lane per 7-point row -- measured the same as 9)
   // Round 5, second step: the lanes of a row are sized for MOST rows, not for the longest -- capacity C = lanes x chunks per lane = the
   // smallest power of two that leaves at most 0.3 % of the rows longer; those read their further chunks inside stage C (`long_row`).  The
   // level-1 operator of the 256^3 rank-block hierarchy: 4.1 chunks per row on average, 99.86 % of the rows within 8, the longest 11 --
   // 16 lanes' worth of instructions were issued for every row, a quarter of them for entries; level 2: mean 12, 98 % within 16, longest 24.
   // (what a long row costs: loads from memory inside the update stage -- the wavefront waits for everything it has requested for the next
   //  three rounds, then for the L2, and every row behind it in the sweep waits with it.  Measured on the 256^3 rank-block hierarchy: 0.2 % of
   //  such rows on level 1 still leave 0.86 -> 0.60 ms per sweep for halving the lanes; 2.5 % on level 2 cost 0.44 -> 0.57 ms with half the
   //  lanes, and on blocks of a few hundred rows a single one shows (0.144 -> 0.184 ms): 0.3 % at most, and blocks of 2048 rows or more.)
   int C = 1;
   while (C < h[1] && C < 128 && (maxblock < 2048 || longer[__builtin_ctz(C)] > (int)(0.003 * n))) C <<= 1;
   constexpr int maxc2_from = 8;
   // (two chunks per lane cost 188 registers against 120 -- one workgroup to a CU instead of two -- and still win on blocks of 2048 rows
   //  and more: 64.6 against 66.2 ms per 256^3 solve with (4, 2) / (16, 2) against (8, 1) / (32, 1) lanes x chunks on levels 1 / 2)
   const int maxc = (C > 64 || (C >= maxc2_from && maxblock >= 2048)) ? 2 : 1;
   int       lpr  = std::max(C / maxc, 1);
   // the ring's guard (k_gs_blocks_free stage C): round 0 can start only if G = 512 / lpr groups fit twice into what the guard leaves
   // free, G <= RING - RING / 4 - G - 1 -- with one lane per row (rows of <= 4 off-diagonal entries: a 2-D five-point or tridiagonal
   // operator on blocks of more than 16384 rows) that needs 2048 slots where 1024 cover the reach (round-4 ADVICE: every wavefront
   // waited for a progress nobody could make, and the sweep ended in its spin limit)
   if (!(maxblock > 0 && maxblock <= 16384))
      while (lpr <= 64 && ring <= 16384 && 2 * (512 / lpr) > ring - ring / 4 - 1) ring <<= 1;
   if (ring > 16384 || lpr > 64)
   {
      if (getenv("HDA_VERBOSE"))
         fprintf(stderr, "[hda] block Gauss-Seidel plan: barrier-free kernel not used (dependency reach %d positions, %d chunks in the longest row)\n", h[0], h[1]);
      return out;
   }
   out.ring = ring;
   out.lpr  = lpr;
   out.maxc = maxc;
   out.lng  = lpr * maxc < h[1]; // rows beyond the lanes' capacity exist: the kernel with the long-row path
   if (getenv("HDA_VERBOSE"))
      fprintf(stderr, "[hda] block Gauss-Seidel plan: barrier-free kernel%s, dependency reach %d positions, largest block %d rows (ring %d), %d lanes per row, %d chunks per lane "
                      "(longest row %d chunks; %.2f %% of the rows beyond the lanes' %d)\n",
              what, h[0], maxblock, ring, lpr, maxc, h[1], lpr * maxc < 256 && lpr * maxc <= 128 ? 100.0 * longer[__builtin_ctz(lpr * maxc)] / std::max(n, 1) : 0.0, lpr * maxc);
   return out;
}

// the forward sweep from the zero guess -- every down sweep of a V(1,1) cycle, the L solve of a block ILU -- multiplies everything but a
// row's in-block columns EARLIER in the sweep by zero: a second copy with those entries alone (about half the operator) is what that sweep
// streams, with lanes sized for its shorter rows
__global__ __launch_bounds__(256) void k_gs_dep_count(int n, const int *__restrict__ srp4, const int *__restrict__ scj, int *__restrict__ len4)
{
   const int q = xcd_chunk_block() * 256 + threadIdx.x;
   if (q >= n) return;
   int c = 0;
   for (int e = 4 * srp4[q]; e < 4 * srp4[q + 1]; e++) c += (scj[e] >= 0 && scj[e] < q);
   len4[q] = (c + 3) >> 2;
}
__global__ __launch_bounds__(256) void k_gs_dep_fill(int n, const int *__restrict__ srp4, const int *__restrict__ scj, const double *__restrict__ sv,
                                                     const int *__restrict__ lrp4, int *__restrict__ lcj, double *__restrict__ lv)
{ // eight lanes to a row, a chunk of four entries each per step; the entries keep their order
   constexpr int G = 8;
   const int     q = xcd_chunk_block() * (256 / G) + (int)threadIdx.x / G, gl = (int)threadIdx.x % G;
   if (q >= n) return;
   const int c0 = srp4[q], c1 = srp4[q + 1];
   int       d  = 4 * lrp4[q];
   for (int base = c0; base < c1; base += G)
   {
      const int  ch   = base + gl;
      const bool mine = ch < c1;
      int        col[4];
      double     val[4];
      int        cnt = 0;
#pragma unroll
      for (int e = 0; e < 4; e++)
      {
         col[e] = mine ? scj[4 * ch + e] : -1;
         val[e] = mine ? sv[4 * ch + e] : 0.0;
         cnt += (col[e] >= 0 && col[e] < q);
      }
      int before = cnt; // inclusive prefix over the group's lanes
#pragma unroll
      for (int o = 1; o < G; o <<= 1)
      {
         const int up = __shfl_up(before, o, G);
         if (gl >= o) before += up;
      }
      const int total = __shfl(before, G - 1, G);
      int       w     = d + before - cnt;
#pragma unroll
      for (int e = 0; e < 4; e++)
         if (col[e] >= 0 && col[e] < q) { lcj[w] = col[e]; lv[w] = val[e]; w++; }
      d += total;
   }
   for (int w = d + gl, e1 = 4 * lrp4[q + 1]; w < e1; w += G) { lcj[w] = q; lv[w] = 0.0; }
}
// chunks per row alone (maximum and the counts of rows beyond 1, 2, 4, ... chunks): what gs_free_shape needs of a copy whose reach is known
__global__ __launch_bounds__(256) void k_gs_chunk_hist(int n, const int *__restrict__ srp4, int *maxchunks, int *longer)
{
   int k = 0, lg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256)
   {
      const int nc = srp4[q + 1] - srp4[q];
#pragma unroll
      for (int j = 0; j < 8; j++) lg[j] += (nc > (1 << j));
      k = max(k, nc);
   }
   for (int o = 32; o > 0; o >>= 1) k = max(k, __shfl_xor(k, o));
#pragma unroll
   for (int j = 0; j < 8; j++)
      for (int o = 32; o > 0; o >>= 1) lg[j] += __shfl_xor(lg[j], o);
   if ((threadIdx.x & 63) == 0)
   {
      if (k) atomicMax(maxchunks, k);
#pragma unroll
      for (int j = 0; j < 8; j++)
         if (lg[j]) atomicAdd(&longer[j], lg[j]);
   }
}
static void gs_free_plan(const GsPlan &plan, int n)
{
   plan.free_lpr = plan.dep_lpr = 0;
   if (!plan.sorted || !plan.s_x0.size()) return;
   int               reach = 0;
   const GsFreeShape f     = gs_free_shape(plan, n, plan.s_rowptr.data(), plan.s_col.data(), "", -1, &reach);
   if (f.lpr == 0) return;
   plan.free_ring = f.ring;
   plan.free_lpr  = f.lpr;
   plan.free_maxc = f.maxc;
   plan.free_long = f.lng;
   // the dependency copy (forward sweeps from zero)
   DArray<int> len4((size_t)n + 1);
   k_gs_dep_count<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, plan.s_rowptr.data(), plan.s_col.data(), len4.data());
   plan.d_rowptr.alloc((size_t)n + 1);
   exclusive_scan(n, len4.data(), plan.d_rowptr.data(), nullptr);
   int chunks = 0;
   HDA_HIP(hipMemcpyAsync(&chunks, plan.d_rowptr.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   plan.d_col.alloc((size_t)4 * chunks + 4); // (one spare chunk, as in the full copy)
   plan.d_val.alloc((size_t)4 * chunks + 4);
   HDA_HIP(hipMemsetAsync(plan.d_col.data() + (size_t)4 * chunks, 0, 4 * sizeof(int), STREAM));
   HDA_HIP(hipMemsetAsync(plan.d_val.data() + (size_t)4 * chunks, 0, 4 * sizeof(double), STREAM));
   k_gs_dep_fill<<<8 * ceil_div(ceil_div(n, 32), 8), 256, 0, STREAM>>>(n, plan.s_rowptr.data(), plan.s_col.data(), plan.s_val.data(), plan.d_rowptr.data(),
                                                      plan.d_col.data(), plan.d_val.data());
   const GsFreeShape g = gs_free_shape(plan, n, plan.d_rowptr.data(), plan.d_col.data(), " (dependency copy)", reach);
   if (g.lpr == 0) return;
   plan.dep_ring = g.ring;
   plan.dep_lpr  = g.lpr;
   plan.dep_maxc = g.maxc;
   plan.dep_long = g.lng;
}
static int           *g_free_err = nullptr;
static std::once_flag g_free_err_once;
static int *gs_free_error_flag()
{ // (one flag per process, shared by the rank threads of the test seam: allocated once, under a lock)
   std::call_once(g_free_err_once, [] {
      int *p = nullptr;
      HDA_HIP(hipMalloc((void **)&p, sizeof(int)));
      HDA_HIP(hipMemset(p, 0, sizeof(int)));
      g_free_err = p;
   });
   return g_free_err;
}
// did a barrier-free sweep run into its spin limit since the last look?  (one 4-byte read-back; the callers are at a host sync anyway)
void gs_free_check()
{
   if (!g_free_err) return;
   int e = 0;
   HDA_HIP(hipMemcpyAsync(&e, g_free_err, sizeof(int), hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   if (e)
   {
      HDA_HIP(hipMemsetAsync(g_free_err, 0, sizeof(int), STREAM));
      throw Error("barrier-free Gauss-Seidel sweep: a row waited beyond the spin limit (protocol error " + std::to_string(e) + ")");
   }
}

template <int LPR, int MAXC, bool LONG>
static void gs_blocks_free_t(const DCsr &A, const GsPlan &p, const double *dinv, const double *b, const double *xin, double *xout, bool forward,
                             bool zero_in, bool dep)
{ // dep: the dependency copy (forward from zero: gs_free_plan) instead of the whole operator
   const int n = A.nrows;
   constexpr int NT = 512; // (1024 threads leave a lane 128 registers: the four rows in flight spill)
   int *err = gs_free_error_flag(); // (one flag per process: the kernel raises it instead of spinning forever; read by gs_free_check)
   // (what the kernel reads in sweep order: divisors and right-hand side, gathered when they are new; the sweep-start iterate, gathered
   //  unless it is zero.  Its result it writes in sweep order too, and a second kernel scatters it: stores to the caller's numbering
   //  from inside the sweep were tried in round 5 -- 8-byte stores all over a block's part of x, whose lines leave the L2 half
   //  written on blocks of 65 536 rows and more: 106.6 -> 115.5 ms per 256^3 solve on 64 blocks, nothing gained on 512)
   if (!zero_in || p.to_b || p.to_d)
      k_gs_to_sweep_order<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, zero_in ? 1 : 0, p.perm.data(), xin, p.to_b, p.to_d, nullptr, p.s_b.data(), p.s_d.data(),
                                                             zero_in ? nullptr : p.s_x0.data());
   const int     ring = dep ? p.dep_ring : p.free_ring;
   const size_t  lds  = sizeof(double) * (size_t)ring;
   const int    *rp4  = dep ? p.d_rowptr.data() : p.s_rowptr.data();
   const int4   *cj4  = (const int4 *)(dep ? p.d_col.data() : p.s_col.data());
   const double2 *v2  = (const double2 *)(dep ? p.d_val.data() : p.s_val.data());
#define HDA_GS_FREE_LAUNCH(BW)                                                                                                               \
   HDA_HIP(hipFuncSetAttribute((const void *)k_gs_blocks_free<LPR, MAXC, NT, BW, LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));  \
   k_gs_blocks_free<LPR, MAXC, NT, BW, LONG><<<p.nblk, NT, lds, STREAM>>>(ring - 1, zero_in ? 1 : 0, p.blk_part.data(), rp4, cj4, v2, p.s_d.data(), \
                                                                    p.s_b.data(), p.s_aii.data(), p.s_x0.data(), p.s_x.data(), err)
   if (forward) { HDA_GS_FREE_LAUNCH(false); }
   else { HDA_GS_FREE_LAUNCH(true); }
#undef HDA_GS_FREE_LAUNCH
   k_gs_from_sweep_order<<<xcd_chunk_grid(n), 256, 0, STREAM>>>(n, p.perm.data(), p.s_x.data(), xout);
   static const bool check = getenv("HDA_GS_FREE_CHECK") != nullptr; // (tests: after every sweep; otherwise at the end of every Krylov solve)
   if (check) gs_free_check();
}

template <int LPR, int NPF>
static void gs_blocks_t(const DCsr &A, const GsPlan &p, const double *dinv, const double *b, const double *xin, double *xout, bool forward,
                        bool zero_in)
{
   const int    lds_levels = std::min(p.blk_max_levels + 1, 12 * 1024); // (48 KB of level offsets at most; longer blocks read them from memory)
   const size_t lds        = sizeof(int) * (size_t)lds_levels;
   k_gs_blocks<LPR, NPF><<<p.nblk, 1024, lds, STREAM>>>(forward ? 0 : 1, zero_in ? 1 : 0, lds_levels, p.blk_part.data(), p.blk_lvl_ptr.data(),
                                                       p.blk_lvl.data(), p.perm.data(), p.rbeg.data(), p.rend.data(), A.col.data(), A.val.data(),
                                                       dinv, b, xin, xout);
}

void gs_sweep_blocks(const DCsr &A, const GsPlan &plan, const double *dinv, const double *b, const double *xin, double *xout, bool forward,
                     bool zero_in, bool b_unchanged)
{
   HDA_REQUIRE(plan.built && plan.nblk > 0, "row-block Gauss-Seidel plan missing");
   HDA_REQUIRE(zero_in || (xin && xin != xout), "row-block Gauss-Seidel sweeps out of place");
   if (A.nrows == 0) return;
   // the sweep-order copies of the divisors and of the right-hand side are reused where they are current (round-5 series-B trace:
   // k_gs_to_sweep_order / k_gs_from_sweep_order were 18.5 of 80 ms per solve, two thirds of the gathers of the first for b and dinv)
   plan.to_d   = (plan.sd_src == dinv && plan.s_d.size() == (size_t)A.nrows) ? nullptr : dinv;
   plan.to_b   = (b_unchanged && plan.sb_src == b && plan.s_b.size() == (size_t)A.nrows) ? nullptr : b;
   plan.sd_src = dinv;
   plan.sb_src = b;
   if (plan.span_rp != A.rowptr.data() || plan.span_nnz != A.nnz || plan.span_gen != A.gen)
   { // another matrix behind a kept plan (preconditioner.reuse)
      gs_row_spans(A, plan);
      if (plan.sorted) gs_sorted_copy(A, plan);
      if (plan.ring) gs_ring_copy(A, plan);
      if (plan.sorted) gs_free_plan(plan, A.nrows);
   }
   const double a = A.avg_row();
   constexpr bool use_sorted = true;
   // barrier-free kernel or ring kernel (gs_free_beats_ring; the plan holds the level-wise copy only where that said ring) --
   // HDA_GS_FREE=1 / 0 force either
   bool use_free = plan.free_lpr > 0 && use_sorted;
   if (use_free && plan.ring && !getenv("HDA_GS_FREE"))
      use_free = gs_free_beats_ring(plan, (double)plan.r_pass[0].size() / 8.0 / std::max(plan.nblk, 1) - 3.0); // (three spare records per block)
   if (getenv("HDA_GS_FREE")) use_free = use_free && atoi(getenv("HDA_GS_FREE")) != 0;
   if (use_free)
   {
#define HDA_GS_FREE(L)                                                                                      \
   do                                                                                                       \
   {                                                                                                        \
      if (f_long)                                                                                           \
      {                                                                                                     \
         if (f_maxc == 2) gs_blocks_free_t<L, 2, true>(A, plan, dinv, b, xin, xout, forward, zero_in, dep); \
         else gs_blocks_free_t<L, 1, true>(A, plan, dinv, b, xin, xout, forward, zero_in, dep);             \
      }                                                                                                     \
      else if (f_maxc == 2) gs_blocks_free_t<L, 2, false>(A, plan, dinv, b, xin, xout, forward, zero_in, dep); \
      else gs_blocks_free_t<L, 1, false>(A, plan, dinv, b, xin, xout, forward, zero_in, dep);               \
   } while (0)
      // a forward sweep from zero streams the dependency copy where the plan has one (HDA_GS_DEP=0: the whole operator, for the tests)
      const bool dep    = zero_in && forward && plan.dep_lpr > 0 && !(getenv("HDA_GS_DEP") && atoi(getenv("HDA_GS_DEP")) == 0);
      const int  f_lpr  = dep ? plan.dep_lpr : plan.free_lpr, f_maxc = dep ? plan.dep_maxc : plan.free_maxc;
      const bool f_long = dep ? plan.dep_long : plan.free_long;
      switch (f_lpr)
      {
         case 1: HDA_GS_FREE(1); break;
         case 2: HDA_GS_FREE(2); break;
         case 4: HDA_GS_FREE(4); break;
         case 8: HDA_GS_FREE(8); break;
         case 16: HDA_GS_FREE(16); break;
         case 32: HDA_GS_FREE(32); break;
         default: HDA_GS_FREE(64); break;
      }
#undef HDA_GS_FREE
      return;
   }
   if (plan.ring && use_sorted && !(getenv("HDA_GS_RING") && atoi(getenv("HDA_GS_RING")) == 0))
   { // (lanes per row and workgroup size were fixed when the pass lists were made)
#define HDA_GS_RING_NT(L)                                                                                      \
   do                                                                                                          \
   {                                                                                                           \
      if (plan.ring_nt == 256) gs_blocks_ring_t<L, 256>(A, plan, dinv, b, xin, xout, forward, zero_in);        \
      else if (plan.ring_nt == 512) gs_blocks_ring_t<L, 512>(A, plan, dinv, b, xin, xout, forward, zero_in);   \
      else gs_blocks_ring_t<L, 1024>(A, plan, dinv, b, xin, xout, forward, zero_in);                           \
   } while (0)
      if (plan.ring_lpr == 2) HDA_GS_RING_NT(2);
      else if (plan.ring_lpr == 4) HDA_GS_RING_NT(4);
      else if (plan.ring_lpr == 8) HDA_GS_RING_NT(8);
      else HDA_GS_RING_NT(16);
#undef HDA_GS_RING_NT
      return;
   }
   if (plan.sorted && use_sorted)
   {
      const int lpr = (a <= 10.0) ? 2 : (a <= 40.0) ? 8 : 16; // lanes per row: 4 entries each per pass
      // workgroup size: what a typical level of a block keeps busy (every wavefront of the workgroup runs every pass and meets every
      // barrier, rows or not: idle ones only take issue slots and queue loads in front of the busy ones')
      const double     lanes  = plan.blk_mean_rows_per_level * std::min(std::max(lpr, 2), 16);
      const int        nt     = lanes <= 192.0 ? 256 : lanes <= 640.0 ? 512 : 1024;
#define HDA_GS_DISPATCH(L)                                                                              \
   do                                                                                                   \
   {                                                                                                    \
      if (nt <= 256) gs_blocks_sorted_t<L, 256>(A, plan, dinv, b, xin, xout, forward, zero_in);         \
      else if (nt <= 512) gs_blocks_sorted_t<L, 512>(A, plan, dinv, b, xin, xout, forward, zero_in);    \
      else gs_blocks_sorted_t<L, 1024>(A, plan, dinv, b, xin, xout, forward, zero_in);                  \
   } while (0)
      if (lpr <= 2) HDA_GS_DISPATCH(2);
      else if (lpr <= 4) HDA_GS_DISPATCH(4);
      else if (lpr <= 8) HDA_GS_DISPATCH(8);
      else HDA_GS_DISPATCH(16);
#undef HDA_GS_DISPATCH
      return;
   }
   if (a <= 10.0) gs_blocks_t<4, 2>(A, plan, dinv, b, xin, xout, forward, zero_in);
   else if (a <= 40.0) gs_blocks_t<8, 8>(A, plan, dinv, b, xin, xout, forward, zero_in);
   else gs_blocks_t<16, 8>(A, plan, dinv, b, xin, xout, forward, zero_in);
}

// LPR lanes per row in the per-level launches, FL in the single-workgroup runs of small levels: there a level is a few
// hundred rows at most and every round of rows is a chain of dependent loads (permutation, row pointer, entries, x), so
// the narrower group -- four times the rows in flight -- wins over the wider reduction
template <int LPR, int FL>
static void gs_sweep_t(const DCsr &A, const GsPlan &p, const double *dinv, const double *b, double *x, bool forward)
{
   const int ns = (int)p.segments.size();
   for (int si = 0; si < ns; si++)
   {
      const auto &sg = p.segments[(size_t)(forward ? si : ns - 1 - si)];
      if (sg.second - sg.first == 1 && p.lvl_ptr[(size_t)sg.first + 1] - p.lvl_ptr[(size_t)sg.first] > 512)
      {
         const int first = p.lvl_ptr[(size_t)sg.first], count = p.lvl_ptr[(size_t)sg.first + 1] - first;
         const int grid  = std::min(ceil_div((long long)count * LPR, 256), 2048);
         k_gs_level<LPR><<<grid, 256, 0, STREAM>>>(first, count, p.perm.data(), p.rbeg.data(), p.rend.data(), A.col.data(), A.val.data(), dinv, b, x);
      }
      else
      {
         const size_t     lds  = sizeof(int) * (size_t)(sg.second - sg.first + 1);
         if (lds <= 48 * 1024)
            k_gs_levels_pipe<FL, (FL >= 8 ? 8 : 2)><<<1, 1024, lds, STREAM>>>(sg.first, sg.second, forward ? 0 : 1, p.d_lvl_ptr.data(), p.perm.data(),
                                                                          p.rbeg.data(), p.rend.data(), A.col.data(), A.val.data(), dinv, b, x);
         else
            k_gs_levels_fused<FL><<<1, 1024, 0, STREAM>>>(sg.first, sg.second, forward ? 0 : 1, p.d_lvl_ptr.data(), p.perm.data(),
                                                         p.rbeg.data(), p.rend.data(), A.col.data(), A.val.data(), dinv, b, x);
      }
   }
}

// hypre_BoomerAMGRelax types 3/13 (forward) and 4/14 (backward); dinv = weight / d with
// d = a_ii (3/4) or the l1 divisor of option 4 (13/14)
void gs_sweep(const DCsr &A, const GsPlan &plan, const double *dinv, const double *b, double *x, bool forward)
{
   HDA_REQUIRE(plan.built, "Gauss-Seidel plan missing");
   if (A.nrows == 0) return;
   if (plan.span_rp != A.rowptr.data() || plan.span_nnz != A.nnz || plan.span_gen != A.gen) gs_row_spans(A, plan); // another matrix behind a kept plan (preconditioner.reuse)
   const double a = A.avg_row();
   if (a <= 10.0) gs_sweep_t<8, 4>(A, plan, dinv, b, x, forward);
   else if (a <= 40.0) gs_sweep_t<32, 8>(A, plan, dinv, b, x, forward);
   else gs_sweep_t<64, 16>(A, plan, dinv, b, x, forward);
}

} // namespace hda
