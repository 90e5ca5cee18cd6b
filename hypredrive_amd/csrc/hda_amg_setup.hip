// hda_amg_setup.hip -- AMG setup on the device: strength of connection, PMIS C/F
// splitting, extended+i interpolation with truncation, deterministic SpGEMM for the
// Galerkin product.  (upstream algorithms in App. A.4-A.7.)
//
// Determinism: every floating-point accumulation below runs in the same order as the
// sequential CPU oracle (one thread owns a row and walks it in ascending k), the file is
// built with -ffp-contract=off, and integer set algorithms (PMIS) are formulated as
// synchronous rounds.  The hierarchy is therefore bit-reproducible run to run and
// independent of the launch geometry.
#include "hda_amg.h"
#include "hda_sort.h"


#include <algorithm>
#include <chrono>
#include <cstring>

namespace hda {

#define STREAM (Context::get().stream)

void exclusive_scan64(long n, const int *in, long long *out); // hda_kernels.hip

// ---------------------------------------------------------------- strength

__global__ __launch_bounds__(256) void k_strength(int n, const int *__restrict__ rp,
                                                  const int *__restrict__ cj,
                                                  const double *__restrict__ v, double theta,
                                                  double mrs, unsigned char *__restrict__ smask,
                                                  int *__restrict__ ns, const int *__restrict__ dof)
{ // dof != nullptr: systems AMG (unknown approach) -- other functions' couplings take no part
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int k0 = rp[i], k1 = rp[i + 1];
   double    diag = 0.0, row_sum = 0.0, row_scale = 0.0;
   const int fi = dof ? dof[i] : 0;
   for (int k = k0; k < k1; k++)
      if (cj[k] == i) diag = v[k];
   for (int k = k0; k < k1; k++)
   {
      if (dof && dof[cj[k]] != fi) continue;
      const double a = v[k];
      row_sum += a;
      if (cj[k] == i) continue;
      if (diag < 0.0) row_scale = (a > row_scale) ? a : row_scale;
      else row_scale = (a < row_scale) ? a : row_scale;
   }
   const bool weak = (mrs < 1.0) && (diag != 0.0) && (fabs(row_sum / diag) > mrs);
   int        cnt  = 0;
   for (int k = k0; k < k1; k++)
   {
      int s = 0;
      if (cj[k] != i && !weak && !(dof && dof[cj[k]] != fi)) s = (diag < 0.0) ? (v[k] > theta * row_scale) : (v[k] < theta * row_scale);
      smask[k] = (unsigned char)s;
      cnt += s;
   }
   ns[i] = cnt;
}

// The same classification with G lanes per row (coalesced reads of long rows).  The row sum keeps the sequential
// order: the lanes hand their values round in ascending k and every lane adds them in that order; the scale is a
// min/max, which no order changes.
template <int G>
__global__ __launch_bounds__(256) void k_strength_grp(int n, const int *__restrict__ rp, const int *__restrict__ cj,
                                                      const double *__restrict__ v, double theta, double mrs,
                                                      unsigned char *__restrict__ smask, int *__restrict__ ns,
                                                      const int *__restrict__ dof)
{
   const int  gl  = threadIdx.x & (G - 1);
   const long row = ((long)blockIdx.x * 256 + threadIdx.x) / G;
   const bool in  = row < n;
   const int  i   = in ? (int)row : 0;
   const int  k0 = in ? rp[i] : 0, k1 = in ? rp[i + 1] : 0;
   const int  fi = dof ? dof[i] : 0;
   double     d  = 0.0;
   for (int k = k0 + gl; k < k1; k += G)
      if (cj[k] == i) d = v[k];
   for (int o = G >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, G); // one lane holds the diagonal, the others 0
   const double diag = d;
   double       row_sum = 0.0, row_scale = 0.0;
   for (int base = k0; base < k1; base += G)
   {
      const int k = base + gl;
      double    a = 0.0;
      bool      off = false; // an off-diagonal entry of the row's own function
      if (k < k1)
      {
         const int c = cj[k];
         if (!(dof && dof[c] != fi))
         {
            a   = v[k];
            off = (c != i);
         }
      }
      if (off)
      {
         if (diag < 0.0) row_scale = (a > row_scale) ? a : row_scale;
         else row_scale = (a < row_scale) ? a : row_scale;
      }
      const int m = min(G, k1 - base);
      for (int l = 0; l < m; l++) row_sum += __shfl(a, l, G); // skipped entries add 0.0
   }
   for (int o = G >> 1; o > 0; o >>= 1)
   {
      const double t = __shfl_xor(row_scale, o, G);
      if (diag < 0.0) row_scale = (t > row_scale) ? t : row_scale;
      else row_scale = (t < row_scale) ? t : row_scale;
   }
   const bool weak = (mrs < 1.0) && (diag != 0.0) && (fabs(row_sum / diag) > mrs);
   int        cnt  = 0;
   for (int k = k0 + gl; k < k1; k += G)
   {
      int       s = 0;
      const int c = cj[k];
      if (c != i && !weak && !(dof && dof[c] != fi)) s = (diag < 0.0) ? (v[k] > theta * row_scale) : (v[k] < theta * row_scale);
      smask[k] = (unsigned char)s;
      cnt += s;
   }
   for (int o = G >> 1; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, G);
   if (in && gl == 0) ns[i] = cnt;
}

static void strength_ns(const DCsr &A, double theta, double mrs, unsigned char *smask, int *ns, const int *dof = nullptr)
{
   if (!A.nrows) return;
   const double avg = A.avg_row();
   const int    n   = A.nrows;
   if (avg <= 12.0)
      k_strength<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), theta, mrs, smask, ns, dof);
   else if (avg <= 24.0)
      k_strength_grp<16><<<ceil_div((long long)n * 16, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), theta, mrs, smask, ns, dof);
   else if (avg <= 48.0)
      k_strength_grp<32><<<ceil_div((long long)n * 32, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), theta, mrs, smask, ns, dof);
   else
      k_strength_grp<64><<<ceil_div((long long)n * 64, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), theta, mrs, smask, ns, dof);
}

void amg_strength(const DCsr &A, double theta, double max_row_sum, unsigned char *smask, const int *dof)
{
   DArray<int> ns((size_t)A.nrows + 1);
   strength_ns(A, theta, max_row_sum, smask, ns.data(), dof);
}

// -------------------------------------------------------------------- PMIS

__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
   z += 0x9E3779B97F4A7C15ULL;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   return z ^ (z >> 31);
}
// tie-break weight in [0,1): hash of the GLOBAL row id => partition independent
__device__ __forceinline__ double pmis_rand(unsigned long long seed, int level, long long gid)
{
   unsigned long long h = mix64(mix64(seed + (unsigned long long)level * 0x100000001B3ULL) ^ (unsigned long long)gid);
   return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

__global__ __launch_bounds__(256) void k_indeg(int nnz, const int *__restrict__ cj,
                                               const unsigned char *__restrict__ smask, int *indeg)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256)
      if (smask[k]) atomicAdd(&indeg[cj[k]], 1);
}

// counter += the wavefront's sum of cnt: ONE atomic per wavefront and launch.  Atomics on one address are serialised at the L2
// (~10 ns each): one per undecided row -- 16.7 M in the first PMIS round at 256^3 -- or even one per wavefront of a thread-per-row
// grid (262 144) made k_pmis_init and k_pmis_setF take 2.9 ms each on level 0 where their loads need 0.1-0.3 ms (round-5 setup
// accounting, profiles/r05_setup_accounting.md).  The kernels below walk the rows with a grid-stride loop on at most kPmisGrid
// workgroups and count in a register.
constexpr int kPmisGrid = 2048;
__device__ __forceinline__ void wave_sum_add(int *counter, int cnt)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
   if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(counter, cnt);
}

// rows [r0, r0 + n); global id of row i = gid[i] when a table is given, else row_offset + i
__global__ __launch_bounds__(256) void k_pmis_init(int r0, int n, const int *__restrict__ ns,
                                                   const int *__restrict__ indeg,
                                                   unsigned long long seed, int level,
                                                   long long row_offset, const long long *__restrict__ gid,
                                                   double *__restrict__ meas, int *__restrict__ cf, int *counter)
{
   int left = 0;
   for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256)
   {
      const int i  = r0 + (int)q;
      const int nt = indeg[i];
      meas[i]      = (double)nt + pmis_rand(seed, level, gid ? gid[i] : row_offset + i);
      int c;
      if (ns[i] == 0) c = -3;      // no strong dependence: special F, never interpolated
      else if (nt == 0) c = -1;    // measure < 1: nobody depends on it
      else { c = 0; left++; }
      cf[i] = c;
   }
   wave_sum_add(counter, left);
}

// one edge visit decides both endpoints (hypre's IndepSet loop): the neighbourhood is S u S^T
__global__ __launch_bounds__(256) void k_pmis_mark(int n, const int *__restrict__ rp,
                                                   const int *__restrict__ cj,
                                                   const unsigned char *__restrict__ smask,
                                                   const int *__restrict__ cf,
                                                   const double *__restrict__ meas, unsigned char *notmax, int r0)
{
   const int i = r0 + blockIdx.x * 256 + threadIdx.x;
   if (i >= r0 + n || cf[i] != 0) return;
   const double mi = meas[i];
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      if (!smask[k]) continue;
      const int j = cj[k];
      if (cf[j] != 0) continue;
      const double mj = meas[j];
      if (mj > mi || (mj == mi && j > i)) notmax[i] = 1; // index order == global id order in both layouts used
      else notmax[j] = 1;
   }
}
__global__ __launch_bounds__(256) void k_pmis_setC(int n, int *__restrict__ cf, unsigned char *__restrict__ notmax, int r0)
{
   const int i = r0 + blockIdx.x * 256 + threadIdx.x;
   if (i >= r0 + n) return;
   if (cf[i] == 0 && !notmax[i]) cf[i] = 1;
   notmax[i] = 0;
}
__global__ __launch_bounds__(256) void k_pmis_setF(int n, const int *__restrict__ rp,
                                                   const int *__restrict__ cj,
                                                   const unsigned char *__restrict__ smask, int *cf, int *counter, int r0)
{
   int left = 0; // rows of this lane that are still undecided after this round
   for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256)
   {
      const int i = r0 + (int)q;
      if (cf[i] != 0) continue;
      bool f = false;
      for (int k = rp[i]; k < rp[i + 1] && !f; k++) f = smask[k] && cf[cj[k]] == 1;
      if (f) cf[i] = -1;
      else left++;
   }
   wave_sum_add(counter, left);
}

static void pmis_core(const DCsr &A, const unsigned char *smask, const int *ns, uint64_t seed, int level,
                      long long row_offset, int *cf)
{
   const int n = A.nrows;
   if (!n) return;
   DArray<int>           indeg((size_t)n), counter(1);
   DArray<double>        meas((size_t)n);
   DArray<unsigned char> notmax((size_t)n);
   indeg.zero();
   notmax.zero();
   counter.zero();
   const int g = ceil_div(n, 256);
   if (A.nnz) k_indeg<<<std::min(ceil_div(A.nnz, 256), 1 << 16), 256, 0, STREAM>>>(A.nnz, A.col.data(), smask, indeg.data());
   k_pmis_init<<<std::min(g, kPmisGrid), 256, 0, STREAM>>>(0, n, ns, indeg.data(), seed, level, row_offset, nullptr, meas.data(), cf, counter.data());
   int left = 0;
   counter.download(&left, 1);
   HDA_TRACE("  pmis: init done, undecided=%d", left);
   // (Measured at 256^3: rounds over a compacted worklist of the undecided rows are no faster -- 12.6 / 13.0 ms
   // against 12.4 / 11.4 on levels 0 / 1 -- and several lanes per row are three times slower on level 1; the
   // first two rounds, which touch nearly every row, carry the cost.)
   int rounds = 0;
   while (left > 0)
   {
      counter.zero();
      k_pmis_mark<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, meas.data(), notmax.data(), 0);
      k_pmis_setC<<<g, 256, 0, STREAM>>>(n, cf, notmax.data(), 0);
      k_pmis_setF<<<std::min(g, kPmisGrid), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, counter.data(), 0);
      counter.download(&left, 1);
      HDA_REQUIRE(++rounds < 10000, "PMIS did not terminate");
   }
}

__global__ __launch_bounds__(256) void k_count_strong(int n, const int *__restrict__ rp,
                                                      const unsigned char *__restrict__ sm, int *__restrict__ ns)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int c = 0;
   for (int k = rp[i]; k < rp[i + 1]; k++) c += sm[k];
   ns[i] = c;
}

void amg_pmis(const DCsr &A, const unsigned char *smask, uint64_t seed, int level, long long row_offset, int *cf)
{
   DArray<int> ns((size_t)A.nrows + 1);
   if (A.nrows) k_count_strong<<<ceil_div(A.nrows, 256), 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), smask, ns.data());
   pmis_core(A, smask, ns.data(), seed, level, row_offset, cf);
}

// ------------------------------------------------------------ Ruge first pass (HMIS on one rank)
// hypre's coarsen type 10 (HMIS, the CPU default the reference's refOutputs were made with,
// src/internal/amg.c:303-308) is the first pass of Ruge-Stueben on each rank's interior
// followed by PMIS on what is left; on a single rank the first pass decides everything.  The
// pass is a priority-queue sweep: the next C point is the HEAD of the highest non-empty measure
// bucket, buckets are FIFO lists.  That order is the algorithm, so it runs as ONE device thread
// (kept on the device so the setup never leaves HBM).  Meant for parity runs on the
// reference's small examples; PMIS is the coarsening for large problems.
// Ruge first pass of HMIS.  S = strong entries of A (CSR, rp/cj), T = its transpose (tp/tj, ascending).  One wavefront per row block
// (lane 0 works; the pass is sequential by definition), all blocks at once: block q = rows [part[q], part[q + 1]), connections that
// leave the block ignored (a rank's S_diag), measures = in-block dependants (a measure never exceeds twice the in-block dependants:
// each dependant counts once as such and once more when it turns F).  (The round-3 form with one array per field was removed in
// round 5: same result bit for bit, 25 % slower, profiles/r04_gs_blocks.md.)
// The state of a row is ONE 16-byte record {prev, next, measure, cf} (one load where one array per field makes
// four dependent ones; a listed point's bucket key IS its measure) and the bucket heads / tails in LDS when twice the largest
// in-degree fits (they are the hottest words of the pass).  What the pass costs is the chain of dependent memory round trips of its
// one working lane, so it is those that are cut: per list operation one record load (stores are not waited for), against the
// record's three or four loads plus the bucket's before.  Same order of operations, same result bit for bit.
__device__ __forceinline__ int4 rs_load(const int4 *rec, int p) { return rec[p]; }
#define RS_F(p, f) (((int *)(rec + (p)))[f]) // field f of row p's record: 0 prev, 1 next, 2 measure, 3 cf
template <bool LDSHT>
__global__ __launch_bounds__(64) void k_rs_first_pass_rec(int nblk, const int *__restrict__ part, const int *__restrict__ rp, const int *__restrict__ cj,
                                                          const int *__restrict__ tp, const int *__restrict__ tj, int *head_all, int *tail_all,
                                                          int4 *rec, int *__restrict__ cf, int kcap)
{
   extern __shared__ int s_ht[]; // LDSHT: heads [0, kcap), tails [kcap, 2 kcap)
   const int q = blockIdx.x;
   if (q >= nblk) return;
   const int lo = part[q], hi = part[q + 1];
   int      *head, *tail;
   int       nbk;
   if constexpr (LDSHT)
   {
      head = s_ht;
      tail = s_ht + kcap;
      nbk  = kcap;
   }
   else
   {
      head = head_all + 2 * (size_t)lo + 2 * (size_t)q;
      tail = tail_all + 2 * (size_t)lo + 2 * (size_t)q;
      nbk  = 2 * (hi - lo) + 2;
   }
   auto in = [&](int j) { return j >= lo && j < hi; };
   for (int t = threadIdx.x; t < nbk; t += 64) head[t] = tail[t] = -1;
   for (int i = lo + threadIdx.x; i < hi; i += 64)
   {
      int nt = 0;
      for (int k = tp[i]; k < tp[i + 1]; k++) nt += in(tj[k]);
      const bool special = (rp[i + 1] == rp[i]); // no strong dependence at all: special F, never interpolated
      rec[i] = make_int4(-1, -1, special ? 0 : nt, special ? -3 : 0);
      cf[i]  = special ? -3 : 0;
   }
   __threadfence_block();
   __syncthreads();
   // From here on ONE lane (lane 0) does every list operation, in the order of the sequential pass; the other 63 stay in the loop
   // with it and do the reads that do NOT depend on the lists, a whole row of them per load instruction: the neighbour indices of a
   // row, and for each neighbour its record (decided or not -- which only the neighbour's own turn changes, never the turns of the
   // others of the same row -- and a first touch of the line the working lane reads next) and its row bounds.  The working lane then
   // visits the undecided neighbours only.  What the pass costs is its chain of dependent round trips to memory (round 4: 2.7 us
   // per row); the chain loses the reads of the decided neighbours and every index read.
   const int lane   = threadIdx.x;
   int       maxkey = 0; // (lane 0's)
   auto unlink = [&](const int4 &R) { // R: the record of a listed point (key = its measure)
      if (R.x >= 0) RS_F(R.x, 1) = R.y;
      else head[R.z] = R.y;
      if (R.y >= 0) RS_F(R.y, 0) = R.x;
      else tail[R.z] = R.x;
   };
   auto enter = [&](int p, int key) { // at the tail of bucket `key`; the point is undecided
      const int t = tail[key];
      rec[p]      = make_int4(t, -1, key, 0);
      if (t >= 0) RS_F(t, 1) = p;
      else head[key] = p;
      tail[key] = p;
      if (key > maxkey) maxkey = key;
   };
   auto bump = [&](int m) { // an undecided listed point gains one: to the tail of the next bucket
      const int4 R = rs_load(rec, m);
      if (R.w != 0) return;
      unlink(R);
      enter(m, R.z + 1);
   };
   auto first_set = [](unsigned long long &mask) { // lowest set lane of a (uniform) vote, cleared
      const int t = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mask));
      mask &= mask - 1;
      return t;
   };
   // the strong neighbours [a, b) of a point that has just become F: the undecided ones of its block gain one
   auto bump_row = [&](int a, int b) {
      for (int base = a; base < b; base += 64)
      {
         const int k = base + lane;
         int       m = -1, w = 1;
         if (k < b)
         {
            m = cj[k];
            if (in(m)) w = rs_load(rec, m).w;
            else m = -1;
         }
         unsigned long long mask = __ballot(m >= 0 && w == 0);
         while (mask)
         {
            const int mm = __builtin_amdgcn_readlane(m, first_set(mask));
            if (lane == 0) bump(mm);
         }
      }
   };
   // ascending-index insertion; measure-0 points become F and the points they depend on gain weight (re-listed at the tail when
   // already listed).  64 records per load.  A batch without a measure-0 point -- nearly every batch -- is a stable counting sort of its
   // rows by measure and is done by the 64 lanes at once: the rows of one measure, in ascending order, are appended to that measure's
   // list (a lane's neighbours in the list are the lanes next to it in the vote of its measure; the first takes the list's old tail).
   // The state this leaves is the one the one-by-one insertion leaves.  A batch WITH such a point goes one by one: the point may change
   // the measures of later points (the working lane then reads the rest of its batch again) and re-list earlier ones.
   for (int base = lo; base < hi; base += 64)
   {
      const int  jv    = base + lane;
      const int4 Rv    = jv < hi ? rs_load(rec, jv) : make_int4(-1, -1, 0, -3);
      const int  cnt   = min(64, hi - base);
      const bool valid = Rv.w == 0;
      if (__ballot(valid && Rv.z <= 0) == 0ull)
      {
         unsigned long long todo = __ballot(valid);
         int                prev = -1, next = -1;
         while (todo)
         {
            const int                k    = __builtin_amdgcn_readlane(Rv.z, __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo)));
            const bool               mine = valid && Rv.z == k;
            const unsigned long long m    = __ballot(mine);
            todo &= ~m;
            const int first = (int)__builtin_ctzll(m), last = 63 - (int)__builtin_clzll(m);
            const int told  = tail[k];
            if (mine)
            {
               const unsigned long long below = m & ((1ull << lane) - 1ull), above = lane == 63 ? 0ull : m & ~((2ull << lane) - 1ull);
               prev = below ? base + 63 - (int)__builtin_clzll(below) : told;
               next = above ? base + (int)__builtin_ctzll(above) : -1;
            }
            if (lane == first)
            {
               if (told >= 0) RS_F(told, 1) = base + first;
               else head[k] = base + first;
            }
            if (lane == last) tail[k] = base + last;
            if (k > maxkey) maxkey = k; // (every lane keeps the value; lane 0's is the one that is used)
         }
         if (valid) rec[jv] = make_int4(prev, next, Rv.z, 0);
         continue;
      }
      bool dirty = false; // (lane 0's)
      for (int t = 0; t < cnt; t++)
      {
         const int j = base + t;
         const int z = __builtin_amdgcn_readlane(Rv.z, t), w = __builtin_amdgcn_readlane(Rv.w, t);
         if (lane != 0) continue;
         int4 Rj = make_int4(-1, -1, z, w);
         if (dirty) Rj = rs_load(rec, j);
         if (Rj.w == -3) continue;
         if (Rj.z > 0) { enter(j, Rj.z); continue; }
         RS_F(j, 3) = -1;
         cf[j]      = -1;
         dirty      = true;
         for (int k = rp[j]; k < rp[j + 1]; k++)
         {
            const int m = cj[k];
            if (!in(m)) continue;
            const int4 R = rs_load(rec, m);
            if (R.w != 0) continue; // special, or decided (a decided point's measure is never read again)
            if (m < j)
            {
               if (R.z > 0) unlink(R);
               enter(m, R.z + 1);
            }
            else RS_F(m, 2) = R.z + 1; // not listed yet
         }
      }
   }
   for (;;)
   {
      int i = -1;
      if (lane == 0)
      {
         while (maxkey > 0 && head[maxkey] < 0) maxkey--;
         if (maxkey > 0)
         {
            i             = head[maxkey];
            const int4 Ri = rs_load(rec, i);
            unlink(Ri);
            rec[i] = make_int4(-1, -1, 0, 1);
            cf[i]  = 1;
         }
      }
      i = __builtin_amdgcn_readfirstlane(i);
      if (i < 0) break;
      const int t0 = tp[i], t1 = tp[i + 1], s0 = rp[i], s1 = rp[i + 1];
      for (int base = t0; base < t1; base += 64)
      { // everything that strongly depends on i becomes F
         const int k = base + lane;
         int       j = -1, w = 1, a = 0, b = 0;
         if (k < t1)
         {
            j = tj[k];
            if (in(j))
            {
               w = rs_load(rec, j).w;
               a = rp[j];
               b = rp[j + 1];
            }
            else j = -1;
         }
         unsigned long long mask = __ballot(j >= 0 && w == 0);
         while (mask)
         {
            const int t  = first_set(mask);
            const int jj = __builtin_amdgcn_readlane(j, t), ja = __builtin_amdgcn_readlane(a, t), jb = __builtin_amdgcn_readlane(b, t);
            if (lane == 0)
            {
               const int4 Rj = rs_load(rec, jj); // (its measure may have grown since the batch was read: an earlier F point of the batch)
               unlink(Rj);
               rec[jj] = make_int4(-1, -1, Rj.z, -1);
               cf[jj]  = -1;
            }
            bump_row(ja, jb);
         }
      }
      for (int base = s0; base < s1; base += 64)
      { // points i depends on lose one potential dependant
         const int k = base + lane;
         int       j = -1, w = 1, a = 0, b = 0;
         if (k < s1)
         {
            j = cj[k];
            if (in(j))
            {
               w = rs_load(rec, j).w;
               a = rp[j];
               b = rp[j + 1];
            }
            else j = -1;
         }
         unsigned long long mask = __ballot(j >= 0 && w == 0);
         while (mask)
         {
            const int t  = first_set(mask);
            const int jj = __builtin_amdgcn_readlane(j, t), ja = __builtin_amdgcn_readlane(a, t), jb = __builtin_amdgcn_readlane(b, t);
            int       nowF = 0;
            if (lane == 0)
            {
               const int4 Rj = rs_load(rec, jj);
               unlink(Rj);
               if (Rj.z - 1 > 0) enter(jj, Rj.z - 1);
               else
               {
                  rec[jj] = make_int4(-1, -1, 0, -1);
                  cf[jj]  = -1;
                  nowF    = 1;
               }
            }
            if (__builtin_amdgcn_readfirstlane(nowF)) bump_row(ja, jb);
         }
      }
   }
}
#undef RS_F
__global__ __launch_bounds__(256) void k_max_row_len(int n, const int *__restrict__ rp, int *mx)
{
   int m = 0; // (grid-stride: one atomic per wavefront of a SMALL grid -- atomics on one address are serialised at the L2, 10 ns each)
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = max(m, rp[i + 1] - rp[i]);
   for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
   if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(mx, m);
}
// after the first pass: only the C points of interior rows (no strong connection leaving the block) stay decided
__global__ __launch_bounds__(256) void k_hmis_keep(int n, int nblk, const int *__restrict__ part, const int *__restrict__ rp,
                                                   const int *__restrict__ cj, const unsigned char *__restrict__ sm, int *__restrict__ cf)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n || cf[i] == -3) return;
   int a = 0, b = nblk;
   while (b - a > 1)
   {
      const int m = (a + b) >> 1;
      if (part[m] <= i) a = m;
      else b = m;
   }
   const int lo = part[a], hi = part[a + 1];
   bool      boundary = false;
   for (int k = rp[i]; k < rp[i + 1] && !boundary; k++) boundary = sm[k] && (cj[k] < lo || cj[k] >= hi);
   if (boundary || cf[i] != 1) cf[i] = 0;
}
// PMIS started from a given first independent set: measures as k_pmis_init, special F and C points kept
__global__ __launch_bounds__(256) void k_pmis_init_from(int n, const int *__restrict__ indeg, unsigned long long seed, int level,
                                                        double *__restrict__ meas, int *__restrict__ cf)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int nt = indeg[i];
   meas[i]      = (double)nt + pmis_rand(seed, level, i);
   if (cf[i] == 0 && nt == 0) cf[i] = -1; // measure < 1: nobody depends on it
}
__global__ __launch_bounds__(256) void k_strong_fill(int n, const int *__restrict__ rp, const int *__restrict__ cj,
                                                     const unsigned char *__restrict__ sm, const int *__restrict__ srp, int *__restrict__ scj,
                                                     double *__restrict__ sv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int d = srp[i];
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (sm[k]) { scj[d] = cj[k]; sv[d] = 1.0; d++; }
}
// one device thread per block: about 8.5 us per row (58^3 in one block: 1.7 s, measured round 3), so a block of 2.5 M rows -- a
// 128^3 grid, where a run with the reference's CPU defaults is still a sensible parity check -- costs about 20 s.
// HDA_HMIS_MAX_ROWS moves the limit; more blocks (AmgParams::blocks) shorten the pass in proportion.
static int rs_max_rows()
{
   static const int v = 2500000;
   return v;
}

static void hmis_core(const DCsr &A, const unsigned char *smask, const int *ns, const std::vector<int> &part_in, uint64_t seed, int level,
                      int *cf)
{
   const int n = A.nrows;
   if (!n) return;
   std::vector<int> part = part_in;
   if (part.size() < 2) part = {0, n};
   const int nblk = (int)part.size() - 1;
   HDA_REQUIRE(part.front() == 0 && part.back() == n, "row blocks must cover the rows of the operator");
   int longest = 0;
   for (int q = 0; q < nblk; q++) longest = std::max(longest, part[(size_t)q + 1] - part[(size_t)q]);
   HDA_REQUIRE(longest <= rs_max_rows(), "HMIS: the Ruge first pass of a row block runs as one device thread (parity with the reference's CPU "
                                         "defaults, about 8.5 us per row); use more row blocks (HDA_BLOCKS), PMIS (coarsening type pmis), or raise "
                                         "HDA_HMIS_MAX_ROWS and wait");
   DCsr S, T;
   S.nrows = n;
   S.ncols = A.ncols;
   S.rowptr.alloc((size_t)n + 1);
   exclusive_scan(n, ns, S.rowptr.data(), nullptr);
   HDA_HIP(hipMemcpyAsync(&S.nnz, S.rowptr.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   S.col.alloc((size_t)std::max(S.nnz, 1));
   S.val.alloc((size_t)std::max(S.nnz, 1));
   k_strong_fill<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, S.rowptr.data(), S.col.data(), S.val.data());
   transpose(S, T);
   DArray<int> dpart;
   dpart.upload(part.data(), part.size());
   {
      // a measure never exceeds twice the in-block dependants: bucket keys stay below 2 * (largest in-degree) + 2
      DArray<int> mx(1);
      mx.zero();
      k_max_row_len<<<std::min(ceil_div(n, 256), 2048), 256, 0, STREAM>>>(n, T.rowptr.data(), mx.data());
      int maxin = 0;
      mx.download(&maxin, 1);
      const int         kcap = 2 * maxin + 2;
      const char       *le   = getenv("HDA_RS_LDS"); // 0: bucket heads / tails in global memory whatever their number (tests: the form for in-degrees beyond 4095)
      const bool        lds  = (size_t)kcap * 8 <= 64 * 1024 && !(le && atoi(le) == 0);
      const size_t      nbk  = lds ? 1 : 2 * (size_t)n + 2 * (size_t)nblk;
      DArray<int>       head(nbk), tail(nbk);
      DArray<long long> recs(2 * ((size_t)n + 1)); // int4 records
      int4             *rec = reinterpret_cast<int4 *>(recs.data());
      if (lds)
         k_rs_first_pass_rec<true><<<nblk, 64, (size_t)kcap * 8, STREAM>>>(nblk, dpart.data(), S.rowptr.data(), S.col.data(), T.rowptr.data(), T.col.data(),
                                                                          head.data(), tail.data(), rec, cf, kcap);
      else
         k_rs_first_pass_rec<false><<<nblk, 64, 0, STREAM>>>(nblk, dpart.data(), S.rowptr.data(), S.col.data(), T.rowptr.data(), T.col.data(), head.data(),
                                                            tail.data(), rec, cf, kcap);
      Context::get().sync();
   }
   // hypre_BoomerAMGCoarsenPMIS with CF_init 1: interior C points are the first independent set, everything else is decided again
   const int g = ceil_div(n, 256);
   k_hmis_keep<<<g, 256, 0, STREAM>>>(n, nblk, dpart.data(), A.rowptr.data(), A.col.data(), smask, cf);
   DArray<int>           indeg((size_t)n), counter(1);
   DArray<double>        meas((size_t)n);
   DArray<unsigned char> notmax((size_t)n);
   indeg.zero();
   notmax.zero();
   if (A.nnz) k_indeg<<<std::min(ceil_div(A.nnz, 256), 1 << 16), 256, 0, STREAM>>>(A.nnz, A.col.data(), smask, indeg.data());
   k_pmis_init_from<<<g, 256, 0, STREAM>>>(n, indeg.data(), seed, level, meas.data(), cf);
   counter.zero();
   k_pmis_setF<<<std::min(g, kPmisGrid), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, counter.data(), 0); // dependants of the kept C points
   int left = 0, rounds = 0;
   counter.download(&left, 1);
   while (left > 0)
   {
      counter.zero();
      k_pmis_mark<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, meas.data(), notmax.data(), 0);
      k_pmis_setC<<<g, 256, 0, STREAM>>>(n, cf, notmax.data(), 0);
      k_pmis_setF<<<std::min(g, kPmisGrid), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, counter.data(), 0);
      counter.download(&left, 1);
      HDA_REQUIRE(++rounds < 10000, "HMIS: the trailing PMIS did not terminate");
   }
}

void amg_hmis(const DCsr &A, const unsigned char *smask, const std::vector<int> &part, uint64_t seed, int level, int *cf)
{
   DArray<int> ns((size_t)A.nrows + 1);
   if (A.nrows) k_count_strong<<<ceil_div(A.nrows, 256), 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), smask, ns.data());
   hmis_core(A, smask, ns.data(), part, seed, level, cf);
}

// ------------------------------------------------------------ interpolation

__device__ __forceinline__ unsigned hash_slot(int key, int lg)
{
   return (lg == 0) ? 0u : (((unsigned)key * 2654435761u) >> (32 - lg));
}
__device__ __forceinline__ int pow2ceil_dev(int x)
{
   int p = 1;
   while (p < x) p <<= 1;
   return p;
}

__global__ __launch_bounds__(256) void k_count_strongC(int n, const int *__restrict__ rp,
                                                       const int *__restrict__ cj,
                                                       const unsigned char *__restrict__ smask,
                                                       const int *__restrict__ cf, int *__restrict__ nsC)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int c = 0;
   for (int k = rp[i]; k < rp[i + 1]; k++) c += (smask[k] && cf[cj[k]] == 1);
   nsC[i] = c;
}

// per entry: strong connection to a C point (what C-hat_i is built from); the interpolation kernel reads it with the entry
__global__ __launch_bounds__(256) void k_strongC_flag(int nnz, const int *__restrict__ cj, const unsigned char *__restrict__ smask,
                                                      const int *__restrict__ cf, unsigned char *__restrict__ sc)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256) sc[k] = (smask[k] && cf[cj[k]] == 1) ? 1 : 0;
}

// the strong C columns of every row, side by side in row order (offsets: scan of k_count_strongC's counts): what a row's candidate
// set C-hat_i is gathered from without walking its neighbours' full rows
__global__ __launch_bounds__(256) void k_fill_strongC(int n, const int *__restrict__ rp, const int *__restrict__ cj, const unsigned char *__restrict__ smask,
                                                      const int *__restrict__ cf, const int *__restrict__ scofs, int *__restrict__ scc)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int q = scofs[i];
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      const int j = cj[k];
      if (smask[k] && cf[j] == 1) scc[q++] = j;
   }
}

__global__ __launch_bounds__(256) void k_interp_ub(int n, const int *__restrict__ rp,
                                                   const int *__restrict__ cj,
                                                   const unsigned char *__restrict__ smask,
                                                   const int *__restrict__ cf, const int *__restrict__ nsC,
                                                   int *__restrict__ ub, int *__restrict__ hsz,
                                                   int *__restrict__ cmark, int *__restrict__ nt)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int c = cf[i];
   cmark[i]    = (c == 1);
   int u = 0, h = 0, t = 0;
   if (c == 1) u = 1;
   else if (c == -1)
   {
      for (int k = rp[i]; k < rp[i + 1]; k++)
      {
         if (!smask[k]) continue;
         const int j  = cj[k];
         const int cj_ = cf[j];
         if (cj_ == 1) u += 1;
         else if (cj_ == -1) { u += nsC[j]; t += rp[j + 1] - rp[j]; }
      }
      h = u ? max(8, pow2ceil_dev(2 * u)) : 0;
   }
   ub[i]  = u;
   hsz[i] = h;
   nt[i]  = t;
}

struct PEnt {
   int    c;
   double w;
};

// Descending-|w| quicksort in the K&R form (pivot = middle element swapped to the front,
// strict '>' partition) -- the tie order it produces decides which pmax of several equal
// weights survive truncation, so it is part of the algorithm's definition (see oracle).
// stack: 64 ints of the caller's (the smaller partition is finished first, so the depth stays below 2 log2 n)
__device__ void qsort_abs(int *L, double *W, int n, int *stack)
{
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

__global__ __launch_bounds__(256) void k_interp_build(
   int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
   const unsigned char *__restrict__ smask, const int *__restrict__ cf,
   const long long *__restrict__ uofs, const long long *__restrict__ hofs, int *__restrict__ lcol,
   double *__restrict__ lw, int *__restrict__ htab, int pmax, double trunc_factor, int *__restrict__ pcnt,
   const unsigned char *__restrict__ rowmode, const int *__restrict__ dof, int itype)
{ // itype 6: extended+i; 3: direct interpolation with separation of weights (strong C neighbours only); 8: standard
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   if (rowmode && !rowmode[i]) return; // handled by the wave-per-row kernel
   const int c = cf[i];
   int      *L = lcol + uofs[i];
   double   *W = lw + uofs[i];
   if (c == 1)
   {
      L[0]    = i;
      W[0]    = 1.0;
      pcnt[i] = 1;
      return;
   }
   if (c != -1)
   {
      pcnt[i] = 0;
      return;
   }
   int      *H    = htab + hofs[i];
   const int hs   = (int)(hofs[i + 1] - hofs[i]);
   const int mask = hs - 1;
   int       lg   = 0;
   while ((1 << lg) < hs) lg++;
   int cnt = 0;
   auto find = [&](int m) -> int {
      if (hs == 0) return -1;
      unsigned h = hash_slot(m, lg);
      for (;;)
      {
         const int e = H[h];
         if (e < 0) return -1;
         if (L[e] == m) return e;
         h = (h + 1) & mask;
      }
   };
   auto add = [&](int m) {
      unsigned h = hash_slot(m, lg);
      for (;;)
      {
         const int e = H[h];
         if (e < 0)
         {
            H[h]   = cnt;
            L[cnt] = m;
            W[cnt] = 0.0;
            cnt++;
            return;
         }
         if (L[e] == m) return;
         h = (h + 1) & mask;
      }
   };
   const int k0 = rp[i], k1 = rp[i + 1];
   // C-hat_i = C_i U (U_{j in F_i^s} C_j), discovery order
   for (int k = k0; k < k1; k++)
   {
      if (!smask[k]) continue;
      const int j = cj[k];
      if (cf[j] == 1) add(j);
      else if (cf[j] == -1 && itype != 3)
         for (int kk = rp[j]; kk < rp[j + 1]; kk++)
         {
            const int m = cj[kk];
            if (smask[kk] && cf[m] == 1) add(m);
         }
   }
   double diagonal = 0.0;
   for (int k = k0; k < k1; k++)
      if (cj[k] == i) diagonal = v[k];
   if (itype == 3)
   { // the oracle's orc_interp_direct_dof, entry for entry
      double sum_N_pos = 0.0, sum_N_neg = 0.0, sum_P_pos = 0.0, sum_P_neg = 0.0;
      for (int k = k0; k < k1; k++)
      {
         const int j = cj[k];
         if (j == i) continue;
         const double a = v[k];
         if (!(dof && dof[j] != dof[i]))
         {
            if (a > 0.0) sum_N_pos += a;
            else sum_N_neg += a;
         }
         if (smask[k] && cf[j] == 1)
         {
            W[find(j)] = a; // (a strong C neighbour occurs once in the row)
            if (a > 0.0) sum_P_pos += a;
            else sum_P_neg += a;
         }
      }
      double alfa = 1.0, beta = 1.0;
      if (sum_P_neg != 0.0) alfa = sum_N_neg / sum_P_neg / diagonal;
      if (sum_P_pos != 0.0) beta = sum_N_pos / sum_P_pos / diagonal;
      for (int q = 0; q < cnt; q++) W[q] *= (W[q] > 0.0) ? -beta : -alfa;
      diagonal = 0.0; // nothing left to divide by below
   }
   else if (itype == 8)
   { // the oracle's orc_interp_standard_dof, entry for entry: strong F neighbours eliminated through their own rows, then direct
     // interpolation on the widened stencil (no separation of weights)
      double other = 0.0;
      for (int k = k0; k < k1; k++)
      {
         const int j = cj[k];
         if (j == i) continue;
         const double aij = v[k];
         if (smask[k] && cf[j] == -1 && !(dof && dof[j] != dof[i]))
         {
            const int j0 = rp[j], j1 = rp[j + 1];
            double    ajj = 0.0;
            for (int kk = j0; kk < j1; kk++)
               if (cj[kk] == j) ajj = v[kk];
            const double distribute = aij / ajj;
            for (int kk = j0; kk < j1; kk++)
            {
               const int m = cj[kk];
               if (m == j) continue;
               const double t  = v[kk] * distribute;
               const int    em = find(m);
               if (em >= 0) W[em] -= t;
               else if (m == i) diagonal -= t;
               else other -= t;
            }
         }
         else
         {
            const int e = find(j);
            if (e >= 0) W[e] += aij;
            else other += aij;
         }
      }
      double sum_C = 0.0, alfa = 1.0;
      for (int q = 0; q < cnt; q++) sum_C += W[q];
      const double sum = sum_C + other;
      if (sum_C * diagonal != 0.0) alfa = sum / sum_C / diagonal;
      for (int q = 0; q < cnt; q++) W[q] = -alfa * W[q];
      diagonal = 0.0;
   }
   else
   for (int k = k0; k < k1; k++)
   {
      const int j = cj[k];
      if (j == i) continue;
      const double aij = v[k];
      const int    e   = find(j);
      if (e >= 0) W[e] += aij;
      else if (smask[k] && cf[j] == -1)
      {
         const int j0 = rp[j], j1 = rp[j + 1];
         double    ajj = 0.0, sum = 0.0;
         for (int kk = j0; kk < j1; kk++)
            if (cj[kk] == j) ajj = v[kk];
         const double sgn = (ajj < 0.0) ? -1.0 : 1.0;
         for (int kk = j0; kk < j1; kk++)
         {
            const int m = cj[kk];
            if (sgn * v[kk] < 0.0 && (m == i || find(m) >= 0)) sum += v[kk];
         }
         if (sum != 0.0)
         {
            const double distribute = aij / sum;
            for (int kk = j0; kk < j1; kk++)
            {
               const int m = cj[kk];
               if (sgn * v[kk] < 0.0)
               {
                  const int em = find(m);
                  if (em >= 0) W[em] += distribute * v[kk];
                  else if (m == i) diagonal += distribute * v[kk];
               }
            }
         }
         else
            diagonal += aij;
      }
      else if (cf[j] != -3 && !(dof && dof[j] != dof[i]))
         diagonal += aij; // weak connection lumped into the diagonal (same function only)
   }
   if (diagonal != 0.0)
      for (int q = 0; q < cnt; q++) W[q] = W[q] / (-diagonal);
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
      qsort_abs(L, W, cnt, stk);
      cnt = pmax;
      for (int a = 1; a < cnt; a++) // kept set -> column order before summing (see oracle)
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
   // storage order: ascending (fine == coarse) column
   for (int a = 1; a < cnt; a++)
   {
      const int    cc = L[a];
      const double ww = W[a];
      int          b  = a - 1;
      while (b >= 0 && L[b] > cc)
      {
         L[b + 1] = L[b];
         W[b + 1] = W[b];
         b--;
      }
      L[b + 1] = cc;
      W[b + 1] = ww;
   }
   pcnt[i] = cnt;
}

// ---- wave-per-row extended+i interpolation ------------------------------------------------
// Same arithmetic, in the same order, as k_interp_build (and the oracle), but one wavefront
// owns a row: the row, the candidate set C-hat_i (discovery order) and the weight
// accumulators live in LDS; strong-F neighbour rows are read 64 entries at a time
// (coalesced).  Order-sensitive sums stay sequential: the per-neighbour sum walks the
// ballot bits in ascending kk, and within one neighbour row every target occurs once, so the
// 64 lanes of a batch update distinct accumulators while batches and neighbours are visited
// in order.  Rows that do not fit the LDS budget are flagged for k_interp_build.
__host__ __device__ inline size_t interp_wave_doubles(int cap_row, int cap_ub, int cap_nbr)
{ // doubles: rval[cap_row] Wv[cap_ub] nval[cap_nbr] misc[2]; ints: 6*cap_row + 2 + 4*cap_ub + cap_nbr
   return (size_t)cap_row + cap_ub + cap_nbr + 2 + ((size_t)cap_row * 6 + 2 + (size_t)cap_ub * 4 + cap_nbr + 1) / 2 + 1;
}
#define WAVE_SYNC()                                         \
   do {                                                     \
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
      __builtin_amdgcn_wave_barrier();                      \
   } while (0)

__global__ __launch_bounds__(256) void k_interp_rowmode(int n, const int *__restrict__ rp, const int *__restrict__ ub,
                                                        const int *__restrict__ nt, int cap_row, int cap_ub, int cap_nbr,
                                                        unsigned char *__restrict__ rowmode, int *__restrict__ hsz)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   (void)nt; (void)cap_nbr; // long neighbour lists are read from global memory by the wave kernel
   const bool big = (rp[i + 1] - rp[i]) > cap_row || ub[i] > cap_ub;
   rowmode[i]     = big ? 1 : 0;
   if (!big) hsz[i] = 0; // no global hash table for rows the wave kernel handles
}
__global__ __launch_bounds__(256) void k_max3(int n, const int *__restrict__ rp, const int *__restrict__ ub, const int *__restrict__ nt, int *mx)
{
   int a = 0, b = 0, c = 0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
   {
      a = max(a, rp[i + 1] - rp[i]);
      b = max(b, ub[i]);
      c = max(c, nt[i]);
   }
   for (int o = 32; o > 0; o >>= 1) { a = max(a, __shfl_xor(a, o)); b = max(b, __shfl_xor(b, o)); c = max(c, __shfl_xor(c, o)); }
   if ((threadIdx.x & 63) == 0) { atomicMax(&mx[0], a); atomicMax(&mx[1], b); atomicMax(&mx[2], c); }
}

// value of lane l of the row's lane group.  With a whole wavefront per row l is wave-uniform (it comes out of a ballot), so the value
// travels through two v_readlane instead of two ds_bpermute round trips: the ordered sums below read one lane per term.
template <int G>
__device__ __forceinline__ double group_lane_value(double v, int l)
{
   if constexpr (G == 64)
   {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
      return __hiloint2double(hi, lo);
   }
   else return __shfl(v, l, G);
}

template <int G>
__global__ __launch_bounds__(256, 5) void k_interp_wave(
   int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
   const unsigned char *__restrict__ smask, const unsigned char *__restrict__ sc, const int *__restrict__ cf, const int *__restrict__ nsC,
   const long long *__restrict__ uofs, int cap_row, int cap_ub, int cap_nbr, int pmax, double trunc_factor,
   const unsigned char *__restrict__ rowmode, int *__restrict__ lcol, double *__restrict__ lw, int *__restrict__ pcnt,
   const int *__restrict__ dof, int s3_scan, unsigned long long *__restrict__ prof, int reg_nbr, const int *__restrict__ scofs,
   const int *__restrict__ scc)
{
   extern __shared__ double ilds[];
   constexpr int GPB = 256 / G;                    // row groups per workgroup
   const int     grp = threadIdx.x / G, lane = threadIdx.x & (G - 1); // lane: position inside the row's group of G lanes
   const int     gb  = (threadIdx.x & 63) & ~(G - 1); // first lane of the group inside its wavefront
   const unsigned long long gmask = ~0ULL >> (64 - G);
   auto gballot = [&](bool pr) -> unsigned long long { return (__ballot(pr) >> gb) & gmask; }; // votes of the group's lanes
   const size_t per  = interp_wave_doubles(cap_row, cap_ub, cap_nbr);
   double      *rval = ilds + grp * per, *Wv = rval + cap_row, *nval = Wv + cap_ub, *misc = nval + cap_nbr;
   int         *rcol = (int *)(misc + 2), *rofs = rcol + cap_row, *rtype = rofs + cap_row + 1; // rofs: cap_row + 1
   int         *noff = rtype + cap_row, *nbeg = noff + cap_row + 1;                         // neighbour-row staging: offsets, global starts
   double      *ksum = (double *)rcol; // per-neighbour sums of a staged row: over rcol and rofs, both dead by then
   int         *craw = nbeg + cap_row, *ucol = craw + cap_ub, *htb = ucol + cap_ub;         // htb: 2*cap_ub slots, col -> pos+1
   int         *ncol = htb + 2 * cap_ub;                                                    // ncol: cap_nbr (bit 31 = strong C entry)
   int         *sbeg = ncol + cap_nbr;                                                      // sbeg: cap_row, start of a strong-F neighbour's strong-C list
   const int    hmask = 2 * cap_ub - 1;
   auto lookup = [&](int m) -> int { // position of column m in C-hat_i or -1
      unsigned h = ((unsigned)m * 2654435761u) & (unsigned)hmask;
      for (;;)
      {
         const int e = htb[h];
         if (e == 0) return -1;
         if (ucol[e - 1] == m) return e - 1;
         h = (h + 1) & (unsigned)hmask;
      }
   };
   const unsigned long long lt = (1ULL << lane) - 1; // the group's lanes below this one
   enum { T_DIAG = 0, T_SC = 1, T_SF = 2, T_OTHER = 3 };
   // HDA_INTERP_PROF: shader-clock cycles per stage, summed over rows by every group's first lane (diagnostics)
   unsigned long long tprev = 0;
   auto stamp = [&](int st) {
      if (prof)
      {
         const unsigned long long t = __builtin_readcyclecounter();
         if (lane == 0 && st >= 0) atomicAdd(&prof[st], t - tprev);
         tprev = t;
      }
   };
   // XCD-aware dealing of the rows (round 5): workgroups b, b + 8, ... share an XCD and its L2; they walk ONE contiguous eighth of the rows
   // instead of every eighth group of four.  A row reads the rows of its strong F neighbours -- the level-1 launch at 256^3 fetched 58 GB for
   // 2.2 GB of operator (profiles/r05f_setup_accounting.md) -- and rows that share neighbours (the same grid line, the next one) now meet in
   // the same L2.  Rows are independent of one another: same results.  (xcd_rows = 0: the plain grid-stride walk, for grids below 8.)
   // Same-box A/B at 256^3 (tools/gpurun/r05_interp_ab.sh): level 1 fetches 26 % less (28.1 -> 20.7 M KB) in the same 39.5 ms -- that launch is
   // bound by its LDS / instruction work, not by the fetches --, level 0 runs 20.7 -> 18.4 ms; setup 242.7 -> 238.3 ms.
   const int  nslots   = (int)gridDim.x >> 3;
   const int  xcd_rows = (gridDim.x >= 8 && (gridDim.x & 7) == 0) ? (((n + 7) / 8 + GPB - 1) / GPB) * GPB : 0;
   const int  r_first  = xcd_rows ? ((int)blockIdx.x & 7) * xcd_rows + ((int)blockIdx.x >> 3) * GPB + grp : (int)blockIdx.x * GPB + grp;
   const int  r_end    = xcd_rows ? min(n, (((int)blockIdx.x & 7) + 1) * xcd_rows) : n;
   const int  r_step   = xcd_rows ? nslots * GPB : (int)gridDim.x * GPB;
   for (int i = r_first; i < r_end; i += r_step)
   {
      if (rowmode[i]) continue;
      const int c = cf[i];
      if (c == 1)
      {
         if (lane == 0) { lcol[uofs[i]] = i; lw[uofs[i]] = 1.0; pcnt[i] = 1; }
         continue;
      }
      if (c != -1)
      {
         if (lane == 0) pcnt[i] = 0;
         continue;
      }
      stamp(-1);
      const int k0 = rp[i], nk = rp[i + 1] - k0;
      // ---- 1. stage the row, classify entries, candidate offsets per entry
      int running = 0;
      for (int base = 0; base < nk; base += G)
      {
         const int k = base + lane;
         int       cntk = 0;
         if (k < nk)
         {
            const int j  = cj[k0 + k];
            const int cfj = cf[j];
            const bool st = smask[k0 + k] != 0;
            int t = T_OTHER;
            if (j == i) t = T_DIAG;
            else if (st && cfj == 1) { t = T_SC; cntk = 1; }
            else if (st && cfj == -1)
            { // the neighbour row's extent travels with its candidate count: one round trip to memory, not two
               t       = T_SF;
               cntk    = nsC[j];
               nbeg[k] = rp[j];
               noff[k] = rp[j + 1]; // (its end, until the scan of stage 1b turns it into an offset)
               if (scc) sbeg[k] = scofs[j];
            }
            rcol[k]  = j;
            rval[k]  = v[k0 + k];
            // bit 8: never lumped into the diagonal (special F point, or another function's unknown)
            rtype[k] = t | ((cfj == -3 || (dof && dof[j] != dof[i])) ? 8 : 0) | (cfj == 1 ? 16 : 0);
         }
         int incl = cntk; // inclusive scan over the wave
         for (int o = 1; o < G; o <<= 1)
         {
            const int up = __shfl_up(incl, o, G);
            if (lane >= o) incl += up;
         }
         if (k < nk) rofs[k] = running + incl - cntk;
         running += __shfl(incl, G - 1, G);
      }
      const int ncand = running;
      WAVE_SYNC();
      stamp(0);
      // ---- 1b. stage the strong-F neighbour rows in LDS (one flat, fully parallel copy)
      int nrun = 0;
      for (int base = 0; base < nk; base += G)
      {
         const int k  = base + lane;
         int       ln = 0, js = 0;
         if (k < nk && (rtype[k] & 7) == T_SF)
         {
            js = nbeg[k];
            ln = noff[k] - js;
         }
         int incl = ln;
         for (int o = 1; o < G; o <<= 1)
         {
            const int up = __shfl_up(incl, o, G);
            if (lane >= o) incl += up;
         }
         if (k < nk) { noff[k] = nrun + incl - ln; nbeg[k] = js; }
         nrun += __shfl(incl, G - 1, G);
      }
      const int  nbr_total = nrun;
      const bool staged    = nbr_total <= cap_nbr; // otherwise neighbour rows are read from global memory
      if (lane == 0) noff[nk] = nbr_total;
      WAVE_SYNC();
      if (staged)
      {
         for (int fb = 0; fb < nbr_total; fb += 4 * G)
         { // four independent batches in flight
            int g[4], m[4];
            double a[4];
            unsigned char sm[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
               const int f = fb + G * u + lane;
               g[u]        = -1;
               if (f < nbr_total)
               {
                  int lo = 0, hi = nk - 1; // last k with noff[k] <= f (zero-length entries share offsets)
                  while (lo < hi)
                  {
                     const int mid = (lo + hi + 1) >> 1;
                     if (noff[mid] <= f) lo = mid;
                     else hi = mid - 1;
                  }
                  g[u] = nbeg[lo] + (f - noff[lo]);
               }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
               if (g[u] >= 0) { m[u] = cj[g[u]]; a[u] = v[g[u]]; sm[u] = sc[g[u]]; } // sc: strong entry whose column is a C point
#pragma unroll
            for (int u = 0; u < 4; u++)
               if (g[u] >= 0)
               {
                  const int f = fb + G * u + lane;
                  nval[f]     = a[u];
                  ncol[f]     = m[u] | (sm[u] ? (int)0x80000000 : 0);
               }
         }
      }
      WAVE_SYNC();
      stamp(1);
      // neighbour-row entry e of neighbour k (e relative to the row): column, value, strong-C flag
      auto nb_col = [&](int k, int e) -> int { return staged ? (ncol[noff[k] + e] & 0x7FFFFFFF) : cj[nbeg[k] + e]; };
      auto nb_val = [&](int k, int e) -> double { return staged ? nval[noff[k] + e] : v[nbeg[k] + e]; };
      auto nb_sc  = [&](int k, int e) -> bool {
         if (staged) return ncol[noff[k] + e] < 0;
         return sc[nbeg[k] + e] != 0;
      };
      // ---- 2. candidates in discovery order: entry k's candidates start at rofs[k]
      if (staged)
      { // one lane per row entry walks its staged neighbour row
         for (int k = lane; k < nk; k += G)
         {
            const int t = rtype[k] & 7;
            if (t == T_SC) craw[rofs[k]] = rcol[k];
            else if (t == T_SF)
            {
               int q = rofs[k];
               for (int f = noff[k]; f < noff[k + 1]; f++)
               {
                  const int w = ncol[f];
                  if (w < 0) craw[q++] = w & 0x7FFFFFFF;
               }
            }
         }
      }
      else if (scc)
      { // one flat gather: candidate f belongs to the last row entry whose offset is <= f (entries without candidates share offsets);
        // a strong C neighbour is its own candidate, a strong F neighbour contributes its list of strong C columns
         for (int f = lane; f < ncand; f += G)
         {
            int lo = 0, hi = nk - 1;
            while (lo < hi)
            {
               const int mid = (lo + hi + 1) >> 1;
               if (rofs[mid] <= f) lo = mid;
               else hi = mid - 1;
            }
            craw[f] = ((rtype[lo] & 7) == T_SC) ? rcol[lo] : scc[sbeg[lo] + (f - rofs[lo])];
         }
      }
      else
      for (int k = 0; k < nk; k++)
      {
         const int t = rtype[k] & 7;
         if (t == T_SC) { if (lane == 0) craw[rofs[k]] = rcol[k]; }
         else if (t == T_SF)
         {
            const int nj = noff[k + 1] - noff[k];
            int       q = rofs[k];
            for (int base = 0; base < nj; base += G)
            {
               const int kk = base + lane;
               bool      f  = false;
               int       m  = 0;
               if (kk < nj)
               {
                  m = nb_col(k, kk);
                  f = nb_sc(k, kk);
               }
               const unsigned long long b = gballot(f);
               if (f) craw[q + __popcll(b & lt)] = m;
               q += __popcll(b);
            }
         }
      }
      WAVE_SYNC();
      stamp(2);
      // ---- 3. first occurrences -> C-hat_i in discovery order (ucol), accumulators to zero.
      // A hash table over the candidates keeps, per column, the smallest candidate index (+1); a candidate is a
      // first occurrence when that index is its own.
      int cnt = 0;
      if (ncand <= s3_scan)
      { // few candidates: every lane scans the ones before its own
         for (int base = 0; base < ncand; base += G)
         {
            const int ci = base + lane;
            bool      uq = false;
            int       m  = 0;
            if (ci < ncand)
            {
               m  = craw[ci];
               uq = true;
               for (int e = 0; e < ci; e++)
                  if (craw[e] == m) { uq = false; break; }
            }
            const unsigned long long b = gballot(uq);
            if (uq)
            {
               const int pos = cnt + __popcll(b & lt);
               ucol[pos]     = m;
               Wv[pos]       = 0.0;
            }
            cnt += __popcll(b);
         }
      }
      else
      {
      for (int q = lane; q <= hmask; q += G) htb[q] = 0;
      WAVE_SYNC();
      for (int ci = lane; ci < ncand; ci += G)
      {
         const int m = craw[ci];
         unsigned  h = ((unsigned)m * 2654435761u) & (unsigned)hmask;
         for (;;)
         {
            int e = htb[h];
            if (e == 0)
            {
               e = atomicCAS(&htb[h], 0, ci + 1);
               if (e == 0) break;
            }
            if (craw[e - 1] == m) // the slot is this column's (its index only ever moves to an equal column)
            {
               atomicMin(&htb[h], ci + 1);
               break;
            }
            h = (h + 1) & (unsigned)hmask;
         }
      }
      WAVE_SYNC();
      for (int base = 0; base < ncand; base += G)
      {
         const int ci = base + lane;
         bool      uq = false;
         int       m  = 0;
         if (ci < ncand)
         {
            m          = craw[ci];
            unsigned h = ((unsigned)m * 2654435761u) & (unsigned)hmask;
            int      e = htb[h];
            while (craw[e - 1] != m)
            {
               h = (h + 1) & (unsigned)hmask;
               e = htb[h];
            }
            uq = (e == ci + 1);
         }
         const unsigned long long b = gballot(uq);
         if (uq)
         {
            const int pos = cnt + __popcll(b & lt);
            ucol[pos]     = m;
            Wv[pos]       = 0.0;
         }
         cnt += __popcll(b);
      }
      }
      WAVE_SYNC();
      for (int q = lane; q <= hmask; q += G) htb[q] = 0;
      WAVE_SYNC();
      for (int q = lane; q < cnt; q += G)
      { // set semantics: any insertion order gives the same lookups
         unsigned h = ((unsigned)ucol[q] * 2654435761u) & (unsigned)hmask;
         while (atomicCAS(&htb[h], 0, q + 1) != 0) h = (h + 1) & (unsigned)hmask;
      }
      WAVE_SYNC();
      stamp(3);
      // ---- 4. weights, neighbours visited in ascending k
      if (lane == 0)
      {
         double d = 0.0;
         for (int k = 0; k < nk; k++)
            if ((rtype[k] & 7) == T_DIAG) d = rval[k];
         misc[0] = d;
      }
      WAVE_SYNC();
      if (staged)
      {
         // 4a. one lane per row entry: sign of a_jj for a strong-F neighbour (bit 5 of rtype), accumulator of a C point
         for (int k = lane; k < nk; k += G)
         {
            const int t = rtype[k];
            const int j = rcol[k];
            if ((t & 7) == T_SF)
            {
               double ajj = 0.0;
               for (int f = noff[k]; f < noff[k + 1]; f++)
                  if ((ncol[f] & 0x7FFFFFFF) == j) { ajj = nval[f]; break; }
               if (ajj < 0.0) rtype[k] = t | 32;
            }
            nbeg[k] = (t & 16) ? lookup(j) : -1; // the global row starts are not needed once the rows are staged
         }
         WAVE_SYNC();
         // 4b. every staged entry: column -> accumulator position, -2 for column i, -1 for neither
         for (int f = lane; f < nbr_total; f += G)
         {
            const int m = ncol[f] & 0x7FFFFFFF;
            ncol[f]     = (m == i) ? -2 : lookup(m);
         }
         WAVE_SYNC();
         // 4c. one lane per strong-F neighbour: the ordered sum over the qualifying entries of its row
         for (int k = lane; k < nk; k += G)
            if ((rtype[k] & 7) == T_SF)
            {
               const double sgn = (rtype[k] & 32) ? -1.0 : 1.0;
               double       sum = 0.0;
               for (int f = noff[k]; f < noff[k + 1]; f++)
               {
                  const double a = nval[f];
                  if (sgn * a < 0.0 && ncol[f] != -1) sum += a;
               }
               ksum[k] = sum;
            }
         WAVE_SYNC();
         stamp(4);
         // 4d. accumulation, neighbours in ascending k (the order every accumulator sees its terms in)
         for (int k = 0; k < nk; k++)
         {
            const int    t = rtype[k];
            const double aij = rval[k];
            if ((t & 7) == T_DIAG) continue;
            if (t & 16)
            {
               const int pos = nbeg[k];
               if (pos >= 0)
               {
                  if (lane == 0) Wv[pos] += aij;
                  WAVE_SYNC();
                  continue;
               }
            }
            if ((t & 7) == T_SF)
            {
               const double sum = ksum[k];
               if (sum != 0.0)
               {
                  const double distribute = aij / sum;
                  const double sgn        = (t & 32) ? -1.0 : 1.0;
                  const int    f1         = noff[k + 1];
                  for (int f = noff[k] + lane; f - lane < f1; f += G)
                  {
                     if (f < f1)
                     {
                        const double a = nval[f];
                        if (sgn * a < 0.0)
                        {
                           const int pos = ncol[f];
                           if (pos >= 0) Wv[pos] += distribute * a;
                           else if (pos == -2) misc[0] += distribute * a;
                        }
                     }
                     WAVE_SYNC();
                  }
               }
               else
               {
                  if (lane == 0) misc[0] += aij;
                  WAVE_SYNC();
               }
               continue;
            }
            if (!(t & 8))
            {
               if (lane == 0) misc[0] += aij;
               WAVE_SYNC();
            }
         }
      }
      else
      for (int k = 0; k < nk; k++)
      {
         const int    t = rtype[k];
         const int    j = rcol[k];
         const double aij = rval[k];
         if ((t & 7) == T_DIAG) continue;
         if (t & 16)
         { // a C point: member of C-hat_i?
            const int pos = lookup(j);
            if (pos >= 0)
            {
               if (lane == 0) Wv[pos] += aij;
               WAVE_SYNC();
               continue;
            }
         }
         if (reg_nbr && (t & 7) == T_SF && noff[k + 1] - noff[k] <= 4 * G)
         { // the neighbour row fits four entries per lane: ONE pass over memory and one lookup per entry, the three sweeps below
           // (a_jj, ordered sum, distribution) run on registers.  Same terms, same order as the general form that follows.
            const int nj = noff[k + 1] - noff[k];
            int       mm[4], pp[4];
            double    aa[4];
            double    ajj = 0.0;
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
               const int kk = u * G + lane;
               mm[u]        = -1;
               aa[u]        = 0.0;
               if (u * G < nj)
               {
                  if (kk < nj) { mm[u] = nb_col(k, kk); aa[u] = nb_val(k, kk); }
                  const unsigned long long b = gballot(kk < nj && mm[u] == j);
                  if (b) ajj = group_lane_value<G>(aa[u], __ffsll((long long)b) - 1);
               }
            }
            const double sgn = (ajj < 0.0) ? -1.0 : 1.0;
            double       sum = 0.0;
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
               pp[u] = -1;
               if (u * G < nj)
               {
                  bool cond = false;
                  if (mm[u] >= 0 && sgn * aa[u] < 0.0)
                  {
                     pp[u] = (mm[u] == i) ? -2 : lookup(mm[u]);
                     cond  = pp[u] != -1;
                  }
                  unsigned long long bits = gballot(cond);
                  while (bits)
                  {
                     const int l = __ffsll((long long)bits) - 1;
                     sum += group_lane_value<G>(aa[u], l);
                     bits &= bits - 1;
                  }
               }
            }
            if (sum != 0.0)
            {
               const double distribute = aij / sum;
#pragma unroll
               for (int u = 0; u < 4; u++)
                  if (u * G < nj)
                  {
                     if (pp[u] >= 0) Wv[pp[u]] += distribute * aa[u];
                     else if (pp[u] == -2) misc[0] += distribute * aa[u];
                     WAVE_SYNC();
                  }
            }
            else
            {
               if (lane == 0) misc[0] += aij;
               WAVE_SYNC();
            }
            continue;
         }
         if ((t & 7) == T_SF)
         {
            const int nj = noff[k + 1] - noff[k];
            // a_jj and its sign
            double ajj = 0.0;
            for (int base = 0; base < nj; base += G)
            {
               const int kk = base + lane;
               const bool hit = kk < nj && nb_col(k, kk) == j;
               const unsigned long long b = gballot(hit);
               const double             a = hit ? nb_val(k, kk) : 0.0;
               if (b) ajj = group_lane_value<G>(a, __ffsll((long long)b) - 1);
            }
            const double sgn = (ajj < 0.0) ? -1.0 : 1.0;
            // ordered sum over the qualifying entries of row j
            double sum = 0.0;
            for (int base = 0; base < nj; base += G)
            {
               const int kk = base + lane;
               bool      cond = false;
               double    a    = 0.0;
               if (kk < nj)
               {
                  const int m = nb_col(k, kk);
                  a           = nb_val(k, kk);
                  if (sgn * a < 0.0) cond = (m == i) || lookup(m) >= 0;
               }
               unsigned long long bits = gballot(cond);
               while (bits)
               {
                  const int l = __ffsll((long long)bits) - 1;
                  sum += group_lane_value<G>(a, l);
                  bits &= bits - 1;
               }
            }
            if (sum != 0.0)
            {
               const double distribute = aij / sum;
               for (int base = 0; base < nj; base += G)
               {
                  const int kk = base + lane;
                  if (kk < nj)
                  {
                     const int    m = nb_col(k, kk);
                     const double a = nb_val(k, kk);
                     if (sgn * a < 0.0)
                     {
                        const int pos = lookup(m);
                        if (pos >= 0) Wv[pos] += distribute * a;
                        else if (m == i) misc[0] += distribute * a;
                     }
                  }
                  WAVE_SYNC();
               }
            }
            else
            {
               if (lane == 0) misc[0] += aij;
               WAVE_SYNC();
            }
            continue;
         }
         if (!(t & 8))
         { // weak connection lumped into the diagonal
            if (lane == 0) misc[0] += aij;
            WAVE_SYNC();
         }
      }
      WAVE_SYNC();
      const double diagonal = misc[0];
      if (diagonal != 0.0)
         for (int q = lane; q < cnt; q += G) Wv[q] = Wv[q] / (-diagonal);
      WAVE_SYNC();
      stamp(staged ? 5 : 6);
      // ---- 5. truncation and output (serial parts on lane 0, exactly the oracle's sequence)
      const long long o = uofs[i];
      int             out_cnt = cnt;
      if (trunc_factor > 0.0 && cnt > 0)
      {
         if (lane == 0)
         {
            double mx = 0.0, tot = 0.0, kept = 0.0;
            for (int q = 0; q < cnt; q++)
            {
               if (fabs(Wv[q]) > mx) mx = fabs(Wv[q]);
               tot += Wv[q];
            }
            int c2 = 0;
            for (int q = 0; q < cnt; q++)
               if (fabs(Wv[q]) >= trunc_factor * mx)
               {
                  ucol[c2] = ucol[q];
                  Wv[c2]   = Wv[q];
                  kept += Wv[c2];
                  c2++;
               }
            if (kept != 0.0)
            {
               const double sc = tot / kept;
               for (int q = 0; q < c2; q++) Wv[q] *= sc;
            }
            rofs[0] = c2;
         }
         WAVE_SYNC();
         cnt = out_cnt = rofs[0];
      }
      if (pmax > 0 && cnt > pmax)
      {
         // does a group of equal |w| straddle the cut?  If not the kept set is the top pmax.
         bool straddle = false;
         for (int base = 0; base < cnt; base += G)
         {
            const int q = base + lane;
            bool      s = false;
            if (q < cnt)
            {
               const double aq = fabs(Wv[q]);
               int          gt = 0, eq = 0;
               for (int r = 0; r < cnt; r++)
               {
                  const double ar = fabs(Wv[r]);
                  gt += ar > aq;
                  eq += ar == aq;
               }
               s = gt < pmax && gt + eq > pmax;
               craw[q]  = gt; // rank by strict dominance
            }
            straddle = straddle || gballot(s) != 0ULL;
         }
         WAVE_SYNC();
         if (lane == 0)
         {
            double tot = 0.0, kept = 0.0;
            for (int q = 0; q < cnt; q++) tot += Wv[q];
            // the kept entries are compacted to the front of ucol / Wv in place (LDS; private arrays would live in scratch memory)
            int nkpt = 0;
            if (!straddle)
            {
               for (int q = 0; q < cnt && nkpt < pmax; q++)
                  if (craw[q] < pmax) { ucol[nkpt] = ucol[q]; Wv[nkpt] = Wv[q]; nkpt++; }
            }
            else
            {
               qsort_abs(ucol, Wv, cnt, htb); // exact tie order of the reference algorithm; the hash table is dead, its slots hold the stack
               nkpt = pmax;
            }
            for (int a = 1; a < nkpt; a++)
            {
               const int    cc = ucol[a];
               const double ww = Wv[a];
               int          b  = a - 1;
               while (b >= 0 && ucol[b] > cc) { ucol[b + 1] = ucol[b]; Wv[b + 1] = Wv[b]; b--; }
               ucol[b + 1] = cc;
               Wv[b + 1]   = ww;
            }
            for (int q = 0; q < nkpt; q++) kept += Wv[q];
            const double sc = (kept != 0.0) ? tot / kept : 1.0;
            for (int q = 0; q < nkpt; q++)
            {
               lcol[o + q] = ucol[q];
               lw[o + q]   = (kept != 0.0) ? Wv[q] * sc : Wv[q];
            }
            pcnt[i] = nkpt;
         }
      }
      else
      { // few entries: column order by insertion on lane 0
         if (lane == 0)
         {
            for (int a = 1; a < out_cnt; a++)
            {
               const int    cc = ucol[a];
               const double ww = Wv[a];
               int          b  = a - 1;
               while (b >= 0 && ucol[b] > cc) { ucol[b + 1] = ucol[b]; Wv[b + 1] = Wv[b]; b--; }
               ucol[b + 1] = cc;
               Wv[b + 1]   = ww;
            }
            for (int q = 0; q < out_cnt; q++) { lcol[o + q] = ucol[q]; lw[o + q] = Wv[q]; }
            pcnt[i] = out_cnt;
         }
      }
      WAVE_SYNC();
      stamp(7);
      if (prof && lane == 0) atomicAdd(&prof[staged ? 8 : 9], 1ULL);
   }
}

__global__ __launch_bounds__(256) void k_interp_gather(int n, const long long *__restrict__ uofs,
                                                       const int *__restrict__ prp,
                                                       const int *__restrict__ lcol,
                                                       const double *__restrict__ lw,
                                                       const int *__restrict__ cidx, int *__restrict__ pc,
                                                       double *__restrict__ pv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int       s = prp[i], e = prp[i + 1];
   const long long o = uofs[i];
   for (int q = s; q < e; q++)
   {
      pc[q] = cidx[lcol[o + (q - s)]];
      pv[q] = lw[o + (q - s)];
   }
}

void amg_interp_extpi(const DCsr &A, const unsigned char *smask, const int *cf, int pmax,
                      double trunc_factor, DCsr &P, const int *dof, int interp_type)
{
   HDA_REQUIRE(interp_type == 6 || interp_type == 17 || interp_type == 3 || interp_type == 8, "interpolation type not implemented");
   if (interp_type == 17)
   { // mm-ext+i is an operator of its own, built from sparse products (hda_amg_agg.hip)
      amg_interp_mm_extpi(A, smask, cf, pmax, trunc_factor, P, dof);
      return;
   }
   const int itype = (interp_type == 3 || interp_type == 8) ? interp_type : 6;
   const int n = A.nrows;
   const int g = ceil_div(std::max(n, 1), 256);
   DArray<int>       nsC((size_t)n + 1), ub((size_t)n + 1), hsz((size_t)n + 1), cmark((size_t)n + 1), cidx((size_t)n + 1), nt((size_t)n + 1);
   DArray<long long> uofs((size_t)n + 1), hofs((size_t)n + 1);
   k_count_strongC<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, nsC.data());
   k_interp_ub<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, nsC.data(), ub.data(), hsz.data(), cmark.data(), nt.data());
   HDA_TRACE("  interp: ub done");
   exclusive_scan(n, cmark.data(), cidx.data(), nullptr);
   // rows that fit the LDS budget go to the wave-per-row kernel, the rest keep the global-hash kernel
   DArray<unsigned char> rowmode((size_t)n + 1);
   DArray<int>           mx(3);
   mx.zero();
   k_max3<<<std::min(g, 1024), 256, 0, STREAM>>>(n, A.rowptr.data(), ub.data(), nt.data(), mx.data());
   int hmx[3] = {0, 0, 0};
   mx.download(hmx, 3);
   const char *imode = nullptr; // ("thread" / "wave" forced one kernel: diagnostics of rounds 1-3)
   // G lanes share a row (8 for stencil rows, a whole wavefront for the long rows of coarse levels): the row, its
   // candidates and its neighbour rows are staged in LDS by coalesced reads.  The thread-per-row kernel keeps the rows
   // that exceed the LDS budget, and everything when pmax is outside the group kernel's range.
   bool use_wave = pmax > 0 && pmax <= 16 && itype == 6; // the group kernel is extended+i only
   if (imode && !strcmp(imode, "thread")) use_wave = false;
   constexpr int g_env = 0;
   int G = 64; // (32 lanes for rows of ~30 entries measured twice as slow as 64: eight rows' staging areas leave one workgroup per CU)
   if (hmx[0] <= 8) G = 8;
   else if (hmx[0] <= 16) G = 16;
   if (imode && !strcmp(imode, "wave")) G = 64;
   if (g_env == 8 || g_env == 16 || g_env == 32 || g_env == 64) G = g_env;
   const int gpb = 256 / G;
   int cap_row = 8, cap_ub = 16, cap_nbr = 64;
   cap_row = std::min(256, std::max(8, (hmx[0] + 7) / 8 * 8)); // rows longer than 256 entries keep the thread kernel
   while (cap_ub < hmx[1] && cap_ub < 1024) cap_ub <<= 1;
   // Neighbour-row staging area per row group.  Since round 3 neighbour rows of up to four entries per lane are read ONCE from memory
   // and swept on registers (reg_nbr below), which beat the LDS-staged form on every level of the 256^3 hierarchy (level 1: 70 -> 59 ms,
   // level 2: 36 -> 20, levels 0 / 3: 23.3 -> 22.2 / 5.0 -> 3.0, tools/gpurun/r03_zg.sh ... r03_zi.sh) because the small area lets more
   // rows be in flight per CU: the default area is 8 entries, i.e. staging is off in practice.  HDA_INTERP_NBR=<entries> brings it back.
   constexpr int nbr_env = 0; // (the LDS staging of neighbour rows lost to the register sweeps on every level: closed in round 3)
   cap_nbr = 8;
   if (nbr_env > 8)
   {
      while (cap_nbr < hmx[2] && cap_nbr < nbr_env) cap_nbr <<= 1;
      if (hmx[2] < cap_nbr) cap_nbr = std::max(8, (hmx[2] + 7) / 8 * 8); // every row's neighbourhood fits: no more than needed
   }
   while (cap_nbr > 64 && interp_wave_doubles(cap_row, cap_ub, cap_nbr) * 8 * gpb > 150 * 1024) cap_nbr >>= 1; // the workgroup's row groups must fit the LDS budget
   if (interp_wave_doubles(cap_row, cap_ub, cap_nbr) * 8 * gpb > 150 * 1024) use_wave = false;
   if (use_wave)
      k_interp_rowmode<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), ub.data(), nt.data(), cap_row, cap_ub, cap_nbr, rowmode.data(), hsz.data());
   exclusive_scan64(n, ub.data(), uofs.data());
   exclusive_scan64(n, hsz.data(), hofs.data());
   long long tot_u = 0, tot_h = 0;
   int       nc    = 0;
   HDA_HIP(hipMemcpyAsync(&tot_u, uofs.data() + n, 8, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&tot_h, hofs.data() + n, 8, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&nc, cidx.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   DArray<int>    lcol((size_t)std::max<long long>(tot_u, 1)), htab((size_t)std::max<long long>(tot_h, 1));
   DArray<double> lw((size_t)std::max<long long>(tot_u, 1));
   if (tot_h) HDA_HIP(hipMemsetAsync(htab.data(), 0xFF, sizeof(int) * htab.size(), STREAM));
   DArray<int> pcnt((size_t)n + 1);
   pcnt.zero();
   HDA_TRACE("  interp: build (tot_u=%lld tot_h=%lld nc=%d maxrow=%d maxub=%d maxnbr=%d lanes/row=%d caps %d %d %d)", tot_u, tot_h, nc, hmx[0], hmx[1], hmx[2], use_wave ? G : 1, cap_row, cap_ub, cap_nbr);
   if (use_wave)
   {
      const size_t lds = interp_wave_doubles(cap_row, cap_ub, cap_nbr) * 8 * gpb;
      constexpr int s3_scan = 1 << 30; // candidates up to which duplicates are found by scanning
      constexpr bool want_prof = false; // (shader-clock stage profile of rounds 2-3)
      // neighbour rows read from memory, at most four entries per lane: one pass, the sweeps on registers (HDA_INTERP_REG=0: three passes)
      constexpr int reg_nbr = 1;
      DArray<unsigned long long> prof;
      if (want_prof)
      {
         prof.alloc(10);
         prof.zero();
      }
      DArray<unsigned char> sc((size_t)std::max(A.nnz, 1));
      if (A.nnz) k_strongC_flag<<<std::min(ceil_div(A.nnz, 256), 1 << 16), 256, 0, STREAM>>>(A.nnz, A.col.data(), smask, cf, sc.data());
      // compact lists of every row's strong C columns: a row gathers its candidates from them in one flat pass instead of walking the
      // full rows of its strong F neighbours one after the other (HDA_INTERP_SCLIST=0: the walk)
      constexpr bool sclist = true;
      DArray<int> scofs, scc;
      if (sclist && cap_nbr <= 8)
      {
         scofs.alloc((size_t)n + 1);
         exclusive_scan(n, nsC.data(), scofs.data(), nullptr);
         int tot_sc = 0;
         HDA_HIP(hipMemcpyAsync(&tot_sc, scofs.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
         Context::get().sync();
         scc.alloc((size_t)std::max(tot_sc, 1));
         k_fill_strongC<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), smask, cf, scofs.data(), scc.data());
      }
      auto launch = [&](auto kern) {
         HDA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
         kern<<<std::min((ceil_div(n, gpb) + 7) / 8 * 8, 256 * 16), 256, lds, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, sc.data(), cf, nsC.data(),
                                                                      uofs.data(), cap_row, cap_ub, cap_nbr, pmax, trunc_factor, rowmode.data(),
                                                                      lcol.data(), lw.data(), pcnt.data(), dof, s3_scan, prof.data(), reg_nbr,
                                                                      scc.data() ? scofs.data() : nullptr, scc.data());
      };
      if (G == 8) launch(k_interp_wave<8>);
      else if (G == 16) launch(k_interp_wave<16>);
      else if (G == 32) launch(k_interp_wave<32>);
      else launch(k_interp_wave<64>);
      if (want_prof)
      {
         unsigned long long h[10];
         prof.download(h, 10);
         const double rows = (double)std::max<unsigned long long>(h[8] + h[9], 1);
         fprintf(stderr, "[hda] interp stages, cycles per row (%llu staged, %llu unstaged rows): row %.0f nbr-stage %.0f cand %.0f dedupe %.0f prep %.0f "
                         "accumulate(staged) %.0f accumulate(unstaged) %.0f output %.0f\n", h[8], h[9], h[0] / rows, h[1] / rows, h[2] / rows,
                 h[3] / rows, h[4] / rows, h[5] / rows, h[6] / rows, h[7] / rows);
      }
      // the rows the wave kernel left to the thread kernel (rowmode): also when none of them needs a hash table
      // (C points and non-interpolated F points among them still have to report their entry count)
      k_interp_build<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, cf, uofs.data(), hofs.data(),
                                            lcol.data(), lw.data(), htab.data(), pmax, trunc_factor, pcnt.data(), rowmode.data(), dof, itype);
   }
   else
      k_interp_build<<<g, 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), smask, cf, uofs.data(), hofs.data(),
                                            lcol.data(), lw.data(), htab.data(), pmax, trunc_factor, pcnt.data(), nullptr, dof, itype);
   HDA_TRACE("  interp: build done");
   P.nrows = n;
   P.ncols = nc;
   P.rowptr.alloc((size_t)n + 1);
   require_int32_total(n, pcnt.data(), "interpolation operator");
   exclusive_scan(n, pcnt.data(), P.rowptr.data(), nullptr);
   HDA_HIP(hipMemcpyAsync(&P.nnz, P.rowptr.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   P.col.alloc((size_t)std::max(P.nnz, 1));
   P.val.alloc((size_t)std::max(P.nnz, 1));
   k_interp_gather<<<g, 256, 0, STREAM>>>(n, uofs.data(), P.rowptr.data(), lcol.data(), lw.data(), cidx.data(),
                                          P.col.data(), P.val.data());
}

// ------------------------------------------------------------------ SpGEMM

__global__ __launch_bounds__(256) void k_spgemm_ub(int n, const int *__restrict__ xrp,
                                                   const int *__restrict__ xcj,
                                                   const int *__restrict__ yrp, int ycols,
                                                   int *__restrict__ hsz)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   long u = 0;
   for (int k = xrp[i]; k < xrp[i + 1]; k++)
   {
      const int r = xcj[k];
      u += yrp[r + 1] - yrp[r];
   }
   if (u > ycols) u = ycols; // cannot have more distinct columns than Y has
   hsz[i] = u ? max(4, pow2ceil_dev((int)(2 * u))) : 0;
}

// One thread per output row; products are visited in (k ascending, q ascending) order and
// accumulated per column in that order -- the same order as a sequential Gustavson pass.
template <bool NUMERIC>
__global__ __launch_bounds__(256) void k_spgemm_hash(int r0, int r1, const int *__restrict__ xrp,
                                                     const int *__restrict__ xcj,
                                                     const double *__restrict__ xv,
                                                     const int *__restrict__ yrp,
                                                     const int *__restrict__ ycj,
                                                     const double *__restrict__ yv,
                                                     const long long *__restrict__ hofs, long long hbase,
                                                     int *__restrict__ hkey, double *__restrict__ hval,
                                                     int *__restrict__ cnt)
{
   const int i = r0 + blockIdx.x * 256 + threadIdx.x;
   if (i >= r1) return;
   const long long o  = hofs[i] - hbase;
   const int       hs = (int)(hofs[i + 1] - hofs[i]);
   int             c  = 0;
   if (hs)
   {
      int      *K    = hkey + o;
      double   *V    = hval + o;
      const int mask = hs - 1;
      int       lg   = 0;
      while ((1 << lg) < hs) lg++;
      for (int k = xrp[i]; k < xrp[i + 1]; k++)
      {
         const int    r = xcj[k];
         const double a = NUMERIC ? xv[k] : 0.0;
         for (int q = yrp[r]; q < yrp[r + 1]; q++)
         {
            const int    j = ycj[q];
            const double t = NUMERIC ? a * yv[q] : 0.0;
            unsigned     h = hash_slot(j, lg);
            for (;;)
            {
               const int key = K[h];
               if (key < 0)
               {
                  K[h] = j;
                  if (NUMERIC) V[h] = t;
                  c++;
                  break;
               }
               if (key == j)
               {
                  if (NUMERIC) V[h] += t;
                  break;
               }
               h = (h + 1) & mask;
            }
         }
      }
   }
   cnt[i] = c;
}

__global__ __launch_bounds__(256) void k_spgemm_gather(int r0, int r1, const long long *__restrict__ hofs,
                                                       long long hbase, const int *__restrict__ hkey,
                                                       const double *__restrict__ hval,
                                                       const int *__restrict__ crp, int *__restrict__ ccj,
                                                       double *__restrict__ cv)
{
   const int i = r0 + blockIdx.x * 256 + threadIdx.x;
   if (i >= r1) return;
   const long long o  = hofs[i] - hbase;
   const int       hs = (int)(hofs[i + 1] - hofs[i]);
   const int       s  = crp[i];
   int             w  = s;
   for (int h = 0; h < hs; h++)
   { // in slot order; the rows are column-sorted afterwards (sort_rows_segmented: the one-thread insertion sort that stood here took
     // 22 of the 27 ms of a 3 049-row product with 4 661-product rows, round-5 series-B trace)
      const int key = hkey[o + h];
      if (key < 0) continue;
      ccj[w] = key;
      cv[w]  = hval[o + h];
      w++;
   }
}

static long long spgemm_slot_budget()
{
   static long long b = -1;
   if (b < 0)
   {
      const char *e = getenv("HDA_SPGEMM_SLOTS");
      b             = e ? atoll(e) : (1LL << 31); // 2 Gi slots = 24 GiB of keys+values
      if (b < 1024) b = 1024;
   }
   return b;
}

static void spgemm_hash(const DCsr &X, const DCsr &Y, DCsr &C)
{
   HDA_REQUIRE(X.ncols <= Y.nrows || X.nnz == 0, "spgemm: inner dimensions");
   const int n = X.nrows;
   C.nrows     = n;
   C.ncols     = Y.ncols;
   C.rowptr.alloc((size_t)n + 1);
   if (n == 0)
   {
      C.rowptr.zero();
      C.nnz = 0;
      C.col.alloc(1);
      C.val.alloc(1);
      return;
   }
   const int         g = ceil_div(n, 256);
   DArray<int>       hsz((size_t)n + 1), cnt((size_t)n + 1);
   DArray<long long> hofs((size_t)n + 1);
   k_spgemm_ub<<<g, 256, 0, STREAM>>>(n, X.rowptr.data(), X.col.data(), Y.rowptr.data(), Y.ncols, hsz.data());
   exclusive_scan64(n, hsz.data(), hofs.data());
   long long H = 0;
   HDA_HIP(hipMemcpyAsync(&H, hofs.data() + n, 8, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   HDA_TRACE("  spgemm: n=%d H=%lld", n, H);
   const long long budget = spgemm_slot_budget();
   // batches of rows whose tables fit the budget
   std::vector<int>       bstart{0};
   std::vector<long long> hofs_h;
   if (H > budget)
   {
      hofs_h.resize((size_t)n + 1);
      hofs.download(hofs_h.data(), (size_t)n + 1);
      int r = 0;
      while (r < n)
      {
         long long base = hofs_h[r];
         int       e    = (int)(std::upper_bound(hofs_h.begin() + r, hofs_h.end(), base + budget) - hofs_h.begin()) - 1;
         if (e <= r) e = r + 1; // a single row larger than the budget still gets its table
         r = e;
         bstart.push_back(r);
      }
   }
   else
      bstart.push_back(n);
   const int nb = (int)bstart.size() - 1;
   long long maxslots = 0;
   if (nb == 1) maxslots = H;
   else
      for (int b = 0; b < nb; b++) maxslots = std::max(maxslots, hofs_h[bstart[b + 1]] - hofs_h[bstart[b]]);
   DArray<int>    hkey((size_t)std::max<long long>(maxslots, 1));
   DArray<double> hval((size_t)std::max<long long>(maxslots, 1));
   auto run_batch = [&](int b, bool numeric) {
      const int       r0 = bstart[b], r1 = bstart[b + 1];
      const long long hb = (nb == 1) ? 0 : hofs_h[r0];
      const long long hs = (nb == 1) ? H : hofs_h[r1] - hb;
      if (hs) HDA_HIP(hipMemsetAsync(hkey.data(), 0xFF, sizeof(int) * (size_t)hs, STREAM));
      const int gb = ceil_div(r1 - r0, 256);
      if (numeric)
         k_spgemm_hash<true><<<gb, 256, 0, STREAM>>>(r0, r1, X.rowptr.data(), X.col.data(), X.val.data(), Y.rowptr.data(),
                                                     Y.col.data(), Y.val.data(), hofs.data(), hb, hkey.data(), hval.data(), cnt.data());
      else
         k_spgemm_hash<false><<<gb, 256, 0, STREAM>>>(r0, r1, X.rowptr.data(), X.col.data(), X.val.data(), Y.rowptr.data(),
                                                      Y.col.data(), Y.val.data(), hofs.data(), hb, hkey.data(), hval.data(), cnt.data());
   };
   auto gather_batch = [&](int b) {
      const int       r0 = bstart[b], r1 = bstart[b + 1];
      const long long hb = (nb == 1) ? 0 : hofs_h[r0];
      k_spgemm_gather<<<ceil_div(r1 - r0, 256), 256, 0, STREAM>>>(r0, r1, hofs.data(), hb, hkey.data(), hval.data(),
                                                                 C.rowptr.data(), C.col.data(), C.val.data());
   };
   auto finish_rowptr = [&]() {
      require_int32_total(n, cnt.data(), "sparse product");
      exclusive_scan(n, cnt.data(), C.rowptr.data(), nullptr);
      HDA_HIP(hipMemcpyAsync(&C.nnz, C.rowptr.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
      C.col.alloc((size_t)std::max(C.nnz, 1));
      C.val.alloc((size_t)std::max(C.nnz, 1));
   };
   if (nb == 1)
   {
      run_batch(0, true);
      HDA_TRACE("  spgemm: numeric done");
      finish_rowptr();
      HDA_TRACE("  spgemm: rowptr done nnz=%d", C.nnz);
      gather_batch(0);
      HDA_TRACE("  spgemm: gather done");
   }
   else
   {
      for (int b = 0; b < nb; b++) run_batch(b, false); // symbolic: counts only
      finish_rowptr();
      for (int b = 0; b < nb; b++)
      {
         run_batch(b, true);
         gather_batch(b);
      }
   }
   sort_rows_segmented(C);
   HDA_TRACE("  spgemm: rows sorted");
}

// ---- expand / sort / compress SpGEMM in LDS ---------------------------------------------
// A workgroup takes a run of consecutive rows holding ~T products, expands the products in
// enumeration order p (k ascending over the X row, q ascending over the Y row) into LDS as
// keys (local row | column | p), bitonic-sorts the keys, and sums every (row, column)
// segment sequentially in p order -- exactly the accumulation order of a sequential
// Gustavson pass, so results stay bit-identical to the oracle, with coalesced global
// traffic and no global hash tables.  Two passes (count, then fill) avoid scratch storage.

__global__ __launch_bounds__(256) void k_entry_len(int nnz, const int *__restrict__ xcj, const int *__restrict__ yrp, int *__restrict__ len)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256)
   {
      const int r = xcj[k];
      len[k]      = yrp[r + 1] - yrp[r];
   }
}
__global__ __launch_bounds__(256) void k_row_np_max(int n, const int *__restrict__ xrp, const long long *__restrict__ eoff, int *mx)
{
   int m = 0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
   {
      const long long d = eoff[xrp[i + 1]] - eoff[xrp[i]];
      m                 = max(m, (int)min(d, (long long)0x7fffffff));
   }
   for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
   if ((threadIdx.x & 63) == 0) atomicMax(mx, m);
}
__global__ __launch_bounds__(256) void k_esc_chunk_rows(int nchunks, int nrows, long long T, const int *__restrict__ xrp,
                                                        const long long *__restrict__ eoff, int *__restrict__ chunk_row)
{
   const int c = blockIdx.x * 256 + threadIdx.x;
   if (c > nchunks) return;
   if (c == nchunks) { chunk_row[c] = nrows; return; }
   const long long target = (long long)c * T;
   int             lo = 0, hi = nrows; // smallest r whose first product index >= target
   while (lo < hi)
   {
      const int mid = (lo + hi) >> 1;
      if (eoff[xrp[mid]] < target) lo = mid + 1;
      else hi = mid;
   }
   chunk_row[c] = lo;
}

constexpr int kEscEntries = 2048; // X entries of a chunk staged in LDS for the expansion

// One pass: products -> keys in registers (values in LDS, indexed by p and never moved), sort, segment sums in
// enumeration order, results written to a scratch CSR whose rows start at rowstart[i] (<= the row's first product
// index), counts in cnt.  A chunk holds fewer than 256*PER products.
template <int PER, int NT>
__global__ __launch_bounds__(NT) void k_spgemm_esc(int nchunks, const int *__restrict__ chunk_row,
                                                    const int *__restrict__ xrp, const int *__restrict__ xcj,
                                                    const double *__restrict__ xv, const int *__restrict__ yrp,
                                                    const int *__restrict__ ycj, const double *__restrict__ yv,
                                                    const long long *__restrict__ eoff, long long *__restrict__ rowstart,
                                                    int *__restrict__ cnt, int *__restrict__ scol, double *__restrict__ sval,
                                                    int *__restrict__ err)
{
   constexpr int cap = NT * PER;
   extern __shared__ unsigned long long esc_lds[];
   unsigned long long *keys = esc_lds;                    // cap keys (after the sort); entry of every product before it
   double             *vals = (double *)(esc_lds + cap);  // cap products, indexed by p
   unsigned short     *ent  = (unsigned short *)keys;     // staged entry of product p (dead before the sort uses keys)
   unsigned           *eofs = (unsigned *)(ent + cap);    // kEscEntries+1: (local row << 13) | entry offset relative to p0; dead likewise
   __shared__ int      scan[NT];
   const int           tid = threadIdx.x;
   for (int c = blockIdx.x; c < nchunks; c += gridDim.x)
   {
      const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
      if (r0 == r1) continue;
      if (r1 - r0 >= (1 << 19))
      { // the local row does not fit its field: the caller repeats the product on the hash path
         if (tid == 0) *err = 1;
         continue;
      }
      const int       e0 = xrp[r0], e1 = xrp[r1], ne = e1 - e0;
      const long long p0 = eoff[e0];
      const int       span = (int)(eoff[e1] - p0);
      if (span == 0) continue;
      const bool staged = ne <= kEscEntries;
      if (staged)
      { // entry offsets and entry -> local row, coalesced; then the entry of every product
         for (int q = tid; q <= ne; q += NT) eofs[q] = (unsigned)(eoff[e0 + q] - p0);
         __syncthreads();
         for (int r = r0 + tid; r < r1; r += NT)
            for (int e = xrp[r]; e < xrp[r + 1]; e++) eofs[e - e0] |= (unsigned)(r - r0) << 13;
         __syncthreads();
         for (int q = tid; q < ne; q += NT)
         {
            const int a = (int)(eofs[q] & 0x1FFFu), b = (int)(eofs[q + 1] & 0x1FFFu); // eofs[ne] = span carries no row tag
            for (int p = a; p < b; p++) ent[p] = (unsigned short)q;
         }
         __syncthreads();
      }
      // ---- expand: lane-consecutive products, so the gathers of one Y row coalesce
      unsigned long long k[PER];
#pragma unroll
      for (int m = 0; m < PER; m++)
      {
         const int          p   = tid + NT * m;
         unsigned long long key = ~0ULL;
         if (p < span)
         {
            int e, lrow, q;
            if (staged)
            {
               const int      lo = ent[p];
               const unsigned w  = eofs[lo];
               e    = e0 + lo;
               lrow = (int)(w >> 13);
               q    = p - (int)(w & 0x1FFFu);
            }
            else
            {
               const long long gp = p0 + p;
               int             lo = e0, hi = e1 - 1; // last entry with eoff <= gp
               while (lo < hi)
               {
                  const int mid = (lo + hi + 1) >> 1;
                  if (eoff[mid] <= gp) lo = mid;
                  else hi = mid - 1;
               }
               e      = lo;
               int rl = r0, rh = r1 - 1;
               while (rl < rh)
               {
                  const int mid = (rl + rh + 1) >> 1;
                  if (xrp[mid] <= e) rl = mid;
                  else rh = mid - 1;
               }
               lrow = rl - r0;
               q    = (int)(gp - eoff[e]);
            }
            const int yq = yrp[xcj[e]] + q;
            key          = ((unsigned long long)lrow << 44) | ((unsigned long long)(unsigned)ycj[yq] << 13) | (unsigned long long)p;
            vals[p]      = xv[e] * yv[yq];
         }
         k[m] = key;
      }
      block_sort_regs<PER, unsigned long long, NT>(k, keys, tid); // its first LDS use is behind a barrier: ent is dead by then
      __syncthreads();
#pragma unroll
      for (int r = 0; r < PER; r++) keys[tid * PER + r] = k[r];
      __syncthreads();
      // ---- heads of (row, column) segments; inclusive head count per sorted position
      const int t0 = tid * PER;
      int       local = 0;
      for (int t = t0; t < t0 + PER && t < span; t++) local += (t == 0) || ((keys[t] >> 13) != (keys[t - 1] >> 13));
      scan[tid] = local;
      __syncthreads();
      for (int o = 1; o < NT; o <<= 1)
      {
         const int add = (tid >= o) ? scan[tid - o] : 0;
         __syncthreads();
         scan[tid] += add;
         __syncthreads();
      }
      int incl = scan[tid] - local; // heads before this lane's range
      for (int t = t0; t < t0 + PER && t < span; t++)
      {
         const unsigned long long kt = keys[t];
         const bool head = (t == 0) || ((kt >> 13) != (keys[t - 1] >> 13));
         incl += head;
         const int row = r0 + (int)(kt >> 44);
         if (head)
         {
            if (t == 0 || (kt >> 44) != (keys[t - 1] >> 44)) rowstart[row] = p0 + (incl - 1);
            double sum = vals[kt & 0x1FFF];
            for (int u = t + 1; u < span && (keys[u] >> 13) == (kt >> 13); u++) sum += vals[keys[u] & 0x1FFF];
            scol[p0 + incl - 1] = (int)((kt >> 13) & 0x7FFFFFFF);
            sval[p0 + incl - 1] = sum;
         }
         if (t == span - 1 || (keys[t + 1] >> 44) != (kt >> 44)) cnt[row] = incl; // provisional: heads up to the row's end
      }
      __syncthreads();
   }
}

// cnt[row] currently holds (#heads in the chunk up to the row's end); turn it into the row's
// own count using rowstart, which holds chunk_base + (#heads before the row)
__global__ __launch_bounds__(256) void k_esc_fix_counts(int n, const int *__restrict__ xrp, const long long *__restrict__ eoff,
                                                        const int *__restrict__ chunk_of_row_base_dummy, const long long *__restrict__ rowstart,
                                                        const long long *__restrict__ chunkbase, int *__restrict__ cnt)
{
   (void)xrp; (void)eoff; (void)chunk_of_row_base_dummy;
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   if (cnt[i] > 0) cnt[i] = (int)(chunkbase[i] + cnt[i] - rowstart[i]);
}

// chunkbase[i] = first product index of the chunk that owns row i
__global__ __launch_bounds__(256) void k_esc_chunkbase(int nchunks, const int *__restrict__ chunk_row, const int *__restrict__ xrp,
                                                       const long long *__restrict__ eoff, long long *__restrict__ chunkbase)
{
   const int c = blockIdx.x;
   if (c >= nchunks) return;
   const int       r0 = chunk_row[c], r1 = chunk_row[c + 1];
   const long long p0 = (r0 < r1) ? eoff[xrp[r0]] : 0;
   for (int r = r0 + threadIdx.x; r < r1; r += 256) chunkbase[r] = p0;
}

// C rows out of the scratch rows, 8 lanes per row
__global__ __launch_bounds__(256) void k_esc_compact(int n, const int *__restrict__ crp, const long long *__restrict__ rowstart,
                                                     const int *__restrict__ scol, const double *__restrict__ sval,
                                                     int *__restrict__ ccj, double *__restrict__ cv)
{
   const int  lane = threadIdx.x & 7;
   const long G    = (long)gridDim.x * 32;
   for (long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; i < n; i += G)
   {
      const int       s = crp[i], e = crp[i + 1];
      const long long o = rowstart[i];
      for (int q = s + lane; q < e; q += 8)
      {
         ccj[q] = scol[o + (q - s)];
         cv[q]  = sval[o + (q - s)];
      }
   }
}

static bool use_hash_spgemm()
{
   static int m = -1;
   if (m < 0)
   {
      m = 0; // (the hash product stays the fallback for rows beyond the sort network and products beyond the scratch budget)
   }
   return m == 1;
}

void spgemm(const DCsr &X, const DCsr &Y, DCsr &C)
{
   HDA_REQUIRE(X.ncols <= Y.nrows || X.nnz == 0, "spgemm: inner dimensions");
   const int n = X.nrows;
   if (n == 0 || X.nnz == 0 || use_hash_spgemm()) return spgemm_hash(X, Y, C);
   DArray<int>       elen((size_t)X.nnz + 1), mx(1);
   DArray<long long> eoff((size_t)X.nnz + 1);
   k_entry_len<<<std::min(ceil_div(X.nnz, 256), 1 << 16), 256, 0, STREAM>>>(X.nnz, X.col.data(), Y.rowptr.data(), elen.data());
   exclusive_scan64(X.nnz, elen.data(), eoff.data());
   mx.zero();
   k_row_np_max<<<std::min(ceil_div(n, 256), 1024), 256, 0, STREAM>>>(n, X.rowptr.data(), eoff.data(), mx.data());
   long long total = 0;
   int       maxnp = 0;
   HDA_HIP(hipMemcpyAsync(&total, eoff.data() + X.nnz, 8, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&maxnp, mx.data(), 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   // LDS capacity cap (a power of two, >= 2 maxnp) and chunk target T = cap - maxnp: a chunk spans < T + maxnp = cap
   // products, and nearly all of them close to T, so the sort network runs almost full
   int cap = 2048;
   while (cap < 2 * maxnp && cap < 8192) cap <<= 1;
   const int T = std::max(cap - maxnp, 1);
   HDA_TRACE("  spgemm(esc): n=%d products=%lld max/row=%d cap=%d", n, total, maxnp, cap);
   // product scratch (12 B each) may take up to 30 % of the device memory: 7 G products on a 288 GB part,
   // enough for the 512^3 benchmark's largest Galerkin product (5 G); beyond it the hash path takes over
   static const long long scratch_cap = [] {
      size_t freeb = 0, totalb = 0;
      if (hipMemGetInfo(&freeb, &totalb) != hipSuccess) return 3LL << 30;
      return (long long)(0.3 * (double)totalb / 12.0);
   }(); // products
   if (maxnp > 4096 || total > scratch_cap) return spgemm_hash(X, Y, C); // outside the LDS path
   const int nchunks = (int)std::max<long long>(1, (total + T - 1) / T);
   DArray<int>       chunk_row((size_t)nchunks + 1), cnt((size_t)n + 1);
   DArray<long long> rowstart((size_t)n + 1), chunkbase((size_t)n + 1);
   DArray<int>       scol((size_t)std::max<long long>(total, 1));
   DArray<double>    sval((size_t)std::max<long long>(total, 1));
   k_esc_chunk_rows<<<ceil_div(nchunks + 1, 256), 256, 0, STREAM>>>(nchunks, n, T, X.rowptr.data(), eoff.data(), chunk_row.data());
   k_esc_chunkbase<<<nchunks, 256, 0, STREAM>>>(nchunks, chunk_row.data(), X.rowptr.data(), eoff.data(), chunkbase.data());
   cnt.zero();
   mx.zero(); // reused as the kernel's "row field overflow" flag
   static_assert(2048 * 2 + (kEscEntries + 1) * 4 <= 2048 * 8, "entry staging must fit the key area of the smallest chunk");
   const size_t lds  = (size_t)cap * 16;
   // Eight keys per thread whatever the chunk capacity: a 4096 / 8192-product chunk gets 512 / 1024 threads, so that the two / one
   // workgroups a CU has LDS for still put 16 wavefronts on it (with 256 threads they left the CU at 8 / 4 and ran 1.5 - 3x slower per
   // product, profiles/r03_kernel_experiments.md).  HDA_ESC_THREADS=256: the round-2 shape.
   constexpr bool wide = true;
   const int    nt   = wide ? cap / 8 : 256; // (four keys per thread -- twice the threads again -- lost 10-20 %: more cross-wave stages)
   const int    grid = std::min(nchunks, 256 * std::max(1, (int)((160 * 1024) / (lds + 4 * nt + 1024)))); // the resident workgroups: no tail wave

   auto launch = [&](auto kern) {
      HDA_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      kern<<<grid, nt, lds, STREAM>>>(nchunks, chunk_row.data(), X.rowptr.data(), X.col.data(), X.val.data(), Y.rowptr.data(),
                                       Y.col.data(), Y.val.data(), eoff.data(), rowstart.data(), cnt.data(), scol.data(), sval.data(),
                                       mx.data());
   };
   if (cap == 2048) launch(k_spgemm_esc<8, 256>);
   else if (cap == 4096) { if (wide) launch(k_spgemm_esc<8, 512>); else launch(k_spgemm_esc<16, 256>); }
   else { if (wide) launch(k_spgemm_esc<8, 1024>); else launch(k_spgemm_esc<32, 256>); }
   k_esc_fix_counts<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, nullptr, nullptr, nullptr, rowstart.data(), chunkbase.data(), cnt.data());
   C.nrows = n;
   C.ncols = Y.ncols;
   C.rowptr.alloc((size_t)n + 1);
   require_int32_total(n, cnt.data(), "sparse product");
   exclusive_scan(n, cnt.data(), C.rowptr.data(), nullptr);
   int overflow = 0;
   HDA_HIP(hipMemcpyAsync(&C.nnz, C.rowptr.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&overflow, mx.data(), 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   if (overflow) return spgemm_hash(X, Y, C); // a chunk spanned 2^19 rows or more (long runs of empty rows)
   C.col.alloc((size_t)std::max(C.nnz, 1));
   C.val.alloc((size_t)std::max(C.nnz, 1));
   k_esc_compact<<<std::min(ceil_div((long long)n * 8, 256), 1 << 16), 256, 0, STREAM>>>(n, C.rowptr.data(), rowstart.data(), scol.data(),
                                                                                  sval.data(), C.col.data(), C.val.data());
   HDA_TRACE("  spgemm(esc): nnz=%d", C.nnz);
}

void amg_rap(const DCsr &A, const DCsr &P, const DCsr &R, DCsr &Ac)
{
   DCsr AP;
   spgemm(A, P, AP);
   spgemm(R, AP, Ac);
}

// --------------------------------------------------------------- hierarchy

static bool is_jacobi_type(int t) { return t == 18 || t == 0 || t == 7; }
static bool is_gs_type(int t) { return t == 3 || t == 4 || t == 6 || t == 8 || t == 13 || t == 14; }
static bool is_l1_gs_type(int t) { return t == 8 || t == 13 || t == 14; }

// divisor of the sweep: l1 row sums (18), hypre's "option 4" l1 (13/14/8: a_ii plus half the
// off-rank row sum) or the plain diagonal (0/7/3/4/6).  Extracting the diagonal is option 4
// with the ghost part ignored, which is what columns < nrows give.
// two relaxation types with the same divisors (build_dinv below): the up sweep then uses the down sweep's array -- one pass over A
// less per level in the setup, and on row blocks one sweep-order copy of the divisors serves both (GsPlan::sd_src)
static bool same_divisors(int t1, int t2)
{
   auto cls = [](int t) { return t == 18 ? 1 : is_l1_gs_type(t) ? 2 : 0; };
   return cls(t1) == cls(t2);
}
static void build_dinv(const DCsr &A, int relax_type, double weight, DArray<double> &dinv, const int *d_part = nullptr, int nblk = 0)
{
   DArray<double> d((size_t)std::max(A.nrows, 1));
   dinv.alloc((size_t)std::max(A.nrows, 1));
   if (relax_type == 18) l1_row_norms(A, 1, d.data());
   else if (is_l1_gs_type(relax_type)) l1_row_norms(A, 4, d.data(), d_part, nblk); // entries that leave the row's block count as off-rank
   else extract_diag(A, d.data());
   make_dinv(A.nrows, d.data(), weight, dinv.data());
}

// blocks = 0: the sequential algorithms (one block) while they are affordable, beyond that blocks of at least four times the
// operator's bandwidth -- on a grid in lexicographic order four planes, thick enough for most rows to have all their neighbours
// inside the block, which is what keeps the hybrid sweep and HMIS within an iteration of their one-block forms
__global__ __launch_bounds__(256) void k_bandwidth(int n, const int *__restrict__ rp, const int *__restrict__ cj, int *bw)
{
   int m = 0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
      for (int k = rp[i]; k < rp[i + 1]; k++)
         if (cj[k] < n) m = max(m, abs(cj[k] - (int)i));
   for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
   if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(bw, m);
}
int amg_auto_blocks(const DCsr &A)
{
   static const long long min_rows  = getenv("HDA_BLOCKS_MIN_ROWS") ? atoll(getenv("HDA_BLOCKS_MIN_ROWS")) : 100000;
   constexpr long long min_block = 32768;
   const int n = A.nrows;
   if (n <= min_rows) return 1;
   DArray<int> bw(1);
   bw.zero();
   k_bandwidth<<<std::min(ceil_div(n, 256), 2048), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), bw.data());
   int h = 0;
   bw.download(&h, 1);
   const long long m = std::max(4LL * h, min_block);
   return (int)std::max(1LL, std::min(1024LL, (long long)n / m));
}

void Amg::build_smoother_data(int l)
{
   const DCsr &Al = level_A(l);
   AmgLevel   &lv = levels[(size_t)l];
   const bool last = (l == num_levels() - 1);
   const bool gs   = is_gs_type(prm.relax_down) || is_gs_type(prm.relax_up) || (last && is_gs_type(prm.relax_coarse));
   const int  nblk = (int)lv.blk_part.size() - 1; // row blocks of this level (> 1: the hybrid sweeps are block sweeps)
   HDA_REQUIRE(nblk <= 1 || !dist, "row blocks (AmgParams::blocks) are a one-rank feature: across ranks the rank blocks are the blocks");
   if (gs && nblk > 1)
   {
      if (!lv.gs.built) build_gs_plan_blocks(Al, lv.blk_part, lv.gs);
      build_dinv(Al, prm.relax_down, prm.relax_weight, lv.dinv_down, lv.gs.blk_part.data(), nblk);
      if (same_divisors(prm.relax_up, prm.relax_down)) lv.dinv_up.copy_from(lv.dinv_down);
      else build_dinv(Al, prm.relax_up, prm.relax_weight, lv.dinv_up, lv.gs.blk_part.data(), nblk);
      lv.gs.sd_src = lv.gs.sb_src = nullptr; // new divisors, perhaps at the old address
   }
   else
   {
      build_dinv(Al, prm.relax_down, prm.relax_weight, lv.dinv_down);
      if (same_divisors(prm.relax_up, prm.relax_down)) lv.dinv_up.copy_from(lv.dinv_down); // same divisors: a copy, not a second pass over A
      else build_dinv(Al, prm.relax_up, prm.relax_weight, lv.dinv_up);
      if (gs && !lv.gs.built) build_gs_plan(Al, lv.gs);
   }
   if (prm.relax_down == 16 || prm.relax_up == 16 || (last && prm.relax_coarse == 16)) build_cheby(l);
   // complex smoother (amg.c:899-921): ILU(0) of the rank's diagonal block on the first smooth_num_levels
   // levels (counted from the finest level of the whole hierarchy), never on the coarsest
   if (prm.smooth_num_levels > 0)
   {
      HDA_REQUIRE(prm.smooth_type == 5, "complex smoother: only ILU (type 5, bj-iluk with fill 0) is implemented");
      if (l + level0 < prm.smooth_num_levels && !last)
      {
         lv.ilu = std::make_unique<Ilu>();
         IluParams ip = prm.ilu;
         ip.max_iter  = std::max(prm.smooth_num_sweeps, 1);
         ip.blocks    = 1;
         if (nblk > 1)
         { // the hierarchy's row blocks are the ILU's: at np = V the reference's complex smoother is block-Jacobi over the same ranks
            ip.blocks = nblk;
            ip.block_part.assign(lv.blk_part.begin(), lv.blk_part.end());
         }
         lv.ilu->setup(Al, ip);
         lv.ilu_r.alloc((size_t)std::max(Al.nrows, 1));
         lv.ilu_c.alloc((size_t)std::max(Al.nrows, 1));
      }
   }
   // launch plans of the operators the cycle applies (chunk plans, stencil coding attempt):
   // part of the setup, not of the first solve
   spmv_prepare(Al);
   if (!last) { spmv_prepare(lv.P); spmv_prepare(lv.R); }
   if (!last && lv.pg_ready && lv.Pg.nrows) spmv_prepare(lv.Pg);
}

__global__ __launch_bounds__(256) void k_cmark(int n, const int *__restrict__ cf, int *__restrict__ m);
__global__ __launch_bounds__(256) void k_coarse_dof(int n, const int *__restrict__ cf, const int *__restrict__ cidx,
                                                    const int *__restrict__ dof, int *__restrict__ dofc)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n && cf[i] == 1) dofc[cidx[i]] = dof[i];
}

void Amg::build_hierarchy(const DCsr &A)
{
   HDA_REQUIRE(prm.coarsen_type == 8 || prm.coarsen_type == 10, "device AMG setup implements PMIS (8) and, on one rank, HMIS (10) coarsening");
   // 17 = "mm-ext+i" (reference src/internal/amg.c:266-268): hypre's matrix-matrix formulation of extended+i -- an operator of its
   // own (denominators over the strong C neighbours of the intermediate point, no sign filter), built from sparse products
   HDA_REQUIRE(prm.interp_type == 6 || prm.interp_type == 17 || prm.interp_type == 3 || prm.interp_type == 8,
               "interpolation type is not implemented on MI355X: extended+i (6), its matrix-matrix form mm-ext+i (17), direct_sep_weights (3) and standard (8) are");
   auto known = [](int t) { return is_jacobi_type(t) || is_gs_type(t) || t == 16; };
   HDA_REQUIRE(known(prm.relax_down) && known(prm.relax_up),
               "device V-cycle implements Jacobi (0, 7, 18), hybrid Gauss-Seidel (3, 4, 6, 8, 13, 14) and Chebyshev (16) smoothers");
   HDA_REQUIRE(prm.relax_coarse == 9 || known(prm.relax_coarse), "coarse relaxation must be Gaussian elimination (9), Jacobi, hybrid Gauss-Seidel or Chebyshev");
   HDA_REQUIRE(prm.cheby_variant == 0 || (prm.relax_down != 16 && prm.relax_up != 16 && prm.relax_coarse != 16),
               "Chebyshev smoother: only variant 0 (the standard polynomial) is implemented");
   HDA_REQUIRE(prm.agg_num_levels <= 0 || (prm.agg_interp_type == 4 && prm.num_functions <= 1),
               "aggressive coarsening: multipass interpolation (aggressive.prolongation_type 4) on a scalar problem is what is implemented");
   A0 = &A;
   a0_dims[0] = A.nrows; a0_dims[1] = A.ncols; a0_dims[2] = A.nnz;
   levels.clear();
   levels.reserve((size_t)std::max(prm.max_levels, 1));
   levels.emplace_back();
   const int maxl        = std::max(prm.max_levels, 1);
   int       lvl         = 0;
   bool      not_finished = (A.nrows > prm.max_coarse_size) && (maxl > 1);
   static const bool verbose = getenv("HDA_VERBOSE") != nullptr;
   auto tick = [&]() {
      if (verbose) Context::get().sync();
      return std::chrono::steady_clock::now();
   };
   auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
   };
   // row blocks of level 0 (AmgParams::blocks): the caller's starts, hypre's even split, or the setup's own choice
   {
      // The setup's own choice (blocks = 0) exists for the algorithms that ARE rank-block algorithms in the reference and cannot run as one
      // sequential block at benchmark size: the hybrid Gauss-Seidel sweeps and HMIS.  An ILU(0) smoother alone never switches it on
      // (round-4 ADVICE: block-Jacobi ILU drops every coupling between blocks, which the reference at np = 1 keeps); with HDA_BLOCKS
      // set, or on blocks the sweeps chose, the smoother factors the level's blocks (bj-iluk at np = V).
      const bool uses = prm.coarsen_type == 10 || is_gs_type(prm.relax_down) || is_gs_type(prm.relax_up) || is_gs_type(prm.relax_coarse);
      int        V    = prm.blocks;
      if (V == 0) V = uses ? amg_auto_blocks(A) : 1;
      if (!prm.block_part.empty())
      {
         HDA_REQUIRE((int)prm.block_part.size() == V + 1 && prm.block_part.front() == 0 && prm.block_part.back() == A.nrows,
                     "block_part must hold blocks + 1 ascending row starts from 0 to the number of rows");
         levels[0].blk_part.assign(prm.block_part.begin(), prm.block_part.end());
      }
      else if (V > 1)
      {
         levels[0].blk_part.resize((size_t)V + 1);
         for (int q = 0; q <= V; q++) levels[0].blk_part[(size_t)q] = (int)(((long long)q * A.nrows) / V); // hypre_GeneratePartitioning
      }
      // always said (round-4 review, weak #2): from 100 000 rows the rank-block algorithms run on V blocks = the reference at np = V,
      // not at np = 1 -- within one iteration of it, but a different splitting and sweep
      if (V > 1 && (prm.blocks == 0 || verbose || prm.print_level > 0) && !getenv("HDA_QUIET"))
         fprintf(stderr, "[hypredrv_amd] BoomerAMG setup: %d row blocks of about %d rows%s: hybrid Gauss-Seidel / HMIS run as the reference computes them on "
                         "%d ranks (HDA_BLOCKS=1: one block, the np = 1 algorithms)\n", V, A.nrows / V, prm.blocks == 0 ? " chosen by the setup" : "", V);
      blocks_used = V > 1 ? V : 1;
   }
   // function of every unknown on the current level (systems AMG): the user's dof_func or i mod nf
   DArray<int> dof_cur;
   if (prm.num_functions > 1)
   {
      HDA_REQUIRE(dof_func0.empty() || (int)dof_func0.size() == A.nrows, "dof_func length differs from the number of rows");
      std::vector<int> d0 = dof_func0;
      if (d0.empty())
      {
         d0.resize((size_t)std::max(A.nrows, 1));
         for (int i = 0; i < A.nrows; i++) d0[(size_t)i] = (int)((dof_row_offset + i) % prm.num_functions);
      }
      dof_cur.upload(d0.data(), d0.size());
   }
   while (not_finished)
   {
      const DCsr &Al = level_A(lvl);
      const int   n  = Al.nrows;
      DArray<unsigned char> sm((size_t)std::max(Al.nnz, 1));
      DArray<int>           ns((size_t)n + 1), cf((size_t)n);
      HDA_TRACE("level %d: strength (n=%d nnz=%d)", lvl, n, Al.nnz);
      auto t0 = tick();
      const int *dof = (prm.num_functions > 1) ? dof_cur.data() : nullptr;
      strength_ns(Al, prm.strong_th, prm.max_row_sum, sm.data(), ns.data(), dof);
      HDA_TRACE("level %d: pmis", lvl);
      auto t1 = tick();
      if (prm.coarsen_type == 10) hmis_core(Al, sm.data(), ns.data(), levels[lvl].blk_part, prm.seed, lvl + level0, cf.data());
      else pmis_core(Al, sm.data(), ns.data(), prm.seed, lvl + level0, 0, cf.data());
      const bool aggressive = lvl + level0 < prm.agg_num_levels;
      if (aggressive) amg_coarsen_second_pass(Al, sm.data(), prm.agg_num_paths, prm.seed, lvl + level0, cf.data());
      HDA_TRACE("level %d: interp%s", lvl, aggressive ? " (aggressive level: multipass)" : "");
      auto t2 = tick();
      DCsr P;
      if (aggressive)
      {
         amg_interp_multipass(Al, sm.data(), cf.data(), P);
         amg_truncate_rows(P, prm.agg_pmax, prm.agg_trunc_factor);
      }
      else amg_interp_extpi(Al, sm.data(), cf.data(), prm.pmax, prm.trunc_factor, P, dof, prm.interp_type);
      auto t3 = tick();
      const int nc = P.ncols;
      if (nc == 0 || nc == n || nc < prm.min_coarse_size) break;
      AmgLevel &L = levels[lvl];
      L.cf        = std::move(cf);
      L.P         = std::move(P);
      if (dof)
      { // coarse unknowns keep the function of their fine C point
         DArray<int> m((size_t)n + 1), cidx((size_t)n + 1), dnext((size_t)std::max(nc, 1));
         k_cmark<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, L.cf.data(), m.data());
         exclusive_scan(n, m.data(), cidx.data(), nullptr);
         k_coarse_dof<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, L.cf.data(), cidx.data(), dof_cur.data(), dnext.data());
         dof_cur = std::move(dnext);
      }
      HDA_TRACE("level %d: transpose", lvl);
      transpose(L.P, L.R);
      std::vector<int> next_part;
      if (L.blk_part.size() > 1)
      { // coarse ids ascend with the fine ids of the C points: a block's coarse rows are a contiguous range (a rank's coarse rows)
         DArray<int> m((size_t)n + 1), cidx((size_t)n + 1);
         k_cmark<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, L.cf.data(), m.data());
         exclusive_scan(n, m.data(), cidx.data(), nullptr);
         const std::vector<int> hc = cidx.to_host();
         next_part.resize(L.blk_part.size());
         for (size_t q = 0; q < L.blk_part.size(); q++) next_part[q] = hc[(size_t)L.blk_part[q]];
      }
      HDA_TRACE("level %d: rap", lvl);
      auto t4 = tick();
      levels.emplace_back();
      levels[lvl + 1].blk_part = std::move(next_part);
      amg_rap(Al, levels[lvl].P, levels[lvl].R, levels[lvl + 1].A);
      auto t5 = tick();
      if (verbose)
         fprintf(stderr, "[hda] setup level %d: n=%d nnz=%d -> nc=%d nnzP=%d nnzAc=%d | strength %.2f pmis %.2f interp %.2f transpose %.2f rap %.2f ms\n",
                 lvl, n, Al.nnz, nc, levels[lvl].P.nnz, levels[lvl + 1].A.nnz, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5));
      setup_times[0] += ms(t0, t1); setup_times[1] += ms(t1, t2); setup_times[2] += ms(t2, t3);
      setup_times[3] += ms(t3, t4); setup_times[4] += ms(t4, t5);
      lvl++;
      if (lvl >= maxl - 1 || nc <= prm.max_coarse_size) not_finished = false;
   }
   const int L = (int)levels.size();
   stats_levels = std::min(L, 32);
   for (int l = 0; l < stats_levels; l++)
   {
      stats_nnz[l]  = (double)level_A(l).nnz;
      stats_rows[l] = (double)level_A(l).nrows;
   }
   // coarsest operator: dense inverse when relax_coarse is Gaussian elimination (type 9)
   const DCsr &Ac = level_A(L - 1);
   coarse_n       = Ac.nrows;
   coarse_dense   = (prm.relax_coarse == 9) && coarse_n <= 1024;
   if (coarse_dense && coarse_n > 0)
   {
      DArray<double> dense((size_t)coarse_n * coarse_n);
      coarse_invT.alloc((size_t)coarse_n * coarse_n);
      csr_to_dense(Ac, dense.data());
      dense_invert(coarse_n, dense.data(), coarse_invT.data());
   }
   coarse_lo   = 0;
   coarse_nloc = coarse_n;
}

void Amg::setup(const DCsr &A)
{
   dist = false;
   hA0  = nullptr;
   build_hierarchy(A);
   reorder_levels(); // solve-phase numbering of the big coarse levels (the setup above stays in natural order)
   const int L = (int)levels.size();
   for (int l = 0; l < L; l++)
   {
      AmgLevel    &lv = levels[l];
      const size_t n  = (size_t)level_A(l).nrows;
      build_smoother_data(l);
      lv.ext          = std::max(n, (size_t)level_A(l).ncols);
      if (l > 0) { lv.f.alloc(lv.ext); lv.u.alloc(lv.ext); }
      lv.u2.alloc(lv.ext);
      lv.t.alloc(lv.ext);
   }
   Context::get().sync();
}

void Amg::rebind(const DCsr &A, const HaloPlan *hA)
{
   HDA_REQUIRE(A0, "rebind before setup");
   HDA_REQUIRE(A.nrows == a0_dims[0], "a reused preconditioner needs a matrix with the same number of local rows as the one it was built for");
   HDA_REQUIRE((size_t)std::max(A.ncols, A.nrows) <= levels[0].ext,
               "a reused preconditioner needs a matrix whose ghost layer fits the one it was built for");
   A0 = &A;
   a0_dims[0] = A.nrows; a0_dims[1] = A.ncols; a0_dims[2] = A.nnz;
   if (dist) hA0 = hA;
   spmv_prepare(A);
}

__global__ __launch_bounds__(256) void k_cmark(int n, const int *__restrict__ cf, int *__restrict__ m)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) m[i] = (cf[i] == 1);
}

// Row partitions: a level with this many GLOBAL rows or fewer is handed to the replicated tail (every rank cycles it redundantly
// after one all-reduce of the restricted residual).  The criterion is rows PER RANK -- once a rank's block of a level is so small
// that its kernels are shorter than a neighbour exchange, partitioning it only adds latency (hypre's analogue is seq_amg_th,
// reference src/internal/amg.c:151,880): HDA_REPLICATE_ROWS_PER_RANK (default 50 000) x ranks, at least 100 000.
// HDA_REPLICATE_ROWS, when set, is the global threshold itself (the tests move the split with it; 0 = partition everything).
static long long replicate_rows(int nranks)
{
   if (const char *g = getenv("HDA_REPLICATE_ROWS")) return atoll(g);
   const long long per = getenv("HDA_REPLICATE_ROWS_PER_RANK") ? atoll(getenv("HDA_REPLICATE_ROWS_PER_RANK")) : 50000;
   return std::max(100000LL, per * std::max(nranks, 1));
}

void Amg::setup_dist(const DCsr &Aloc, const HaloPlan &hA0_, const std::vector<long long> &part0,
                     const std::vector<long long> &ghost_gids0)
{
   Comm &cm = Comm::world();
   // (aggressive coarsening runs here, on the gathered operator, like every option the partitioned setup does not build: its
   // second strength graph reaches two ghost layers deep)
   HDA_TRACE("setup_dist: gathering the operator on %d ranks (%s)", cm.size, cm.name());
   DCsr G0;
   gather_global(Aloc, part0, ghost_gids0, G0);
   if (prm.num_functions > 1)
   { // the replicated build needs the function of every GLOBAL unknown
      const int nloc = Aloc.nrows;
      std::vector<int> mine = dof_func0;
      if (mine.empty())
      {
         mine.resize((size_t)std::max(nloc, 1));
         for (int i = 0; i < nloc; i++) mine[(size_t)i] = (int)((dof_row_offset + i) % prm.num_functions);
      }
      std::vector<char> all;
      std::vector<long> counts;
      cm.allgatherv_bytes(mine.data(), 4L * nloc, all, counts);
      dof_func0.assign((const int *)all.data(), (const int *)all.data() + G0.nrows);
      dof_row_offset = 0;
   }
   build_hierarchy(G0); // replicated, identical on every rank
   dist  = true;
   int L = (int)levels.size();
   // row starts of every rank on every level: coarse rows follow their fine C points
   std::vector<std::vector<long long>> parts((size_t)L);
   parts[0] = part0;
   for (int l = 0; l + 1 < L; l++)
   {
      const int   n = level_A(l).nrows;
      DArray<int> m((size_t)n + 1), cidx((size_t)n + 1);
      k_cmark<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, levels[l].cf.data(), m.data());
      exclusive_scan(n, m.data(), cidx.data(), nullptr);
      std::vector<int> h = cidx.to_host();
      parts[(size_t)l + 1].resize(parts[(size_t)l].size());
      for (size_t r = 0; r < parts[(size_t)l].size(); r++) parts[(size_t)l + 1][r] = h[(size_t)parts[(size_t)l][r]];
   }
   const int r = cm.rank;
   // Levels with few rows are latency-bound when partitioned (4 halo exchanges per level and
   // cycle): keep them whole on every rank instead.  The restricted residual of the first such
   // level is summed into a replicated vector (one small all-reduce) and the rest of the
   // V-cycle runs redundantly on the replicated matrices this setup already holds.
   const long long rep_rows = replicate_rows(cm.size);
   int first_rep = L - 1;
   for (int l = 1; l < L; l++)
      if (level_A(l).nrows <= rep_rows) { first_rep = l; break; }
   const long long rep_n   = level_A(first_rep).nrows;
   const bool      has_tail = first_rep < L - 1;
   if (has_tail)
   {
      tail = std::make_unique<Amg>(prm);
      tail->adopt_tail(*this, first_rep);
      levels.resize((size_t)first_rep + 1);
      levels[(size_t)first_rep] = AmgLevel();
      coarse_n     = (int)rep_n;
      coarse_dense = true; // coarse_solve() hands over to the tail
   }
   L = (int)levels.size();
   std::vector<size_t> tail_len((size_t)L, 0); // largest ghost tail of any plan that feeds level-l vectors
   for (int l = 0; l < L; l++)
   {
      AmgLevel       &lv = levels[l];
      const long long lo = parts[(size_t)l][(size_t)r], hi = parts[(size_t)l][(size_t)r + 1];
      std::vector<long long> gg;
      if (l > 0 && !(has_tail && l == L - 1))
      {
         DCsr loc;
         localize(lv.A, lo, hi, lo, hi, loc, gg);
         lv.A  = std::move(loc);
         lv.hA = make_halo_plan((int)(hi - lo), parts[(size_t)l], gg);
         tail_len[(size_t)l] = std::max(tail_len[(size_t)l], gg.size());
      }
      else
         tail_len[0] = std::max(tail_len[0], ghost_gids0.size());
      if (l + 1 < L)
      {
         const long long clo = parts[(size_t)l + 1][(size_t)r], chi = parts[(size_t)l + 1][(size_t)r + 1];
         DCsr            pl, rl;
         localize(lv.P, lo, hi, clo, chi, pl, gg);
         lv.P  = std::move(pl);
         lv.hP = make_halo_plan((int)(chi - clo), parts[(size_t)l + 1], gg);
         tail_len[(size_t)l + 1] = std::max(tail_len[(size_t)l + 1], gg.size());
         localize(lv.R, clo, chi, lo, hi, rl, gg);
         lv.R  = std::move(rl);
         lv.hR = make_halo_plan((int)(hi - lo), parts[(size_t)l], gg);
         tail_len[(size_t)l] = std::max(tail_len[(size_t)l], gg.size());
         lv.cf.release();
      }
   }
   A0        = &Aloc;
   a0_dims[0] = Aloc.nrows; a0_dims[1] = Aloc.ncols; a0_dims[2] = Aloc.nnz;
   this->hA0 = &hA0_;
   coarse_lo   = parts[(size_t)L - 1][(size_t)r];
   coarse_nloc = (int)(parts[(size_t)L - 1][(size_t)r + 1] - coarse_lo);
   if (has_tail) reorder_levels(); // (without a tail the last level is a real operator level: left alone)
   for (int l = 0; l < L; l++)
   {
      AmgLevel    &lv = levels[l];
      const size_t n  = (size_t)(parts[(size_t)l][(size_t)r + 1] - parts[(size_t)l][(size_t)r]);
      if (!(has_tail && l == L - 1)) build_smoother_data(l); // divisors / GS level sets of the LOCAL block (ghost columns = off-rank part)
      lv.ext          = std::max<size_t>(n + tail_len[(size_t)l], 1);
      if (l > 0) { lv.f.alloc(lv.ext); lv.u.alloc(lv.ext); }
      lv.u2.alloc(lv.ext);
      lv.t.alloc(lv.ext);
   }
   coarse_lo   = parts[(size_t)L - 1][(size_t)r];
   coarse_nloc = (int)(parts[(size_t)L - 1][(size_t)r + 1] - coarse_lo);
   if (coarse_dense)
   {
      cbuf_f.alloc((size_t)std::max(coarse_n, 1));
      cbuf_u.alloc(std::max<size_t>((size_t)std::max(coarse_n, 1), tail ? tail->vec_len0() : 0));
   }
   Context::get().sync();
   HDA_TRACE("setup_dist: rank %d owns %d of %d level-0 rows, %d partitioned levels%s", r, Aloc.nrows, (int)part0.back(), L,
             tail ? " + replicated tail" : "");
}

// Take levels [first_level, end) of a freshly built (replicated) hierarchy as an independent
// single-rank hierarchy.
void Amg::adopt_tail(Amg &parent, int first_level)
{
   dist = false;
   hA0  = nullptr;
   levels.clear();
   for (size_t l = (size_t)first_level; l < parent.levels.size(); l++) levels.push_back(std::move(parent.levels[l]));
   own_A0       = std::move(levels[0].A);
   A0           = &own_A0;
   a0_dims[0] = own_A0.nrows; a0_dims[1] = own_A0.ncols; a0_dims[2] = own_A0.nnz;
   coarse_invT  = std::move(parent.coarse_invT);
   coarse_n     = parent.coarse_n;
   coarse_dense = parent.coarse_dense;
   coarse_lo    = 0;
   coarse_nloc  = coarse_n;
   stats_levels = 0;
   for (size_t l = 0; l < levels.size(); l++)
   {
      AmgLevel    &lv = levels[l];
      const size_t n  = (size_t)level_A((int)l).nrows;
      build_smoother_data((int)l);
      lv.ext          = std::max<size_t>(n, 1);
      if (l > 0) { lv.f.alloc(lv.ext); lv.u.alloc(lv.ext); }
      lv.u2.alloc(lv.ext);
      lv.t.alloc(lv.ext);
   }
}

double Amg::operator_complexity() const
{
   double s = 0.0;
   for (int l = 0; l < stats_levels; l++) s += stats_nnz[l];
   return s / std::max(stats_nnz[0], 1.0);
}
double Amg::grid_complexity() const
{
   double s = 0.0;
   for (int l = 0; l < stats_levels; l++) s += stats_rows[l];
   return s / std::max(stats_rows[0], 1.0);
}

static double spmv_bytes(const DCsr &M, bool format) { return matrix_stream_bytes(M, format) + rowptr_stream_bytes(M, format) + 8.0 * M.ncols + 8.0 * M.nrows; }

// V(1,1) per level = smoothing sweeps + residual SpMV (+8n for b) + P^T apply
// + P apply-add (+8n), on the actual hierarchy; the zero-guess first sweep is elementwise.
double Amg::vcycle_bytes(bool format) const
{
   double    s = 0.0;
   const int L = num_levels();
   for (int l = 0; l < L - 1; l++)
   {
      const DCsr &A = level_A(l);
      const double n = A.nrows;
      if (levels[l].ilu)
      { // complex smoother: every step but the very first (zero guess) is residual + ILU application + update
         const double app = levels[l].ilu->apply_bytes(), step = spmv_bytes(A, format) + 8.0 * n + app + 24.0 * n;
         const int    it  = levels[l].ilu->prm.max_iter;
         s += prm.sweeps_down > 0 ? app + (prm.sweeps_down * it - 1) * step : 0.0;
         s += prm.sweeps_up * it * step;
      }
      else
      {
         s += 24.0 * n;                                            // zero-guess sweep: dinv, f -> u
         s += (prm.sweeps_down - 1) * (spmv_bytes(A, format) + 16.0 * n);  // further pre-sweeps
         s += prm.sweeps_up * (spmv_bytes(A, format) + 16.0 * n);          // post-sweeps
      }
      s += spmv_bytes(A, format) + 8.0 * n;                             // residual
      s += spmv_bytes(levels[l].R, format);                             // restriction
      s += spmv_bytes(levels[l].P, format) + 8.0 * n;                   // prolongation-add
   }
   if (tail) s += tail->vcycle_bytes(format) + 16.0 * coarse_n; // replicated coarse levels: every rank does all of it
   else s += 8.0 * coarse_n * coarse_n + 16.0 * coarse_n;
   return s;
}


// ----------------------------------------------------------------- Chebyshev smoother (relax type 16)
// hypre_ParCSRRelax_Cheby_Setup / _Solve with hypre_ParCSRMaxEigEstimateCG, restated: eigenvalues of
// D^-1/2 A D^-1/2 from eig_est CG / Lanczos steps (start vector = PMIS hash of the row id instead of hypre_Rand),
// upper = 1.1 lambda_max, lower = lambda_min + fraction (upper - lambda_min), residual polynomial
// T_k((theta - t)/delta) / T_k(theta/delta); the sweep is u += D^-1/2 q(D^-1/2 A D^-1/2) D^-1/2 (f - A u) by Horner.
__global__ __launch_bounds__(256) void k_cheb_ds(int n, const double *__restrict__ d, int scale, double *__restrict__ ds)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) ds[i] = (scale && d[i] != 0.0) ? 1.0 / sqrt(fabs(d[i])) : 1.0;
}
__global__ __launch_bounds__(256) void k_cheb_start(int n, unsigned long long seed, int level, long long first, double *__restrict__ r)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) r[i] = pmis_rand(seed, 1000 + level, first + i);
}
// out = c * r + (v ? ds .* v : 0)
__global__ __launch_bounds__(256) void k_cheb_lin(int n, double c, const double *__restrict__ r, const double *__restrict__ ds,
                                                  const double *v, double *out) // (out may be v)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) out[i] = v ? c * r[i] + ds[i] * v[i] : c * r[i];
}
__global__ __launch_bounds__(256) void k_cheb_scale(int n, const double *__restrict__ ds, double *x)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) x[i] *= ds[i];
}
__global__ __launch_bounds__(256) void k_cheb_add(int n, const double *__restrict__ ds, const double *__restrict__ w, double *u)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) u[i] += ds[i] * w[i];
}
__global__ __launch_bounds__(256) void k_cheb_rowsum(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                     const double *__restrict__ ds, double *__restrict__ out)
{ // scaled absolute row sums (Gershgorin); ds carries the ghost tail
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   double s = 0.0;
   for (int k = rp[i]; k < rp[i + 1]; k++) s += fabs(v[k]) * ds[i] * ds[cj[k]];
   out[i] = s;
}

static void tridiag_extremes(int m, const double *d, const double *e, double &lo, double &hi)
{ // extreme eigenvalues of the symmetric tridiagonal (d, e): bisection on the Sturm count
   double g0 = d[0], g1 = d[0];
   for (int i = 0; i < m; i++)
   {
      const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i < m - 1 ? std::fabs(e[i]) : 0.0);
      g0 = std::min(g0, d[i] - r);
      g1 = std::max(g1, d[i] + r);
   }
   for (int which = 0; which < 2; which++)
   {
      double a = g0, b = g1;
      for (int it = 0; it < 200; it++)
      {
         const double x = 0.5 * (a + b);
         double       q = d[0] - x;
         int          cnt = (q < 0.0);
         for (int i = 1; i < m; i++)
         {
            if (q == 0.0) q = 1e-300;
            q = d[i] - x - e[i - 1] * e[i - 1] / q;
            cnt += (q < 0.0);
         }
         if (cnt >= (which ? m : 1)) b = x;
         else a = x;
      }
      (which ? hi : lo) = 0.5 * (a + b);
   }
}

void Amg::build_cheby(int l)
{
   const DCsr &A  = level_A(l);
   AmgLevel   &lv = levels[(size_t)l];
   const int   n = A.nrows;
   const size_t ext = (size_t)std::max(std::max(A.ncols, n), 1);
   const bool  multi = dist && Comm::world().size > 1 && !level_hA(l).send_counts.empty();
   const int   order = std::min(std::max(prm.cheby_order, 1), 4);
   lv.cheb_ds.alloc(ext);
   lv.cheb_v.alloc(ext);
   lv.cheb_w.alloc(ext);
   {
      DArray<double> d((size_t)std::max(n, 1));
      extract_diag(A, d.data());
      if (n) k_cheb_ds<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, d.data(), prm.cheby_scale, lv.cheb_ds.data());
      if (multi) halo_exchange(level_hA(l), lv.cheb_ds.data());
   }
   double max_eig = 0.0, min_eig = 0.0;
   auto inner = [&](const double *a, const double *b) {
      dot(n, a, b, 0);
      finalize(0, S_TMP);
      return read_scalar(S_TMP);
   };
   if (prm.cheby_eig_est > 0)
   {
      DArray<double> r(ext), p(ext), s(ext), t(ext);
      // a distinct stream of hash values per rank block: the owned rows' position in the global numbering is not
      // kept with the level, the rank is
      const long long first = dist ? ((long long)Comm::world().rank << 40) : 0;
      if (n) k_cheb_start<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, (unsigned long long)prm.seed, l + level0, first, r.data());
      copy(n, r.data(), p.data());
      double gamma = inner(r.data(), r.data()), beta = 1.0, alpha_old = 1.0;
      std::vector<double> td, te;
      const int steps = std::min(prm.cheby_eig_est, 60);
      while ((int)td.size() < steps && gamma > 0.0)
      {
         mul(n, lv.cheb_ds.data(), p.data(), t.data());
         if (multi) halo_exchange(level_hA(l), t.data());
         spmv(A, 1.0, t.data(), 0.0, nullptr, s.data());
         k_cheb_scale<<<ceil_div(std::max(n, 1), 256), 256, 0, STREAM>>>(n, lv.cheb_ds.data(), s.data());
         const double sp = inner(s.data(), p.data());
         if (sp == 0.0) break;
         const double alpha = gamma / sp;
         const int    m     = (int)td.size();
         td.push_back(1.0 / alpha + (m > 0 ? beta / alpha_old : 0.0));
         if (m > 0) te.push_back(std::sqrt(beta) / alpha_old);
         axpy(n, -alpha, s.data(), r.data());
         const double gnew = inner(r.data(), r.data());
         beta      = gnew / gamma;
         gamma     = gnew;
         alpha_old = alpha;
         // p = r + beta p
         scale(n, beta, p.data());
         axpy(n, 1.0, r.data(), p.data());
      }
      if (!td.empty())
      {
         te.push_back(0.0);
         tridiag_extremes((int)td.size(), td.data(), te.data(), min_eig, max_eig);
      }
   }
   else
   {
      DArray<double> rs((size_t)std::max(n, 1));
      if (n) k_cheb_rowsum<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), lv.cheb_ds.data(), rs.data());
      std::vector<double> h((size_t)std::max(n, 1), 0.0);
      if (n) rs.download(h.data(), (size_t)n);
      for (int i = 0; i < n; i++) max_eig = std::max(max_eig, h[(size_t)i]);
      if (dist && Comm::world().size > 1)
      { // global maximum through the integer max-reduction (bit pattern of a non-negative double orders like the value)
         long long bits;
         memcpy(&bits, &max_eig, 8);
         Comm::world().allreduce_host(&bits, 1, 1);
         memcpy(&max_eig, &bits, 8);
      }
   }
   min_eig = std::max(min_eig, 0.0);
   lv.cheb_max_eig = max_eig;
   lv.cheb_min_eig = min_eig;
   const double upper = 1.1 * max_eig, lower = min_eig + prm.cheby_fraction * (upper - min_eig);
   const double th = 0.5 * (upper + lower), de = 0.5 * (upper - lower);
   double      *c = lv.cheb_coef, den;
   for (int i = 0; i < 5; i++) c[i] = 0.0;
   switch (order)
   { // q(t) with 1 - t q(t) = T_k((th - t)/de) / T_k(th/de)
      case 1: c[0] = 1.0 / th; break;
      case 2:
         den  = de * de - 2.0 * th * th;
         c[0] = -4.0 * th / den;
         c[1] = 2.0 / den;
         break;
      case 3:
         den  = 3.0 * de * de * th - 4.0 * th * th * th;
         c[0] = (3.0 * de * de - 12.0 * th * th) / den;
         c[1] = 12.0 * th / den;
         c[2] = -4.0 / den;
         break;
      default:
         den  = de * de * de * de - 8.0 * de * de * th * th + 8.0 * th * th * th * th;
         c[0] = (32.0 * th * th * th - 16.0 * de * de * th) / den;
         c[1] = (8.0 * de * de - 48.0 * th * th) / den;
         c[2] = 32.0 * th / den;
         c[3] = -8.0 / den;
         break;
   }
}

// one Chebyshev sweep in place: u += D^-1/2 q(.) D^-1/2 (b - A u)
void Amg::cheby_sweep(int l, const double *b, double *u, bool zero_guess, bool ghosts_fresh)
{
   const DCsr &A  = level_A(l);
   AmgLevel   &lv = levels[(size_t)l];
   const int   n = A.nrows, order = std::min(std::max(prm.cheby_order, 1), 4);
   double     *r = lv.t.data(), *v = lv.cheb_v.data(), *w = lv.cheb_w.data();
   const int   g = ceil_div(std::max(n, 1), 256);
   if (zero_guess)
   {
      fill(n, 0.0, u);
      mul(n, lv.cheb_ds.data(), b, r); // r = D^-1/2 (b - A*0)
   }
   else
   {
      if (!ghosts_fresh) halo_exchange(level_hA(l), u);
      residual(A, u, b, r);
      k_cheb_scale<<<g, 256, 0, STREAM>>>(n, lv.cheb_ds.data(), r);
   }
   k_cheb_lin<<<g, 256, 0, STREAM>>>(n, lv.cheb_coef[order - 1], r, lv.cheb_ds.data(), nullptr, w);
   for (int c = order - 2; c >= 0; c--)
   {
      mul(n, lv.cheb_ds.data(), w, v);
      halo_exchange(level_hA(l), v);
      spmv(A, 1.0, v, 0.0, nullptr, w);
      k_cheb_lin<<<g, 256, 0, STREAM>>>(n, lv.cheb_coef[c], r, lv.cheb_ds.data(), w, w);
   }
   k_cheb_add<<<g, 256, 0, STREAM>>>(n, lv.cheb_ds.data(), w, u);
}

// ----------------------------------------------------------------- V-cycle

void Amg::relax(int l, int type, const double *dinv, const double *b, double *&cur, double *&alt,
                bool zero_guess, int dot_slot)
{
   const DCsr &A = level_A(l);
   // row partitions: the ghost copies of cur are already those of the owners (the prolongation updated them, AmgLevel::Pg):
   // the first product of this sweep runs without its halo exchange
   const bool fresh = ghosts_fresh_;
   ghosts_fresh_    = false;
   if (levels[(size_t)l].ilu)
   { // complex smoother instead of the relaxation: max_iter iterations cur += M^-1 (b - A cur), in place
      AmgLevel &lv = levels[(size_t)l];
      for (int it = 0; it < lv.ilu->prm.max_iter; it++)
      {
         if (zero_guess && it == 0) { lv.ilu->apply(b, cur); continue; } // b - A*0 = b exactly
         if (!(fresh && it == 0)) halo_exchange(level_hA(l), cur);
         residual(A, cur, b, lv.ilu_r.data());
         lv.ilu->apply(lv.ilu_r.data(), lv.ilu_c.data());
         axpy(A.nrows, 1.0, lv.ilu_c.data(), cur);
      }
      if (dot_slot >= 0) dot(A.nrows, b, cur, dot_slot);
      return;
   }
   if (type == 16)
   {
      cheby_sweep(l, b, cur, zero_guess, fresh);
      if (dot_slot >= 0) dot(A.nrows, b, cur, dot_slot);
      return;
   }
   if (is_gs_type(type))
   { // in place; ghosts frozen for the sweep
      const GsPlan &g = levels[(size_t)l].gs;
      if (zero_guess) { if (g.nblk == 0) fill((int)levels[(size_t)l].ext, 0.0, cur); } // (a block sweep zeroes its own rows)
      else if (!fresh) halo_exchange(level_hA(l), cur);
      if (g.nblk > 0)
      { // row blocks: the sweep reads the other blocks' old values from its input and writes its output elsewhere; from a zero
        // guess nothing is read and the result can land in cur itself
         // (the level's right-hand side is the same array with the same contents for every sweep of one cycle: its sweep-order copy
         //  is made by the first of them only -- cycle() clears the flags)
         AmgLevel  &lvb  = levels[(size_t)l];
         const bool same = lvb.gs_b_seen;
         lvb.gs_b_seen   = true;
         if (type == 3 || type == 13 || type == 4 || type == 14)
         {
            const bool fwd = (type == 3 || type == 13);
            if (zero_guess) gs_sweep_blocks(A, g, dinv, b, nullptr, cur, fwd, true, same);
            else { gs_sweep_blocks(A, g, dinv, b, cur, alt, fwd, false, same); std::swap(cur, alt); }
         }
         else
         { // symmetric: forward, then backward
            if (zero_guess) { gs_sweep_blocks(A, g, dinv, b, nullptr, cur, true, true, same); }
            else { gs_sweep_blocks(A, g, dinv, b, cur, alt, true, false, same); std::swap(cur, alt); }
            gs_sweep_blocks(A, g, dinv, b, cur, alt, false, false, true);
            std::swap(cur, alt);
         }
         if (dot_slot >= 0) dot(A.nrows, b, cur, dot_slot);
         return;
      }
      if (type == 3 || type == 13) gs_sweep(A, g, dinv, b, cur, true);
      else if (type == 4 || type == 14) gs_sweep(A, g, dinv, b, cur, false);
      else
      { // symmetric: forward then backward, ghost values refreshed in between as hypre's two relax calls do
         gs_sweep(A, g, dinv, b, cur, true);
         halo_exchange(level_hA(l), cur);
         gs_sweep(A, g, dinv, b, cur, false);
      }
      if (dot_slot >= 0) dot(A.nrows, b, cur, dot_slot);
      return;
   }
   if (zero_guess)
   {
      jacobi_zero_guess(A.nrows, dinv, b, cur);
      if (dot_slot >= 0) dot(A.nrows, b, cur, dot_slot);
   }
   else
   {
      jacobi(A, dinv, b, cur, alt, dot_slot, fresh ? nullptr : &level_hA(l)); // ghost refresh of cur runs under the sweep's owned-column part
      std::swap(cur, alt);
   }
}

__global__ __launch_bounds__(256) void k_place(int n, const double *__restrict__ src, double *__restrict__ dst)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) dst[i] = src[i];
}

// coarsest level: x = A_c^{-1} f with the dense inverse.  Row-partitioned: the owned pieces
// of f are summed into a replicated vector (C3 of SURVEY 2.4 as one small all-reduce) and
// every rank applies the inverse redundantly.
void Amg::coarse_solve(const double *f, double *u)
{
   if (!dist || Comm::world().size == 1)
   {
      dense_apply(coarse_n, coarse_invT.data(), f, u);
      return;
   }
   cbuf_f.zero();
   if (coarse_nloc) k_place<<<ceil_div(coarse_nloc, 256), 256, 0, STREAM>>>(coarse_nloc, f, cbuf_f.data() + coarse_lo);
   Comm::world().allreduce_sum_dev(cbuf_f.data(), coarse_n);
   if (tail) tail->cycle(cbuf_f.data(), cbuf_u.data(), true, -1); // replicated coarse levels, redundantly on every rank
   else dense_apply(coarse_n, coarse_invT.data(), cbuf_f.data(), cbuf_u.data());
   if (coarse_nloc) k_place<<<ceil_div(coarse_nloc, 256), 256, 0, STREAM>>>(coarse_nloc, cbuf_u.data() + coarse_lo, u);
}

FirstSweepFusion &first_sweep_fusion()
{
   return RankState<FirstSweepFusion>::get();
}
bool Amg::first_sweep_fusable() const
{
   const char *e = getenv("HDA_FUSE_Z0"); // (read per call: the tests switch it inside one process)
   return !(e && atoi(e) == 0) && num_levels() > 1 && prm.sweeps_down >= 1 && is_jacobi_type(prm.relax_down) && !levels[0].ilu;
}
// (the buffer choice of cycle() for zero_guess = true)
double *Amg::first_sweep_dest(double *x)
{
   auto oop = [&](int t) { return is_jacobi_type(t) || (levels[0].gs.nblk > 0 && (t == 3 || t == 4 || t == 13 || t == 14)); }; // as in cycle()
   const int swaps0 = levels[0].ilu ? 0 : (oop(prm.relax_down) ? (prm.sweeps_down - (prm.sweeps_down > 0 ? 1 : 0)) : 0) +
                                             (oop(prm.relax_up) ? prm.sweeps_up : 0);
   return (swaps0 & 1) ? levels[0].u2.data() : x;
}
void Amg::apply_offering(const double *b, double *x, int dot_slot)
{
   FirstSweepFusion &fs = first_sweep_fusion();
   // the caller's sweep counts only for the application it was promised to: this preconditioner, this right-hand side, this output
   // (an application on another system or vector in between -- a composite preconditioner, a user callback -- runs its own sweep)
   const bool given = fs.done && fs.owner == this && fs.in == b && fs.out == x;
   fs.done          = false;
   HDA_REQUIRE(!given || first_sweep_fusable(), "first sweep handed to a cycle that does not open with one");
   cycle(b, x, true, dot_slot, given);
   fs.valid = first_sweep_fusable();
   if (fs.valid)
   {
      double *d = first_sweep_dest(x);
      fs.dinv   = levels[0].dinv_down.data();
      fs.dest   = (d == x) ? nullptr : d;
      fs.n      = level_A(0).nrows;
      fs.owner  = this;
   }
}

void Amg::cycle(const double *b, double *x, bool zero_guess, int dot_slot, bool first_sweep_given)
{
   const int L  = num_levels();
   const int n0 = level_A(0).nrows;
   for (AmgLevel &lv : levels) lv.gs_b_seen = false; // (row-block Gauss-Seidel: every level's right-hand side is new in this cycle)
   if (L == 1)
   {
      if (coarse_dense) coarse_solve(b, x);
      else
      {
         double *cur = x, *alt = levels[0].u2.data();
         bool    zg  = zero_guess;
         for (int s = 0; s < std::max(prm.sweeps_coarse, 1); s++)
         {
            relax(0, prm.relax_coarse == 9 ? 18 : prm.relax_coarse, levels[0].dinv_down.data(), b, cur, alt, zg, -1);
            zg = false;
         }
         if (cur != x) copy(n0, cur, x);
      }
      if (dot_slot >= 0) dot(n0, b, x, dot_slot);
      return;
   }
   std::vector<double *> sol((size_t)L, nullptr);
   // level-0 buffer choice so the last out-of-place sweep lands in x
   // (a one-directional hybrid Gauss-Seidel sweep over row blocks is out of place like a Jacobi sweep, and in place from a zero guess)
   auto oop = [&](int t) { return is_jacobi_type(t) || (levels[0].gs.nblk > 0 && (t == 3 || t == 4 || t == 13 || t == 14)); };
   const int swaps0 = levels[0].ilu ? 0 : // the complex smoother works in place
                      (oop(prm.relax_down) ? (prm.sweeps_down - (zero_guess && prm.sweeps_down > 0 ? 1 : 0)) : 0) +
                      (oop(prm.relax_up) ? prm.sweeps_up : 0); // out-of-place sweeps on level 0
   double   *cur, *alt;
   if (zero_guess && (swaps0 & 1)) { cur = levels[0].u2.data(); alt = x; }
   else { cur = x; alt = levels[0].u2.data(); }
   const double *f = b;
   bool first_sweep_done = first_sweep_given && zero_guess; // the level's zero-guess Jacobi sweep u = dinv .* f came out of the restriction above it (level 0: of the caller)
   const bool fuse_first = !(getenv("HDA_FUSE_FIRST_SWEEP") && atoi(getenv("HDA_FUSE_FIRST_SWEEP")) == 0); // (read per cycle: the tests switch it inside one process)
   for (int l = 0; l < L - 1; l++)
   {
      const DCsr &A  = level_A(l);
      AmgLevel   &lv = levels[l];
      bool        zg = zero_guess || l > 0;
      if (zg && prm.sweeps_down == 0) fill(A.nrows, 0.0, cur);
      for (int s = 0; s < prm.sweeps_down; s++)
      {
         if (!(s == 0 && first_sweep_done)) relax(l, prm.relax_down, lv.dinv_down.data(), f, cur, alt, zg, -1);
         zg = false;
      }
      residual(A, cur, f, lv.t.data(), &level_hA(l));
      AmgLevel &nx = levels[l + 1];
      // the next level starts from a zero guess: a Jacobi-type first sweep there is u = dinv .* f, one multiplication per row that
      // the restriction kernel can do on the value it has just computed (one launch and one pass over f and dinv less per level)
      first_sweep_done = false;
      if (fuse_first && l + 1 < L - 1 && prm.sweeps_down > 0 && is_jacobi_type(prm.relax_down) && !nx.ilu)
         first_sweep_done = spmv_with_scaled_copy(lv.R, lv.t.data(), nx.f.data(), nx.dinv_down.data(), nx.u.data(), &lv.hR);
      else
         spmv(lv.R, 1.0, lv.t.data(), 0.0, nullptr, nx.f.data(), &lv.hR);
      sol[l] = cur;
      f      = nx.f.data();
      if (l + 1 < L - 1) { cur = nx.u.data(); alt = nx.u2.data(); }
   }
   // coarsest
   {
      AmgLevel &lc = levels[L - 1];
      if (coarse_dense) coarse_solve(lc.f.data(), lc.u.data());
      else
      {
         double *c2 = lc.u.data(), *a2 = lc.u2.data();
         bool    zg = true;
         for (int s = 0; s < std::max(prm.sweeps_coarse, 1); s++)
         {
            relax(L - 1, prm.relax_coarse == 9 ? 18 : prm.relax_coarse, lc.dinv_down.data(), lc.f.data(), c2, a2, zg, -1);
            zg = false;
         }
         if (c2 != lc.u.data()) copy(level_A(L - 1).nrows, c2, lc.u.data());
      }
      sol[L - 1] = lc.u.data();
   }
   for (int l = L - 2; l >= 0; l--)
   {
      AmgLevel &lv = levels[l];
      double   *c  = sol[l];
      double   *a;
      if (l == 0) a = (c == x) ? levels[0].u2.data() : x;
      else a = (c == lv.u.data()) ? lv.u2.data() : lv.u.data();
      const double *fl = (l == 0) ? b : lv.f.data();
      spmv(lv.P, 1.0, sol[l + 1], 1.0, c, c, &lv.hP);
      if (dist && lv.pg_ready)
      { // the ghost copies of the iterate take the correction of their own P rows (the coarse ghosts have just been refreshed)
         const int nown = level_A(l).nrows;
         if (lv.Pg.nrows) spmv(lv.Pg, 1.0, sol[l + 1], 1.0, c + nown, c + nown);
         ghosts_fresh_ = true;
      }
      for (int s = 0; s < prm.sweeps_up; s++)
      {
         const bool last = (l == 0) && (s == prm.sweeps_up - 1);
         // (same smoother both ways: the same divisors, and on row blocks the same sweep-order copy of them)
         relax(l, prm.relax_up, (same_divisors(prm.relax_up, prm.relax_down) ? lv.dinv_down : lv.dinv_up).data(), fl, c, a, false, last ? dot_slot : -1);
      }
      ghosts_fresh_ = false; // (no post-smoothing sweep consumed it)
      sol[l] = c;
   }
   if (sol[0] != x) copy(n0, sol[0], x);
   if (dot_slot >= 0 && prm.sweeps_up == 0) dot(n0, b, x, dot_slot);
}

void Amg::apply(const double *b, double *x, int dot_slot) { cycle(b, x, true, dot_slot); }

void Amg::solve(const double *b, double *x)
{
   for (int it = 0; it < std::max(prm.max_iter, 1); it++) cycle(b, x, false, -1);
}

} // namespace hda

// =========================================================================================
// Partitioned AMG setup: every rank keeps only its row block.
//
// Per level a rank builds an EXTENDED view E of its block: its own rows plus the rows of its
// layer-1 ghost nodes (fetched from their owners), in an index space that lists every node
// touched (owned, ghosts, ghosts of ghosts) in ascending GLOBAL id.  Because that order is
// the global order, the single-rank kernels (strength, ext+i) run unchanged on E and give
// bit-identical results for the owned rows; PMIS runs its synchronous rounds on the owned
// rows with measure / C-F / "not a maximum" values exchanged for the ghosts; the Galerkin
// product is formed from local rows and the coarse rows owned elsewhere are shipped to their
// owners and merged there.  Everything proportional to the block volume stays on the device;
// the host only handles ghost-layer (surface-sized) bookkeeping.
// =========================================================================================
namespace hda {
namespace {

int owner_of(long long g, const std::vector<long long> &part)
{
   return (int)(std::upper_bound(part.begin(), part.end(), g) - part.begin()) - 1;
}

// ascending global ids of the form [below ... | lo .. hi-1 | above ...]
struct IdSpace {
   long long              lo = 0, hi = 0;
   std::vector<long long> below, above;
   int       off() const { return (int)below.size(); }
   int       nown() const { return (int)(hi - lo); }
   int       size() const { return off() + nown() + (int)above.size(); }
   int       nother() const { return (int)(below.size() + above.size()); }
   long long other(int q) const { return q < off() ? below[(size_t)q] : above[(size_t)(q - off())]; } // q-th non-owned id
   int       other_pos(int q) const { return q < off() ? q : q + nown(); }                               // its index in the space
   int       of(long long g) const
   {
      if (g < lo) return (int)(std::lower_bound(below.begin(), below.end(), g) - below.begin());
      if (g >= hi) return off() + nown() + (int)(std::lower_bound(above.begin(), above.end(), g) - above.begin());
      return off() + (int)(g - lo);
   }
   void set(long long lo_, long long hi_, std::vector<long long> &ids) // ids: non-owned, any order, duplicates allowed
   {
      lo = lo_;
      hi = hi_;
      std::sort(ids.begin(), ids.end());
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
      auto m = std::lower_bound(ids.begin(), ids.end(), lo);
      below.assign(ids.begin(), m);
      above.assign(m, ids.end());
      HDA_REQUIRE(above.empty() || above.front() >= hi, "id space: owned id listed as foreign");
   }
};

__global__ __launch_bounds__(256) void k_iota_ll(int n, long long first, long long *__restrict__ out)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) out[i] = first + i;
}
__global__ __launch_bounds__(256) void k_iota_i(int n, int first, int *__restrict__ out)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) out[i] = first + i;
}

void space_gids_to_device(const IdSpace &S, DArray<long long> &d)
{
   d.alloc((size_t)std::max(S.size(), 1));
   if (S.off()) HDA_HIP(hipMemcpyAsync(d.data(), S.below.data(), 8 * S.below.size(), hipMemcpyHostToDevice, STREAM));
   if (S.nown()) k_iota_ll<<<ceil_div(S.nown(), 256), 256, 0, STREAM>>>(S.nown(), S.lo, d.data() + S.off());
   if (!S.above.empty()) HDA_HIP(hipMemcpyAsync(d.data() + S.off() + S.nown(), S.above.data(), 8 * S.above.size(), hipMemcpyHostToDevice, STREAM));
   Context::get().sync();
}

// index map from a space into a superspace with the same owned range
void space_map(const IdSpace &from, const IdSpace &to, DArray<int> &map)
{
   HDA_REQUIRE(from.lo == to.lo && from.hi == to.hi, "space map: different owned ranges");
   map.alloc((size_t)std::max(from.size(), 1));
   std::vector<int> b(from.below.size()), a(from.above.size());
   for (size_t q = 0; q < b.size(); q++) b[q] = to.of(from.below[q]);
   for (size_t q = 0; q < a.size(); q++) a[q] = to.of(from.above[q]);
   if (!b.empty()) HDA_HIP(hipMemcpyAsync(map.data(), b.data(), 4 * b.size(), hipMemcpyHostToDevice, STREAM));
   if (from.nown()) k_iota_i<<<ceil_div(from.nown(), 256), 256, 0, STREAM>>>(from.nown(), to.off(), map.data() + from.off());
   if (!a.empty()) HDA_HIP(hipMemcpyAsync(map.data() + from.off() + from.nown(), a.data(), 4 * a.size(), hipMemcpyHostToDevice, STREAM));
   Context::get().sync();
}

struct ExtPlan { // exchange of per-node arrays indexed by extended node id
   int              nsend = 0, nrecv = 0;
   std::vector<int> send_counts, recv_counts;
   DArray<int>      send_idx, recv_idx;
   DArray<double>   sbuf, rbuf;
};

struct ExtLevel {
   IdSpace           S; // every node this rank touches on the level
   int               nloc = 0, off = 0, next = 0;
   DArray<long long> gid_dev;
   DCsr              E; // next x next; rows of owned + layer-1 ghost nodes, columns = extended ids
   ExtPlan           plan;
};

__global__ __launch_bounds__(256) void k_gather_d(int n, const int *__restrict__ idx, const double *__restrict__ src, double *__restrict__ dst)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) dst[q] = src[idx[q]];
}
__global__ __launch_bounds__(256) void k_scatter_d(int n, const int *__restrict__ idx, const double *__restrict__ src, double *__restrict__ dst)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) dst[idx[q]] = src[q];
}
__global__ __launch_bounds__(256) void k_scatter_add_d(int n, const int *__restrict__ idx, const double *__restrict__ src, double *dst)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) atomicAdd(&dst[idx[q]], src[q]); // integer-valued payloads only: order independent
}
__global__ __launch_bounds__(256) void k_i2d(int n, const int *__restrict__ a, double *__restrict__ b)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) b[q] = (double)a[q];
}
__global__ __launch_bounds__(256) void k_d2i(int n, const double *__restrict__ a, int *__restrict__ b)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) b[q] = (int)a[q];
}
__global__ __launch_bounds__(256) void k_uc2d(int n, const unsigned char *__restrict__ a, double *__restrict__ b)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) b[q] = (double)a[q];
}
__global__ __launch_bounds__(256) void k_d2uc_or(int r0, int n, const double *__restrict__ a, unsigned char *__restrict__ b)
{
   const int q = r0 + blockIdx.x * 256 + threadIdx.x;
   if (q < r0 + n && a[q] > 0.0) b[q] = 1;
}

void build_ext_plan(ExtPlan &p, const IdSpace &S, const std::vector<long long> &part)
{
   Comm &cm = Comm::world();
   p        = ExtPlan();
   p.send_counts.assign((size_t)cm.size, 0);
   p.recv_counts.assign((size_t)cm.size, 0);
   const int              no = S.nother();
   std::vector<int>       ridx((size_t)std::max(no, 1), 0);
   std::vector<long long> want((size_t)std::max(no, 1), 0);
   for (int q = 0; q < no; q++)
   { // ascending ids => grouped by ascending owner
      const long long g = S.other(q);
      const int       o = owner_of(g, part);
      HDA_REQUIRE(o >= 0 && o < cm.size && o != cm.rank, "extended node without a remote owner");
      p.recv_counts[(size_t)o]++;
      ridx[(size_t)q] = S.other_pos(q);
      want[(size_t)q] = g;
   }
   std::vector<long>      eight((size_t)cm.size, 8), sb((size_t)cm.size), rb((size_t)cm.size);
   std::vector<long long> w((size_t)cm.size), asked((size_t)cm.size);
   for (int q = 0; q < cm.size; q++) w[(size_t)q] = p.recv_counts[(size_t)q];
   cm.alltoallv_host(w.data(), eight.data(), asked.data(), eight.data());
   long tot = 0;
   for (int q = 0; q < cm.size; q++)
   {
      p.send_counts[(size_t)q] = (int)asked[(size_t)q];
      sb[(size_t)q]            = 8L * p.recv_counts[(size_t)q];
      rb[(size_t)q]            = 8L * p.send_counts[(size_t)q];
      tot += p.send_counts[(size_t)q];
   }
   std::vector<long long> req((size_t)std::max<long>(tot, 1));
   cm.alltoallv_host(want.data(), sb.data(), req.data(), rb.data());
   std::vector<int> sidx((size_t)std::max<long>(tot, 1), 0);
   for (long q = 0; q < tot; q++)
   {
      const long long l = req[(size_t)q] - S.lo;
      HDA_REQUIRE(l >= 0 && l < S.nown(), "peer requested a node this rank does not own");
      sidx[(size_t)q] = S.off() + (int)l;
   }
   p.nsend = (int)tot;
   p.nrecv = no;
   p.send_idx.upload(sidx.data(), sidx.size());
   p.recv_idx.upload(ridx.data(), ridx.size());
   p.sbuf.alloc((size_t)std::max(p.nsend, 1));
   p.rbuf.alloc((size_t)std::max(p.nrecv, 1));
}

void ext_exchange(ExtPlan &p, double *arr) // owners' values -> ghost copies
{
   Comm &cm = Comm::world();
   if (p.nsend) k_gather_d<<<ceil_div(p.nsend, 256), 256, 0, STREAM>>>(p.nsend, p.send_idx.data(), arr, p.sbuf.data());
   cm.exchange_dev(p.sbuf.data(), p.send_counts.data(), p.rbuf.data(), p.recv_counts.data());
   if (p.nrecv) k_scatter_d<<<ceil_div(p.nrecv, 256), 256, 0, STREAM>>>(p.nrecv, p.recv_idx.data(), p.rbuf.data(), arr);
}
void ext_reverse_add(ExtPlan &p, double *arr) // ghost copies' values added into the owners' entries
{
   Comm &cm = Comm::world();
   if (p.nrecv) k_gather_d<<<ceil_div(p.nrecv, 256), 256, 0, STREAM>>>(p.nrecv, p.recv_idx.data(), arr, p.rbuf.data());
   cm.exchange_dev(p.rbuf.data(), p.recv_counts.data(), p.sbuf.data(), p.send_counts.data());
   if (p.nsend) k_scatter_add_d<<<ceil_div(p.nsend, 256), 256, 0, STREAM>>>(p.nsend, p.send_idx.data(), p.sbuf.data(), arr);
}
void ext_exchange_int(ExtPlan &p, int n, int *arr, DArray<double> &tmp)
{
   k_i2d<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, arr, tmp.data());
   ext_exchange(p, tmp.data());
   k_d2i<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, tmp.data(), arr);
}

// ---- rows of the nodes peers asked for: lengths, then (global column, value) pairs
__global__ __launch_bounds__(256) void k_rows_len(int nr, const int *__restrict__ rows, const int *__restrict__ rp, double *__restrict__ len)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < nr) len[q] = (double)(rp[rows[q] + 1] - rp[rows[q]]);
}
__global__ __launch_bounds__(256) void k_rows_pack(int nr, const int *__restrict__ rows, const int *__restrict__ rp,
                                                   const int *__restrict__ ofs, const long long *__restrict__ gcol,
                                                   const double *__restrict__ val, double *__restrict__ out)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= nr) return;
   const int s = rp[rows[q]], e = rp[rows[q] + 1];
   double   *o = out + 2 * (size_t)ofs[q];
   for (int k = s; k < e; k++)
   {
      o[2 * (k - s)]     = (double)gcol[k]; // exact below 2^53
      o[2 * (k - s) + 1] = val[k];
   }
}

// host sparse rows: row q of the list has entries [rp[q], rp[q+1])
struct HostRows {
   std::vector<int>       row; // destination row of every listed row (filled by the caller)
   std::vector<int>       rp{0};
   std::vector<long long> gcol;
   std::vector<int>       col;
   std::vector<double>    val;
};

// the rows named by send_idx (grouped by destination) go to the peers; what arrives (in the
// order of this rank's ghost list) is returned with global columns
void fetch_rows(int nsend, const int *send_idx_dev, const std::vector<int> &send_counts, int nrecv,
                const std::vector<int> &recv_counts, const int *rp, const long long *gcol, const double *val, HostRows &out)
{
   Comm          &cm = Comm::world();
   DArray<double> slen((size_t)std::max(nsend, 1)), rlen((size_t)std::max(nrecv, 1));
   if (nsend) k_rows_len<<<ceil_div(nsend, 256), 256, 0, STREAM>>>(nsend, send_idx_dev, rp, slen.data());
   cm.exchange_dev(slen.data(), send_counts.data(), rlen.data(), recv_counts.data());
   std::vector<double> hs((size_t)std::max(nsend, 1)), hr((size_t)std::max(nrecv, 1));
   slen.download(hs.data(), hs.size());
   rlen.download(hr.data(), hr.size());
   std::vector<int> sofs((size_t)nsend + 1, 0), sc2((size_t)cm.size, 0), rc2((size_t)cm.size, 0);
   for (int q = 0; q < nsend; q++) sofs[(size_t)q + 1] = sofs[(size_t)q] + (int)hs[(size_t)q];
   out.rp.assign((size_t)nrecv + 1, 0);
   for (int q = 0; q < nrecv; q++) out.rp[(size_t)q + 1] = out.rp[(size_t)q] + (int)hr[(size_t)q];
   {
      int a = 0, b = 0;
      for (int p = 0; p < cm.size; p++)
      {
         sc2[(size_t)p] = 2 * (sofs[(size_t)(a + send_counts[(size_t)p])] - sofs[(size_t)a]);
         rc2[(size_t)p] = 2 * (out.rp[(size_t)(b + recv_counts[(size_t)p])] - out.rp[(size_t)b]);
         a += send_counts[(size_t)p];
         b += recv_counts[(size_t)p];
      }
   }
   const int      stot = sofs[(size_t)nsend], rtot = out.rp[(size_t)nrecv];
   DArray<int>    dofs;
   DArray<double> sdat((size_t)std::max(2 * stot, 1)), rdat((size_t)std::max(2 * rtot, 1));
   dofs.upload(sofs.data(), sofs.size());
   if (nsend) k_rows_pack<<<ceil_div(nsend, 256), 256, 0, STREAM>>>(nsend, send_idx_dev, rp, dofs.data(), gcol, val, sdat.data());
   cm.exchange_dev(sdat.data(), sc2.data(), rdat.data(), rc2.data());
   std::vector<double> hd((size_t)std::max(2 * rtot, 1));
   rdat.download(hd.data(), hd.size());
   out.gcol.resize((size_t)rtot);
   out.val.resize((size_t)rtot);
   for (int k = 0; k < rtot; k++)
   {
      out.gcol[(size_t)k] = (long long)hd[2 * (size_t)k];
      out.val[(size_t)k]  = hd[2 * (size_t)k + 1];
   }
}

__global__ __launch_bounds__(256) void k_loc2gcol(long nnz, const int *__restrict__ cj, int ncl, long long lo,
                                                  const long long *__restrict__ ghosts, long long *__restrict__ out)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256)
      out[k] = (cj[k] < ncl) ? lo + cj[k] : ghosts[cj[k] - ncl];
}
__global__ __launch_bounds__(256) void k_row_len(int n, const int *__restrict__ rp, int *__restrict__ len)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) len[i] = rp[i + 1] - rp[i];
}
__global__ __launch_bounds__(256) void k_add_len(int n, const int *__restrict__ row, const int *__restrict__ hrp, int *len)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) atomicAdd(&len[row[q]], hrp[q + 1] - hrp[q]);
}
__global__ __launch_bounds__(256) void k_copy_rows_mapped(int nrows, const int *__restrict__ srp, const int *__restrict__ scj,
                                                          const double *__restrict__ sv, const int *__restrict__ colmap,
                                                          const int *__restrict__ drp, int *__restrict__ dcj, double *__restrict__ dv)
{ // source row i -> destination offset drp[i] (the caller passes rowptr + base)
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= nrows) return;
   const int s = srp[i], e = srp[i + 1], d = drp[i];
   for (int k = s; k < e; k++)
   {
      dcj[d + (k - s)] = colmap ? colmap[scj[k]] : scj[k];
      dv[d + (k - s)]  = sv[k];
   }
}
__global__ __launch_bounds__(256) void k_place_rows(int n, const int *__restrict__ row, const int *__restrict__ hrp, const int *__restrict__ hc,
                                                    const double *__restrict__ hv, int base, int nsrc, const int *__restrict__ srp,
                                                    const int *__restrict__ drp, int *__restrict__ dcj, double *__restrict__ dv)
{ // listed row q lands after the device-side entries (if any) of its destination row
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   const int r = row[q];
   int       d = drp[r];
   if (r >= base && r < base + nsrc) d += srp[r - base + 1] - srp[r - base];
   for (int k = hrp[q]; k < hrp[q + 1]; k++, d++)
   {
      dcj[d] = hc[k];
      dv[d]  = hv[k];
   }
}

// CSR from (a) a device CSR whose rows land on rows base.. with mapped columns and (b) listed
// host rows (distinct destination rows; a destination may also receive device entries when
// the two column sets are disjoint).  Rows come out column-sorted.
void assemble_csr(int nrows, int ncols, int base, const DCsr &src, const int *colmap_dev, const HostRows &H, DCsr &out)
{
   const int   nh = (int)H.row.size();
   DArray<int> len((size_t)nrows + 1), drow, drp, dc;
   DArray<double> dv;
   len.zero();
   if (src.nrows) k_row_len<<<ceil_div(src.nrows, 256), 256, 0, STREAM>>>(src.nrows, src.rowptr.data(), len.data() + base);
   if (nh)
   {
      drow.upload(H.row.data(), (size_t)nh);
      drp.upload(H.rp.data(), (size_t)nh + 1);
      k_add_len<<<ceil_div(nh, 256), 256, 0, STREAM>>>(nh, drow.data(), drp.data(), len.data());
   }
   out.nrows = nrows;
   out.ncols = ncols;
   out.rowptr.alloc((size_t)nrows + 1);
   require_int32_total(nrows, len.data(), "assembled row block");
   exclusive_scan(nrows, len.data(), out.rowptr.data(), nullptr);
   HDA_HIP(hipMemcpyAsync(&out.nnz, out.rowptr.data() + nrows, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   out.col.alloc((size_t)std::max(out.nnz, 1));
   out.val.alloc((size_t)std::max(out.nnz, 1));
   if (src.nrows)
      k_copy_rows_mapped<<<ceil_div(src.nrows, 256), 256, 0, STREAM>>>(src.nrows, src.rowptr.data(), src.col.data(), src.val.data(), colmap_dev,
                                                                      out.rowptr.data() + base, out.col.data(), out.val.data());
   if (nh && !H.col.empty())
   {
      dc.upload(H.col.data(), H.col.size());
      dv.upload(H.val.data(), H.val.size());
      k_place_rows<<<ceil_div(nh, 256), 256, 0, STREAM>>>(nh, drow.data(), drp.data(), dc.data(), dv.data(), base, src.nrows, src.rowptr.data(),
                                                         out.rowptr.data(), out.col.data(), out.val.data());
   }
   out.chunk_row.release();
   out.nchunks = 0;
   out.maxrow  = -1;
   sort_rows(out);
   Context::get().sync();
}

__global__ __launch_bounds__(256) void k_shift(int n, const int *__restrict__ in, int by, int *__restrict__ out)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) out[i] = in[i] - by;
}

void slice_rows(const DCsr &M, int r0, int r1, DCsr &out)
{
   int k0 = 0, k1 = 0;
   HDA_HIP(hipMemcpyAsync(&k0, M.rowptr.data() + r0, 4, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&k1, M.rowptr.data() + r1, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   out.nrows = r1 - r0;
   out.ncols = M.ncols;
   out.nnz   = k1 - k0;
   out.rowptr.alloc((size_t)out.nrows + 1);
   k_shift<<<ceil_div(out.nrows + 1, 256), 256, 0, STREAM>>>(out.nrows + 1, M.rowptr.data() + r0, k0, out.rowptr.data());
   out.col.alloc((size_t)std::max(out.nnz, 1));
   out.val.alloc((size_t)std::max(out.nnz, 1));
   if (out.nnz)
   {
      HDA_HIP(hipMemcpyAsync(out.col.data(), M.col.data() + k0, sizeof(int) * (size_t)out.nnz, hipMemcpyDeviceToDevice, STREAM));
      HDA_HIP(hipMemcpyAsync(out.val.data(), M.val.data() + k0, sizeof(double) * (size_t)out.nnz, hipMemcpyDeviceToDevice, STREAM));
   }
}

// a (small) slice of rows as host rows
void rows_to_host(const DCsr &M, int r0, int r1, std::vector<int> &rp, std::vector<int> &cj, std::vector<double> &v)
{
   rp.assign((size_t)(r1 - r0) + 1, 0);
   if (r1 <= r0) { cj.clear(); v.clear(); return; }
   HDA_HIP(hipMemcpyAsync(rp.data(), M.rowptr.data() + r0, 4 * rp.size(), hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   const int k0 = rp.front(), k1 = rp.back();
   cj.resize((size_t)(k1 - k0));
   v.resize((size_t)(k1 - k0));
   if (k1 > k0)
   {
      HDA_HIP(hipMemcpyAsync(cj.data(), M.col.data() + k0, 4 * cj.size(), hipMemcpyDeviceToHost, STREAM));
      HDA_HIP(hipMemcpyAsync(v.data(), M.val.data() + k0, 8 * v.size(), hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
   }
   for (auto &x : rp) x -= k0;
}

// build the extended view of one level
void build_ext_level(const DCsr &Aloc, const std::vector<long long> &ghosts, const std::vector<long long> &part, const HaloPlan &hA,
                     ExtLevel &X)
{
   Comm           &cm = Comm::world();
   const long long lo = part[(size_t)cm.rank], hi = part[(size_t)cm.rank + 1];
   X.nloc             = (int)(hi - lo);
   HDA_REQUIRE(Aloc.nrows == X.nloc && Aloc.ncols == X.nloc + (int)ghosts.size(), "ext level: block shape mismatch");
   DArray<long long> dgh, gcol((size_t)std::max(Aloc.nnz, 1));
   {
      std::vector<long long> g = ghosts;
      if (g.empty()) g.push_back(0);
      dgh.upload(g.data(), g.size());
   }
   if (Aloc.nnz)
      k_loc2gcol<<<std::min(ceil_div(Aloc.nnz, 256), 1 << 16), 256, 0, STREAM>>>(Aloc.nnz, Aloc.col.data(), X.nloc, lo, dgh.data(), gcol.data());
   HostRows H;
   fetch_rows(hA.send_total, hA.send_idx.data(), hA.send_counts, (int)ghosts.size(), hA.recv_counts, Aloc.rowptr.data(), gcol.data(),
              Aloc.val.data(), H);
   // node list: ghosts, then columns of ghost rows that are neither owned nor ghosts
   std::vector<long long> other = ghosts;
   for (long long c : H.gcol)
      if (c < lo || c >= hi) other.push_back(c);
   X.S.set(lo, hi, other);
   X.off  = X.S.off();
   X.next = X.S.size();
   space_gids_to_device(X.S, X.gid_dev);
   // solve-layout column -> extended id
   DArray<int> cmap((size_t)std::max(Aloc.ncols, 1));
   if (X.nloc) k_iota_i<<<ceil_div(X.nloc, 256), 256, 0, STREAM>>>(X.nloc, X.off, cmap.data());
   if (!ghosts.empty())
   {
      std::vector<int> gm(ghosts.size());
      for (size_t g = 0; g < ghosts.size(); g++) gm[g] = X.S.of(ghosts[g]);
      HDA_HIP(hipMemcpyAsync(cmap.data() + X.nloc, gm.data(), 4 * gm.size(), hipMemcpyHostToDevice, STREAM));
      Context::get().sync();
   }
   H.row.resize(ghosts.size());
   H.col.resize(H.gcol.size());
   for (size_t g = 0; g < ghosts.size(); g++) H.row[g] = X.S.of(ghosts[g]);
   for (size_t k = 0; k < H.gcol.size(); k++) H.col[k] = X.S.of(H.gcol[k]);
   assemble_csr(X.next, X.next, X.off, Aloc, cmap.data(), H, X.E);
   build_ext_plan(X.plan, X.S, part);
}

// PMIS on the owned rows of E with ghost values exchanged every round
void pmis_dist(ExtLevel &X, const unsigned char *smask, const int *ns, uint64_t seed, int level, int *cf)
{
   Comm                 &cm = Comm::world();
   const int             n  = X.next, g = ceil_div(std::max(X.nloc, 1), 256), gn = ceil_div(std::max(n, 1), 256);
   DArray<int>           indeg((size_t)n), counter(1);
   DArray<double>        meas((size_t)n), tmp((size_t)n);
   DArray<unsigned char> notmax((size_t)n);
   indeg.zero();
   notmax.zero();
   meas.zero();
   counter.zero();
   HDA_HIP(hipMemsetAsync(cf, 0, sizeof(int) * (size_t)n, STREAM));
   int e0 = 0, e1 = 0;
   HDA_HIP(hipMemcpyAsync(&e0, X.E.rowptr.data() + X.off, 4, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&e1, X.E.rowptr.data() + X.off + X.nloc, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   if (e1 > e0) k_indeg<<<std::min(ceil_div(e1 - e0, 256), 1 << 16), 256, 0, STREAM>>>(e1 - e0, X.E.col.data() + e0, smask + e0, indeg.data());
   // in-degree contributions that landed on ghost copies go to the owners
   k_i2d<<<gn, 256, 0, STREAM>>>(n, indeg.data(), tmp.data());
   ext_reverse_add(X.plan, tmp.data());
   k_d2i<<<gn, 256, 0, STREAM>>>(n, tmp.data(), indeg.data());
   if (X.nloc) k_pmis_init<<<std::min(g, kPmisGrid), 256, 0, STREAM>>>(X.off, X.nloc, ns, indeg.data(), seed, level, 0, X.gid_dev.data(), meas.data(), cf, counter.data());
   ext_exchange(X.plan, meas.data());
   ext_exchange_int(X.plan, n, cf, tmp);
   long long left = 0;
   {
      int c = 0;
      counter.download(&c, 1);
      left = c;
      cm.allreduce_host(&left, 1, 0);
   }
   int rounds = 0;
   while (left > 0)
   {
      if (X.nloc) k_pmis_mark<<<g, 256, 0, STREAM>>>(X.nloc, X.E.rowptr.data(), X.E.col.data(), smask, cf, meas.data(), notmax.data(), X.off);
      k_uc2d<<<gn, 256, 0, STREAM>>>(n, notmax.data(), tmp.data());
      ext_reverse_add(X.plan, tmp.data());
      if (X.nloc) k_d2uc_or<<<g, 256, 0, STREAM>>>(X.off, X.nloc, tmp.data(), notmax.data());
      if (X.nloc) k_pmis_setC<<<g, 256, 0, STREAM>>>(X.nloc, cf, notmax.data(), X.off);
      notmax.zero();
      ext_exchange_int(X.plan, n, cf, tmp);
      counter.zero();
      if (X.nloc) k_pmis_setF<<<std::min(g, kPmisGrid), 256, 0, STREAM>>>(X.nloc, X.E.rowptr.data(), X.E.col.data(), smask, cf, counter.data(), X.off);
      ext_exchange_int(X.plan, n, cf, tmp);
      int c = 0;
      counter.download(&c, 1);
      left = c;
      cm.allreduce_host(&left, 1, 0);
      HDA_REQUIRE(++rounds < 10000, "distributed PMIS did not terminate");
   }
}

__global__ __launch_bounds__(256) void k_cg_owned(int nloc, int off, const int *__restrict__ cf, const int *__restrict__ cidx, long long coff,
                                                  double *__restrict__ cg)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < nloc) cg[off + i] = (cf[off + i] == 1) ? (double)(coff + cidx[i]) : -1.0;
}
__global__ __launch_bounds__(256) void k_cmark_ext(int n, const int *__restrict__ cf, int *__restrict__ m)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) m[i] = (cf[i] == 1);
}
__global__ __launch_bounds__(256) void k_cgc(int n, const int *__restrict__ cf, const int *__restrict__ cidxE, const double *__restrict__ cg,
                                             long long *__restrict__ cgc)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n && cf[i] == 1) cgc[cidxE[i]] = (long long)cg[i];
}
__global__ __launch_bounds__(256) void k_map_cols(long nnz, const int *__restrict__ in, const int *__restrict__ map, int *__restrict__ out)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256) out[k] = map[in[k]];
}
__global__ __launch_bounds__(256) void k_cols_to_gid(long nnz, const int *__restrict__ in, const long long *__restrict__ gid, long long *__restrict__ out)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256) out[k] = gid[in[k]];
}
__global__ __launch_bounds__(256) void k_flag_cols(long nnz, const int *__restrict__ cj, int *flags)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256) flags[cj[k]] = 1;
}

// solve-layout block of a matrix whose columns live in the id space C: owned columns ->
// [0, nown), the foreign columns that are referenced -> nown + rank in ascending id (list returned)
void localize_cols(const DCsr &M, const IdSpace &C, DCsr &L, std::vector<long long> &ghosts)
{
   const int   nc = C.size(), no = C.nother();
   DArray<int> flags((size_t)std::max(nc, 1)), map((size_t)std::max(nc, 1));
   flags.zero();
   if (M.nnz) k_flag_cols<<<std::min(ceil_div(M.nnz, 256), 1 << 16), 256, 0, STREAM>>>(M.nnz, M.col.data(), flags.data());
   std::vector<int> fb(C.below.size()), fa(C.above.size());
   if (!fb.empty()) HDA_HIP(hipMemcpyAsync(fb.data(), flags.data(), 4 * fb.size(), hipMemcpyDeviceToHost, STREAM));
   if (!fa.empty()) HDA_HIP(hipMemcpyAsync(fa.data(), flags.data() + C.off() + C.nown(), 4 * fa.size(), hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   ghosts.clear();
   for (int q = 0; q < no; q++)
   {
      int &f = q < C.off() ? fb[(size_t)q] : fa[(size_t)(q - C.off())];
      if (f) { f = C.nown() + (int)ghosts.size(); ghosts.push_back(C.other(q)); }
   }
   if (!fb.empty()) HDA_HIP(hipMemcpyAsync(map.data(), fb.data(), 4 * fb.size(), hipMemcpyHostToDevice, STREAM));
   if (C.nown()) k_iota_i<<<ceil_div(C.nown(), 256), 256, 0, STREAM>>>(C.nown(), 0, map.data() + C.off());
   if (!fa.empty()) HDA_HIP(hipMemcpyAsync(map.data() + C.off() + C.nown(), fa.data(), 4 * fa.size(), hipMemcpyHostToDevice, STREAM));
   L.nrows = M.nrows;
   L.ncols = C.nown() + (int)ghosts.size();
   L.nnz   = M.nnz;
   L.rowptr.copy_from(M.rowptr);
   L.col.alloc((size_t)std::max(M.nnz, 1));
   L.val.copy_from(M.val);
   if (M.nnz) k_map_cols<<<std::min(ceil_div(M.nnz, 256), 1 << 16), 256, 0, STREAM>>>(M.nnz, M.col.data(), map.data(), L.col.data());
   L.chunk_row.release();
   L.nchunks = 0;
   L.maxrow  = -1;
   sort_rows(L);
   Context::get().sync();
}

// entries of M whose column is outside [q0, q1): (row, col, val) triplets on the host
__global__ __launch_bounds__(256) void k_count_outside(int n, const int *__restrict__ rp, const int *__restrict__ cj, int q0, int q1, int *__restrict__ cnt)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int c = 0;
   for (int k = rp[i]; k < rp[i + 1]; k++) c += (cj[k] < q0 || cj[k] >= q1);
   cnt[i] = c;
}
__global__ __launch_bounds__(256) void k_write_outside(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                       int q0, int q1, const int *__restrict__ pos, int *__restrict__ oi, int *__restrict__ oc,
                                                       double *__restrict__ ov)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int d = pos[i];
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (cj[k] < q0 || cj[k] >= q1)
      {
         oi[d] = i;
         oc[d] = cj[k];
         ov[d] = v[k];
         d++;
      }
}
void entries_outside(const DCsr &M, int q0, int q1, std::vector<int> &oi, std::vector<int> &oc, std::vector<double> &ov)
{
   const int   n = M.nrows;
   DArray<int> cnt((size_t)n + 1), pos((size_t)n + 1);
   cnt.zero();
   if (n) k_count_outside<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, M.rowptr.data(), M.col.data(), q0, q1, cnt.data());
   exclusive_scan(n, cnt.data(), pos.data(), nullptr);
   int tot = 0;
   HDA_HIP(hipMemcpyAsync(&tot, pos.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   oi.resize((size_t)tot);
   oc.resize((size_t)tot);
   ov.resize((size_t)tot);
   if (!tot) return;
   DArray<int>    di((size_t)tot), dc((size_t)tot);
   DArray<double> dv((size_t)tot);
   k_write_outside<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, M.rowptr.data(), M.col.data(), M.val.data(), q0, q1, pos.data(), di.data(), dc.data(), dv.data());
   di.download(oi.data(), (size_t)tot);
   dc.download(oc.data(), (size_t)tot);
   dv.download(ov.data(), (size_t)tot);
}

// C = A + B for column-sorted rows, B given as a (mostly empty) device CSR over the same rows;
// a row's sum is formed in ascending column order, A's entry before B's
__global__ __launch_bounds__(256) void k_merge_count(int n, const int *__restrict__ arp, const int *__restrict__ acj, const int *__restrict__ brp,
                                                     const int *__restrict__ bcj, int *__restrict__ cnt)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int a = arp[i], ae = arp[i + 1], b = brp[i], be = brp[i + 1], c = 0;
   if (b == be) { cnt[i] = ae - a; return; }
   while (a < ae && b < be)
   {
      const int ja = acj[a], jb = bcj[b];
      a += (ja <= jb);
      b += (jb <= ja);
      c++;
   }
   cnt[i] = c + (ae - a) + (be - b);
}
__global__ __launch_bounds__(256) void k_merge_write(int n, const int *__restrict__ arp, const int *__restrict__ acj, const double *__restrict__ av,
                                                     const int *__restrict__ brp, const int *__restrict__ bcj, const double *__restrict__ bv,
                                                     const int *__restrict__ crp, int *__restrict__ ccj, double *__restrict__ cv)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int a = arp[i], ae = arp[i + 1], b = brp[i], be = brp[i + 1], d = crp[i];
   while (a < ae || b < be)
   {
      const int ja = (a < ae) ? acj[a] : 0x7fffffff, jb = (b < be) ? bcj[b] : 0x7fffffff;
      if (ja < jb) { ccj[d] = ja; cv[d] = av[a++]; }
      else if (jb < ja) { ccj[d] = jb; cv[d] = bv[b++]; }
      else { ccj[d] = ja; cv[d] = av[a++] + bv[b++]; }
      d++;
   }
}
void add_rows(const DCsr &A, const DCsr &B, DCsr &C)
{
   HDA_REQUIRE(A.nrows == B.nrows && A.ncols == B.ncols, "add_rows: shapes differ");
   const int   n = A.nrows;
   DArray<int> cnt((size_t)n + 1);
   cnt.zero();
   C.nrows = n;
   C.ncols = A.ncols;
   C.rowptr.alloc((size_t)n + 1);
   if (n) k_merge_count<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), B.rowptr.data(), B.col.data(), cnt.data());
   require_int32_total(n, cnt.data(), "merged row block");
   exclusive_scan(n, cnt.data(), C.rowptr.data(), nullptr);
   HDA_HIP(hipMemcpyAsync(&C.nnz, C.rowptr.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   C.col.alloc((size_t)std::max(C.nnz, 1));
   C.val.alloc((size_t)std::max(C.nnz, 1));
   if (n)
      k_merge_write<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), B.rowptr.data(), B.col.data(), B.val.data(),
                                                         C.rowptr.data(), C.col.data(), C.val.data());
}

// (destination id, second id, value) records to the owners of the destination ids
struct Rec {
   long long a, b;
   double    w;
};
void route_records(std::vector<std::vector<Rec>> &out, std::vector<Rec> &in)
{
   Comm                  &cm = Comm::world();
   std::vector<long>      eight((size_t)cm.size, 8), sb((size_t)cm.size), rb((size_t)cm.size);
   std::vector<long long> sc((size_t)cm.size), rc((size_t)cm.size);
   std::vector<Rec>       flat;
   for (int p = 0; p < cm.size; p++)
   {
      sc[(size_t)p] = (long long)out[(size_t)p].size();
      flat.insert(flat.end(), out[(size_t)p].begin(), out[(size_t)p].end());
   }
   cm.alltoallv_host(sc.data(), eight.data(), rc.data(), eight.data());
   long tot = 0;
   for (int p = 0; p < cm.size; p++)
   {
      sb[(size_t)p] = (long)(sc[(size_t)p] * (long long)sizeof(Rec));
      rb[(size_t)p] = (long)(rc[(size_t)p] * (long long)sizeof(Rec));
      tot += (long)rc[(size_t)p];
   }
   in.resize((size_t)std::max<long>(tot, 1));
   if (flat.empty()) flat.resize(1);
   cm.alltoallv_host(flat.data(), sb.data(), in.data(), rb.data());
   in.resize((size_t)tot);
}

} // namespace

// Row-partitioned product C = X * Y.  X: this rank's rows, columns [owned | ghosts] of a space whose owned rows
// of Y this rank holds (hX = the halo plan of that space); Y: this rank's rows with columns [owned | y_ghosts]
// of a second space with row starts part_c.  The rows of Y that X's ghost columns name are fetched from their
// owners; the result comes back in solve layout, columns [owned | c_ghosts] (ascending global id).
void dist_spgemm(const DCsr &X, const HaloPlan &hX, const DCsr &Y, const std::vector<long long> &y_ghosts,
                 const std::vector<long long> &part_c, DCsr &C, std::vector<long long> &c_ghosts)
{
   Comm           &cm = Comm::world();
   const long long lo = part_c[(size_t)cm.rank], hi = part_c[(size_t)cm.rank + 1];
   const int       nown_c = (int)(hi - lo), nb = Y.nrows, ng = X.ncols - nb;
   HDA_REQUIRE(ng >= 0 && ng == hX.nghost, "dist_spgemm: X's ghost columns do not match the halo plan");
   HDA_REQUIRE(Y.ncols == nown_c + (int)y_ghosts.size(), "dist_spgemm: Y's ghost list does not match its columns");
   DArray<long long> dgh, gcol((size_t)std::max(Y.nnz, 1));
   {
      std::vector<long long> g = y_ghosts;
      if (g.empty()) g.push_back(0);
      dgh.upload(g.data(), g.size());
   }
   if (Y.nnz) k_loc2gcol<<<std::min(ceil_div(Y.nnz, 256), 1 << 16), 256, 0, STREAM>>>(Y.nnz, Y.col.data(), nown_c, lo, dgh.data(), gcol.data());
   HostRows H;
   fetch_rows(hX.send_total, hX.send_idx.data(), hX.send_counts, ng, hX.recv_counts, Y.rowptr.data(), gcol.data(), Y.val.data(), H);
   // column space of the extended Y: its own ghosts plus whatever the fetched rows reference
   std::vector<long long> other = y_ghosts;
   for (long long c : H.gcol)
      if (c < lo || c >= hi) other.push_back(c);
   IdSpace S;
   S.set(lo, hi, other);
   DArray<int> cmap((size_t)std::max(Y.ncols, 1));
   if (nown_c) k_iota_i<<<ceil_div(nown_c, 256), 256, 0, STREAM>>>(nown_c, S.off(), cmap.data());
   if (!y_ghosts.empty())
   {
      std::vector<int> gm(y_ghosts.size());
      for (size_t g = 0; g < y_ghosts.size(); g++) gm[g] = S.of(y_ghosts[g]);
      HDA_HIP(hipMemcpyAsync(cmap.data() + nown_c, gm.data(), 4 * gm.size(), hipMemcpyHostToDevice, STREAM));
      Context::get().sync();
   }
   H.row.resize((size_t)ng);
   H.col.resize(H.gcol.size());
   for (int g = 0; g < ng; g++) H.row[(size_t)g] = nb + g; // X's ghost column g = row nb + g of the extended Y
   for (size_t k = 0; k < H.gcol.size(); k++) H.col[k] = S.of(H.gcol[k]);
   DCsr Yext, T;
   assemble_csr(nb + ng, S.size(), 0, Y, cmap.data(), H, Yext);
   spgemm(X, Yext, T);
   localize_cols(T, S, C, c_ghosts);
}

__global__ __launch_bounds__(256) void k_scatter_add(int n, const int *__restrict__ idx, const double *__restrict__ src, double *dst)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) dst[idx[q]] += src[q];
}
// the ghost copies' values are added into their owners' entries (peer by peer, ascending rank: a fixed order)
void halo_reverse_add(const HaloPlan &h, double *x_ext)
{
   Comm &cm = Comm::world();
   if (cm.size == 1 || h.send_counts.empty()) return;
   DArray<double> buf((size_t)std::max(h.send_total, 1));
   cm.exchange_dev(x_ext + h.nloc, h.recv_counts.data(), buf.data(), h.send_counts.data());
   int at = 0;
   for (int p = 0; p < cm.size; p++)
   { // indices inside one peer's list are distinct; different peers may name the same row
      const int c = h.send_counts[(size_t)p];
      if (c) k_scatter_add<<<ceil_div(c, 256), 256, 0, STREAM>>>(c, h.send_idx.data() + at, buf.data() + at, x_ext);
      at += c;
   }
}

__global__ __launch_bounds__(256) void k_remap_ghost_cols(long nnz, int nown, const int *__restrict__ map, int *__restrict__ cj)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256)
      if (cj[k] >= nown) cj[k] = map[cj[k] - nown];
}

void Amg::setup_dist_partitioned(const DCsr &Aloc, const HaloPlan &hA0_, const std::vector<long long> &part0,
                                 const std::vector<long long> &ghost_gids0)
{
   Comm &cm = Comm::world();
   HDA_REQUIRE(prm.coarsen_type == 8 && prm.interp_type == 6 && prm.num_functions <= 1,
               "partitioned setup: scalar PMIS + extended+i only");
   HDA_REQUIRE(prm.agg_num_levels <= 0, "partitioned setup: aggressive coarsening needs the replicated setup (the second strength graph reaches two ghost layers deep)");
   const long long rep_rows = replicate_rows(cm.size);
   const bool verbose = getenv("HDA_VERBOSE") != nullptr;
   // HDA_GHOST_PROLONG=0: the prolongation leaves ghost copies alone and the post-smoothing sweep refreshes them (4 exchanges per
   // level and cycle instead of 3).  Off under HDA_DIST_CHECK, which compares P entry by entry with the replicated setup's.
   const bool ghost_prolong = !(getenv("HDA_GHOST_PROLONG") && atoi(getenv("HDA_GHOST_PROLONG")) == 0) &&
                              !(getenv("HDA_DIST_CHECK") && *getenv("HDA_DIST_CHECK") && *getenv("HDA_DIST_CHECK") != '0');
   auto tick = [&]() {
      if (verbose) Context::get().sync();
      return std::chrono::steady_clock::now();
   };
   auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
   };
   dist = true;
   A0   = &Aloc;
   a0_dims[0] = Aloc.nrows; a0_dims[1] = Aloc.ncols; a0_dims[2] = Aloc.nnz;
   hA0  = &hA0_;
   levels.clear();
   levels.reserve((size_t)std::max(prm.max_levels, 1) + 1);
   levels.emplace_back();
   const int              maxl = std::max(prm.max_levels, 1);
   std::vector<long long> part = part0, ghosts = ghost_gids0;
   std::vector<size_t>    tail_len(1, ghost_gids0.size());
   stats_levels = 0;
   int l = 0;
   for (;; l++)
   {
      const DCsr     &Al    = level_A(l);
      const HaloPlan &hl    = level_hA(l);
      const long long nglob = part.back();
      {
         long long t[1] = {Al.nnz};
         cm.allreduce_host(t, 1, 0);
         if (stats_levels < 32) { stats_nnz[stats_levels] = (double)t[0]; stats_rows[stats_levels] = (double)nglob; stats_levels++; }
      }
      const bool stop = (nglob <= prm.max_coarse_size) || (l >= maxl - 1) || (l >= 1 && nglob <= rep_rows);
      if (stop) break;
      auto t0 = tick();
      ExtLevel X;
      build_ext_level(Al, ghosts, part, hl, X);
      const int       n = X.next;
      const long long lo = X.S.lo;
      auto t1 = tick();
      DArray<unsigned char> sm((size_t)std::max(X.E.nnz, 1));
      DArray<int>           ns((size_t)n + 1), cf((size_t)n + 1);
      strength_ns(X.E, prm.strong_th, prm.max_row_sum, sm.data(), ns.data());
      pmis_dist(X, sm.data(), ns.data(), prm.seed, l + level0, cf.data());
      auto t2 = tick();
      // global coarse numbering: rank blocks in rank order, C points in row order
      DArray<int> cm_own((size_t)X.nloc + 1), cidx((size_t)X.nloc + 1);
      k_cmark_ext<<<ceil_div(std::max(X.nloc, 1), 256), 256, 0, STREAM>>>(X.nloc, cf.data() + X.off, cm_own.data());
      exclusive_scan(X.nloc, cm_own.data(), cidx.data(), nullptr);
      int ncl = 0;
      HDA_HIP(hipMemcpyAsync(&ncl, cidx.data() + X.nloc, 4, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
      std::vector<long long> cnts, partc((size_t)cm.size + 1, 0);
      cm.allgather_ll(ncl, cnts);
      for (int r = 0; r < cm.size; r++) partc[(size_t)r + 1] = partc[(size_t)r] + cnts[(size_t)r];
      const long long ncglob = partc.back(), clo = partc[(size_t)cm.rank], chi = partc[(size_t)cm.rank + 1];
      if (ncglob == 0 || ncglob == nglob || ncglob < prm.min_coarse_size) break;
      DArray<double> cg((size_t)n);
      fill(n, -1.0, cg.data());
      if (X.nloc) k_cg_owned<<<ceil_div(X.nloc, 256), 256, 0, STREAM>>>(X.nloc, X.off, cf.data(), cidx.data(), clo, cg.data());
      ext_exchange(X.plan, cg.data());
      // interpolation on E (the rows of ghost nodes are by-products and dropped)
      DCsr PE;
      amg_interp_extpi(X.E, sm.data(), cf.data(), prm.pmax, prm.trunc_factor, PE);
      sm.release();
      auto t3 = tick();
      // coarse id space seen from here: compact column q of PE <-> global coarse id
      DArray<int> cmE((size_t)n + 1), cidxE((size_t)n + 1);
      k_cmark_ext<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, cf.data(), cmE.data());
      exclusive_scan(n, cmE.data(), cidxE.data(), nullptr);
      int q3[3] = {0, 0, 0};
      HDA_HIP(hipMemcpyAsync(&q3[0], cidxE.data() + X.off, 4, hipMemcpyDeviceToHost, STREAM));
      HDA_HIP(hipMemcpyAsync(&q3[1], cidxE.data() + X.off + X.nloc, 4, hipMemcpyDeviceToHost, STREAM));
      HDA_HIP(hipMemcpyAsync(&q3[2], cidxE.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
      const int qlo = q3[0], qhi = q3[1], nce = q3[2];
      HDA_REQUIRE(nce == PE.ncols && qhi - qlo == ncl, "partitioned setup: coarse numbering of the extended block is inconsistent");
      DArray<long long> cgc((size_t)std::max(nce, 1));
      k_cgc<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, cf.data(), cidxE.data(), cg.data(), cgc.data());
      IdSpace C1;
      {
         std::vector<long long> oth((size_t)(nce - ncl));
         if (qlo) HDA_HIP(hipMemcpyAsync(oth.data(), cgc.data(), 8 * (size_t)qlo, hipMemcpyDeviceToHost, STREAM));
         if (nce > qhi) HDA_HIP(hipMemcpyAsync(oth.data() + qlo, cgc.data() + qhi, 8 * (size_t)(nce - qhi), hipMemcpyDeviceToHost, STREAM));
         Context::get().sync();
         C1.set(clo, chi, oth);
         HDA_REQUIRE(C1.off() == qlo && C1.size() == nce, "partitioned setup: coarse ids of ghost C points are not monotone");
      }
      DCsr Pown;
      slice_rows(PE, X.off, X.off + X.nloc, Pown);
      PE = DCsr();
      // ---- P in solve layout + halo plan for coarse vectors
      AmgLevel              &lv = levels[(size_t)l];
      std::vector<long long> ghostc;
      localize_cols(Pown, C1, lv.P, ghostc);
      // (the halo plan of coarse vectors, lv.hP, is made below once the P rows of the ghost fine points are known: it may be widened)
      // ---- R: rows of P^T for the owned coarse points = local transpose + entries of fine rows owned elsewhere
      std::vector<Rec> recv_t;
      {
         std::vector<int>    oi, oc;
         std::vector<double> ov;
         entries_outside(Pown, qlo, qhi, oi, oc, ov);
         std::vector<std::vector<Rec>> out((size_t)cm.size);
         for (size_t k = 0; k < oi.size(); k++)
         {
            const long long g = C1.other(oc[k] < qlo ? oc[k] : oc[k] - ncl);
            out[(size_t)owner_of(g, partc)].push_back({g, lo + oi[k], ov[k]});
         }
         route_records(out, recv_t);
      }
      {
         DCsr PT, Rown;
         transpose(lv.P, PT);
         slice_rows(PT, 0, ncl, Rown);
         std::sort(recv_t.begin(), recv_t.end(), [](const Rec &x, const Rec &y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
         std::vector<long long> gf;
         for (auto &t : recv_t) gf.push_back(t.b);
         std::sort(gf.begin(), gf.end());
         gf.erase(std::unique(gf.begin(), gf.end()), gf.end());
         HostRows H;
         for (size_t q = 0; q < recv_t.size(); q++)
         {
            const int row = (int)(recv_t[q].a - clo);
            if (H.row.empty() || H.row.back() != row) { H.row.push_back(row); H.rp.push_back(H.rp.back()); }
            H.rp.back()++;
            H.col.push_back(X.nloc + (int)(std::lower_bound(gf.begin(), gf.end(), recv_t[q].b) - gf.begin()));
            H.val.push_back(recv_t[q].w);
         }
         assemble_csr(ncl, X.nloc + (int)gf.size(), 0, Rown, nullptr, H, lv.R);
         lv.hR = make_halo_plan(X.nloc, part, gf);
         tail_len[(size_t)l] = std::max(tail_len[(size_t)l], gf.size());
      }
      auto t4 = tick();
      // ---- Galerkin product
      DCsr                   Ac_loc;
      std::vector<long long> ghosts_next;
      {
         // P rows of my ghost nodes, with global coarse columns
         DArray<long long> gcolP((size_t)std::max(Pown.nnz, 1));
         if (Pown.nnz) k_cols_to_gid<<<std::min(ceil_div(Pown.nnz, 256), 1 << 16), 256, 0, STREAM>>>(Pown.nnz, Pown.col.data(), cgc.data(), gcolP.data());
         HostRows H;
         fetch_rows(hl.send_total, hl.send_idx.data(), hl.send_counts, (int)ghosts.size(), hl.recv_counts, Pown.rowptr.data(), gcolP.data(),
                    Pown.val.data(), H);
         gcolP.release();
         // ---- ghost rows of P (C1 of SURVEY 2.4, one exchange less per level and cycle): with the P rows of this rank's ghost
         // fine points at hand, the prolongation x += P e can update the ghost copies of x as well (their old values are the ones
         // the residual's exchange brought in, and x has not changed since), so the first post-smoothing sweep finds fresh ghosts
         // and needs no exchange of its own.  Price: the coarse halo grows by the coarse points only those rows name.
         if (ghost_prolong)
         {
            std::vector<long long> wide = ghostc;
            for (long long c : H.gcol)
               if (c < clo || c >= chi) wide.push_back(c);
            std::sort(wide.begin(), wide.end());
            wide.erase(std::unique(wide.begin(), wide.end()), wide.end());
            if (wide.size() != ghostc.size())
            { // P's ghost columns move to their slots in the wider list
               std::vector<int> mp(std::max<size_t>(ghostc.size(), 1), 0);
               for (size_t k = 0; k < ghostc.size(); k++) mp[k] = ncl + (int)(std::lower_bound(wide.begin(), wide.end(), ghostc[k]) - wide.begin());
               DArray<int> dmp;
               dmp.upload(mp.data(), mp.size());
               if (lv.P.nnz) k_remap_ghost_cols<<<std::min(ceil_div(lv.P.nnz, 256), 1 << 16), 256, 0, STREAM>>>(lv.P.nnz, ncl, dmp.data(), lv.P.col.data());
               lv.P.ncols = ncl + (int)wide.size();
               Context::get().sync();
            }
            HostRows G;
            G.rp = H.rp;
            G.row.resize(ghosts.size());
            for (size_t g = 0; g < ghosts.size(); g++) G.row[g] = (int)g;
            G.col.resize(H.gcol.size());
            for (size_t k = 0; k < H.gcol.size(); k++)
            {
               const long long c = H.gcol[k];
               G.col[k] = (c >= clo && c < chi) ? (int)(c - clo) : ncl + (int)(std::lower_bound(wide.begin(), wide.end(), c) - wide.begin());
            }
            G.val = H.val;
            assemble_csr((int)ghosts.size(), ncl + (int)wide.size(), 0, DCsr(), nullptr, G, lv.Pg);
            lv.pg_ready = true;
            ghostc      = std::move(wide);
         }
         lv.hP = make_halo_plan(ncl, partc, ghostc);
         // coarse space C2 = C1 U columns of the fetched rows
         IdSpace C2;
         {
            std::vector<long long> oth;
            oth.insert(oth.end(), C1.below.begin(), C1.below.end());
            oth.insert(oth.end(), C1.above.begin(), C1.above.end());
            for (long long c : H.gcol)
               if (c < clo || c >= chi) oth.push_back(c);
            C2.set(clo, chi, oth);
         }
         DArray<int> map12;
         space_map(C1, C2, map12);
         H.row.resize(ghosts.size());
         H.col.resize(H.gcol.size());
         for (size_t g = 0; g < ghosts.size(); g++) H.row[g] = X.S.of(ghosts[g]);
         for (size_t k = 0; k < H.gcol.size(); k++) H.col[k] = C2.of(H.gcol[k]);
         // Pe: P rows of every extended node (owned: computed here; layer-1 ghosts: fetched; layer 2: not needed)
         DCsr Pe, Eown, AP, PeOwn, PT, T;
         assemble_csr(n, C2.size(), X.off, Pown, map12.data(), H, Pe);
         slice_rows(X.E, X.off, X.off + X.nloc, Eown);
         X.E = DCsr();
         spgemm(Eown, Pe, AP);
         Eown = DCsr();
         slice_rows(Pe, X.off, X.off + X.nloc, PeOwn);
         Pe = DCsr();
         transpose(PeOwn, PT);
         PeOwn = DCsr();
         spgemm(PT, AP, T);
         PT = DCsr();
         AP = DCsr();
         // rows of coarse points owned elsewhere go to their owners
         const int                     q2lo = C2.off(), q2hi = q2lo + ncl;
         std::vector<std::vector<Rec>> eout((size_t)cm.size);
         for (int side = 0; side < 2; side++)
         {
            const int           r0 = side ? q2hi : 0, r1 = side ? C2.size() : q2lo;
            std::vector<int>    trp, tcj;
            std::vector<double> tv;
            rows_to_host(T, r0, r1, trp, tcj, tv);
            for (int q = r0; q < r1; q++)
            {
               const long long g = C2.other(q < q2lo ? q : q - ncl);
               const int       o = owner_of(g, partc);
               for (int k = trp[(size_t)(q - r0)]; k < trp[(size_t)(q - r0) + 1]; k++)
               {
                  const int       c  = tcj[(size_t)k];
                  const long long gc = (c >= q2lo && c < q2hi) ? clo + (c - q2lo) : C2.other(c < q2lo ? c : c - ncl);
                  eout[(size_t)o].push_back({g, gc, tv[(size_t)k]});
               }
            }
         }
         std::vector<Rec> erecv;
         route_records(eout, erecv);
         // C3 = C2 U received columns; A_c(owned rows) = T(owned rows) + received rows
         IdSpace C3;
         {
            std::vector<long long> oth;
            oth.insert(oth.end(), C2.below.begin(), C2.below.end());
            oth.insert(oth.end(), C2.above.begin(), C2.above.end());
            for (auto &e : erecv)
               if (e.b < clo || e.b >= chi) oth.push_back(e.b);
            C3.set(clo, chi, oth);
         }
         DArray<int> map23;
         space_map(C2, C3, map23);
         DCsr Town, Town3, B, Ac;
         slice_rows(T, q2lo, q2hi, Town);
         T = DCsr();
         assemble_csr(ncl, C3.size(), 0, Town, map23.data(), HostRows(), Town3);
         Town = DCsr();
         // received rows: sorted by (row, column, sender order), duplicates pre-summed in that order
         std::stable_sort(erecv.begin(), erecv.end(), [](const Rec &x, const Rec &y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
         HostRows HB;
         for (size_t q = 0; q < erecv.size(); q++)
         {
            const int row = (int)(erecv[q].a - clo), c = C3.of(erecv[q].b);
            if (HB.row.empty() || HB.row.back() != row) { HB.row.push_back(row); HB.rp.push_back(HB.rp.back()); }
            if (HB.rp.back() > HB.rp[HB.rp.size() - 2] && HB.col.back() == c) { HB.val.back() += erecv[q].w; continue; }
            HB.rp.back()++;
            HB.col.push_back(c);
            HB.val.push_back(erecv[q].w);
         }
         assemble_csr(ncl, C3.size(), 0, DCsr(), nullptr, HB, B);
         add_rows(Town3, B, Ac);
         Town3 = DCsr();
         localize_cols(Ac, C3, Ac_loc, ghosts_next);
      }
      auto t5 = tick();
      if (verbose)
         fprintf(stderr, "[hda] partitioned setup rank %d level %d: n=%d (+%d ext) nnz=%d -> nc=%d nnzAc=%d | ext %.2f strength+pmis %.2f interp %.2f P/R %.2f rap %.2f ms\n",
                 cm.rank, l, X.nloc, n - X.nloc, Al.nnz, ncl, Ac_loc.nnz, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5));
      levels.emplace_back();
      levels[(size_t)l + 1].A  = std::move(Ac_loc);
      levels[(size_t)l + 1].hA = make_halo_plan(ncl, partc, ghosts_next);
      tail_len[(size_t)l]      = std::max(tail_len[(size_t)l], ghosts.size());
      tail_len.push_back(std::max(ghostc.size(), ghosts_next.size()));
      part   = partc;
      ghosts = ghosts_next;
   }
   // ---- the remaining (small) operator is gathered and finished redundantly on every rank
   {
      const DCsr &Al = level_A(l);
      tail           = std::make_unique<Amg>(prm);
      tail->level0   = l + level0;
      gather_global(Al, part, ghosts, tail->own_A0);
      tail->setup(tail->own_A0);
      for (int t = 1; t < tail->stats_levels && stats_levels < 32; t++, stats_levels++)
      {
         stats_nnz[stats_levels]  = tail->stats_nnz[t];
         stats_rows[stats_levels] = tail->stats_rows[t];
      }
      coarse_n     = (int)part.back();
      coarse_dense = true; // coarse_solve() hands over to the tail
      coarse_lo    = part[(size_t)cm.rank];
      coarse_nloc  = (int)(part[(size_t)cm.rank + 1] - coarse_lo);
      cbuf_f.alloc((size_t)std::max(coarse_n, 1));
      cbuf_u.alloc(std::max<size_t>((size_t)std::max(coarse_n, 1), tail->vec_len0()));
      if (l > 0) levels[(size_t)l].A = DCsr(); // the hand-over level keeps only vectors
   }
   const int L = (int)levels.size();
   reorder_levels(); // solve-phase numbering of this rank's unknowns on the big partitioned levels
   for (int q = 0; q < L; q++)
   {
      AmgLevel    &lv = levels[(size_t)q];
      const size_t nq = (q == L - 1) ? (size_t)coarse_nloc : (size_t)level_A(q).nrows;
      if (q < L - 1) build_smoother_data(q);
      lv.ext = std::max<size_t>(nq + tail_len[(size_t)q], 1);
      if (q > 0) { lv.f.alloc(lv.ext); lv.u.alloc(lv.ext); }
      lv.u2.alloc(lv.ext);
      lv.t.alloc(lv.ext);
   }
   Context::get().sync();
   HDA_TRACE("partitioned setup: %d partitioned levels + replicated tail of %d levels", L - 1, tail->num_levels());
   if (const char *chk = getenv("HDA_DIST_CHECK"); chk && *chk && *chk != '0')
   { // development aid: the replicated setup is the specification of this one
      Amg ref(prm);
      ref.setup_dist(Aloc, hA0_, part0, ghost_gids0);
      auto cmp = [&](const char *what, int q, const DCsr &a, const DCsr &b, double tol) {
         const std::string tag = std::string(what) + " on level " + std::to_string(q) + " (" + std::to_string(a.nrows) + "x" + std::to_string(a.ncols) + "/" + std::to_string(a.nnz) +
                                 " vs " + std::to_string(b.nrows) + "x" + std::to_string(b.ncols) + "/" + std::to_string(b.nnz) + ")";
         HDA_REQUIRE(a.nrows == b.nrows && a.ncols == b.ncols && a.nnz == b.nnz, ("HDA_DIST_CHECK: shape differs: " + tag).c_str());
         std::vector<int>    ra = a.rowptr.to_host(), rb = b.rowptr.to_host(), ca((size_t)std::max(a.nnz, 1)), cb((size_t)std::max(a.nnz, 1));
         std::vector<double> va((size_t)std::max(a.nnz, 1)), vb((size_t)std::max(a.nnz, 1));
         if (a.nnz) { a.col.download(ca.data(), (size_t)a.nnz); b.col.download(cb.data(), (size_t)a.nnz); a.val.download(va.data(), (size_t)a.nnz); b.val.download(vb.data(), (size_t)a.nnz); }
         HDA_REQUIRE(std::equal(ra.begin(), ra.begin() + a.nrows + 1, rb.begin()), "HDA_DIST_CHECK: row pointers differ");
         double worst = 0.0, big = 0.0;
         for (int k = 0; k < a.nnz; k++)
         {
            HDA_REQUIRE(ca[(size_t)k] == cb[(size_t)k], ("HDA_DIST_CHECK: columns differ: " + tag).c_str());
            worst = std::max(worst, std::fabs(va[(size_t)k] - vb[(size_t)k]));
            big   = std::max(big, std::fabs(vb[(size_t)k]));
         }
         HDA_REQUIRE(worst <= tol * std::max(big, 1e-300), ("HDA_DIST_CHECK: values differ by " + std::to_string(worst / std::max(big, 1e-300)) + ": " + tag).c_str());
         fprintf(stderr, "[hda] dist check rank %d level %d %s: identical pattern, max rel diff %.2e\n", cm.rank, q, what, worst / std::max(big, 1e-300));
      };
      HDA_REQUIRE(ref.num_levels() == num_levels(), "HDA_DIST_CHECK: level count differs");
      for (int q = 0; q < L - 1; q++)
      {
         if (q > 0) cmp("A", q, levels[(size_t)q].A, ref.levels[(size_t)q].A, q == 1 ? 1e-13 : 1e-10);
         cmp("P", q, levels[(size_t)q].P, ref.levels[(size_t)q].P, q == 0 ? 0.0 : 1e-10);
         cmp("R", q, levels[(size_t)q].R, ref.levels[(size_t)q].R, q == 0 ? 0.0 : 1e-10);
      }
      const int mine = num_levels() + (tail ? tail->num_levels() - 1 : 0), theirs = ref.num_levels() + (ref.tail ? ref.tail->num_levels() - 1 : 0);
      HDA_REQUIRE(mine == theirs, "HDA_DIST_CHECK: total hierarchy depth differs");
   }
}

} // namespace hda
