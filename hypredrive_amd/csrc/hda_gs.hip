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

#include <algorithm>

namespace hda {

#define STREAM (Context::get().stream)

// distinct neighbours of row i in pattern(A) U pattern(A^T), restricted to owned columns,
// visited in ascending order; f(j) is called once per neighbour j != i
// (lo, hi: only neighbours in [lo, hi) count -- the row's own block in the row-block form, [0, n) otherwise)
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
__global__ __launch_bounds__(256) void k_gs_mark(int n, const int *__restrict__ perm, const int *__restrict__ lvl_of_pos,
                                                 int *__restrict__ level_of_row)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) level_of_row[perm[q]] = lvl_of_pos[q];
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

// ---- sweep-order copy of the operator for the row-block sweeps of big levels
__global__ __launch_bounds__(256) void k_gs_inverse(int n, const int *__restrict__ perm, const int *__restrict__ rp, int *__restrict__ pos_of,
                                                    int *__restrict__ len)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   const int i = perm[q];
   pos_of[i]   = q;
   len[q]      = rp[i + 1] - rp[i];
}
__global__ __launch_bounds__(256) void k_gs_sorted_fill(int n, int nb, const int *__restrict__ part, const int *__restrict__ perm,
                                                        const int *__restrict__ pos_of, const int *__restrict__ rp, const int *__restrict__ cj,
                                                        const double *__restrict__ v, const int *__restrict__ srp, int *__restrict__ scj,
                                                        double *__restrict__ sv)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   const int i = perm[q];
   int       lo, hi;
   gs_block_of(part, nb, i, n, lo, hi);
   int d = srp[q];
   for (int k = rp[i]; k < rp[i + 1]; k++, d++)
   { // same entry order as the row: same order of additions as the row-ordered kernel
      const int c = cj[k];
      scj[d]      = (c >= lo && c < hi) ? pos_of[c] : ~c;
      sv[d]       = v[k];
   }
}
static void gs_sorted_copy(const DCsr &A, const GsPlan &plan)
{
   const int n = A.nrows;
   DArray<int> pos_of((size_t)n), len((size_t)n + 1);
   k_gs_inverse<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.perm.data(), A.rowptr.data(), pos_of.data(), len.data());
   plan.s_rowptr.alloc((size_t)n + 1);
   exclusive_scan(n, len.data(), plan.s_rowptr.data(), nullptr);
   plan.s_col.alloc((size_t)std::max(A.nnz, 1));
   plan.s_val.alloc((size_t)std::max(A.nnz, 1));
   k_gs_sorted_fill<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.nblk, plan.blk_part.data(), plan.perm.data(), pos_of.data(), A.rowptr.data(),
                                                         A.col.data(), A.val.data(), plan.s_rowptr.data(), plan.s_col.data(), plan.s_val.data());
   if (plan.s_x.size() != (size_t)n) { plan.s_x.alloc((size_t)n); plan.s_b.alloc((size_t)n); plan.s_d.alloc((size_t)n); }
   plan.sorted = true;
}

// dependency levels of the symmetrised pattern (restricted to the row blocks of part when given): rows in discovery order in
// plan.perm, level offsets in plan.lvl_ptr; returns the level of every row (host)
static std::vector<int> gs_levels(const DCsr &A, GsPlan &plan, const int *d_part, int nb)
{
   const int n = A.nrows;
   DCsr T;
   transpose(A, T); // rows of T = columns of A; only rows < n are consulted
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
   std::vector<int> lvl_of_pos((size_t)n);
   for (int L = 0; L < plan.nlev; L++)
      for (int q = plan.lvl_ptr[(size_t)L]; q < plan.lvl_ptr[(size_t)L + 1]; q++) lvl_of_pos[(size_t)q] = L;
   DArray<int> dpos, lrow((size_t)n);
   dpos.upload(lvl_of_pos.data(), (size_t)n);
   k_gs_mark<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, plan.perm.data(), dpos.data(), lrow.data());
   return lrow.to_host();
}

void build_gs_plan(const DCsr &A, GsPlan &plan)
{
   const int n = A.nrows;
   plan        = GsPlan();
   plan.built  = true;
   plan.perm.alloc((size_t)std::max(n, 1));
   plan.lvl_ptr.assign(1, 0);
   if (n == 0) return;
   // level of every row -> stable counting order = ascending rows inside each level
   {
      std::vector<int> hl = gs_levels(A, plan, nullptr, 0), cursor(plan.lvl_ptr.begin(), plan.lvl_ptr.end() - 1), hp((size_t)n);
      for (int i = 0; i < n; i++) hp[(size_t)cursor[(size_t)hl[(size_t)i]]++] = i;
      plan.perm.upload(hp.data(), (size_t)n);
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
   std::vector<int> bl_ptr((size_t)nb + 1, 0), bl;
   if (n == 0)
   {
      bl.push_back(0);
      plan.blk_lvl_ptr.upload(bl_ptr.data(), bl_ptr.size());
      plan.blk_lvl.upload(bl.data(), bl.size());
      return;
   }
   const std::vector<int> hl = gs_levels(A, plan, plan.blk_part.data(), nb);
   std::vector<int>       hp((size_t)n), cnt;
   for (int q = 0; q < nb; q++)
   {
      const int lo = part[(size_t)q], hi = part[(size_t)q + 1];
      HDA_REQUIRE(lo <= hi, "row blocks must ascend");
      int nl = 0;
      for (int i = lo; i < hi; i++) nl = std::max(nl, hl[(size_t)i] + 1);
      cnt.assign((size_t)nl + 1, 0);
      for (int i = lo; i < hi; i++) cnt[(size_t)hl[(size_t)i] + 1]++;
      for (int L = 0; L < nl; L++)
      {
         cnt[(size_t)L + 1] += cnt[(size_t)L];
         bl.push_back(lo + cnt[(size_t)L]);
      }
      for (int i = lo; i < hi; i++) hp[(size_t)(lo + cnt[(size_t)hl[(size_t)i]]++)] = i; // ascending rows inside a level
      bl_ptr[(size_t)q + 1]  = (int)bl.size();
      plan.blk_max_levels = std::max(plan.blk_max_levels, nl);
   }
   bl.push_back(n);
   plan.perm.upload(hp.data(), (size_t)n);
   plan.blk_lvl_ptr.upload(bl_ptr.data(), bl_ptr.size());
   plan.blk_lvl.upload(bl.data(), bl.size());
   gs_row_spans(A, plan);
   const int sorted_min = getenv("HDA_GS_SORTED_MIN") ? atoi(getenv("HDA_GS_SORTED_MIN")) : 50000; // (read per plan: the tests move it)
   if (n >= sorted_min) gs_sorted_copy(A, plan);
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
                                                           double *__restrict__ sx, double *__restrict__ sb, double *__restrict__ sd)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q >= n) return;
   const int i = perm[q];
   sx[q]       = zero_in ? 0.0 : xin[i];
   sb[q]       = b[i];
   sd[q]       = dinv[i];
}
__global__ __launch_bounds__(256) void k_gs_from_sweep_order(int n, const int *__restrict__ perm, const double *__restrict__ sx, double *__restrict__ xout)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) xout[perm[q]] = sx[q];
}
template <int LPR, int NPF>
__global__ __launch_bounds__(1024) void k_gs_blocks_sorted(int backward, int zero_in, int lds_levels, const int *__restrict__ blk_lvl_ptr,
                                                           const int *__restrict__ blk_lvl, const int *__restrict__ srp,
                                                           const int *__restrict__ scj, const double *__restrict__ sv,
                                                           const double *__restrict__ sd, const double *__restrict__ sb, const double *xin,
                                                           double *sx)
{
   extern __shared__ int slp_lds[];
   const int blk = blockIdx.x;
   const int L0 = blk_lvl_ptr[blk], nl = blk_lvl_ptr[blk + 1] - L0;
   const int tid = threadIdx.x, lane = tid & (LPR - 1), q = tid / LPR;
   constexpr int RP = 1024 / LPR; // rows per pass
   const int *slp = blk_lvl + L0;
   if (nl + 1 <= lds_levels)
   {
      for (int t = tid; t <= nl; t += 1024) slp_lds[t] = blk_lvl[L0 + t];
      slp = slp_lds;
      __syncthreads();
   }
   struct It { int s, p; };
   auto level = [&](const It &it) { return backward ? nl - 1 - it.s : it.s; };
   auto advance = [&](It it) {
      if (it.s >= nl) return it;
      const int L = level(it);
      it.p++;
      if (it.p * RP >= slp[L + 1] - slp[L]) { it.s++; it.p = 0; }
      return it;
   };
   struct RowA { int pos, k0, k1; bool has; };
   struct RowB { int pos, k0, k1; bool has; int c[NPF]; double a[NPF]; double d, rhs; };
   auto stage_a = [&](const It &it) {
      RowA r;
      r.has = false; r.pos = 0; r.k0 = 0; r.k1 = 0;
      if (it.s < nl)
      {
         const int L = level(it), pos = slp[L] + it.p * RP + q;
         if (pos < slp[L + 1])
         {
            r.has = true;
            r.pos = pos;
            r.k0  = srp[pos];
            r.k1  = srp[pos + 1];
         }
      }
      return r;
   };
   auto stage_b = [&](const RowA &ra) {
      RowB r;
      r.pos = ra.pos; r.k0 = ra.k0; r.k1 = ra.k1; r.has = ra.has; r.d = 0.0; r.rhs = 0.0;
#pragma unroll
      for (int u = 0; u < NPF; u++) { r.c[u] = 0; r.a[u] = 0.0; }
      if (ra.has)
      {
#pragma unroll
         for (int u = 0; u < NPF; u++)
         {
            const int k = ra.k0 + lane + u * LPR;
            if (k < ra.k1) { r.c[u] = scj[k]; r.a[u] = sv[k]; }
            else r.a[u] = 0.0; // (c = 0 with a = 0: the value read for it never reaches the sum, see below)
         }
         r.d   = sd[ra.pos];
         r.rhs = sb[ra.pos];
      }
      return r;
   };
   auto value = [&](int c) { return (c >= 0) ? sx[c] : (zero_in ? 0.0 : xin[~c]); };
   It   itC = {0, 0}, itB = advance(itC), itA = advance(itB);
   RowB cur = stage_b(stage_a(itC));
   RowA nxa = stage_a(itB);
   while (itC.s < nl)
   {
      // the gathers of this pass go out FIRST: loads return in issue order, so requests made before them would have to land before them
      double xs[NPF];
      bool   used[NPF];
#pragma unroll
      for (int u = 0; u < NPF; u++)
      {
         used[u] = cur.has && (cur.k0 + lane + u * LPR < cur.k1);
         xs[u]   = used[u] ? value(cur.c[u]) : 0.0;
      }
      const RowA nx2 = stage_a(itA); // two passes ahead
      const RowB nxb = stage_b(nxa); // one pass ahead
      if (cur.has)
      {
         double sum = 0.0;
#pragma unroll
         for (int u = 0; u < NPF; u++)
            if (used[u]) sum += cur.a[u] * xs[u];
         for (int k = cur.k0 + lane + NPF * LPR; k < cur.k1; k += LPR) sum += sv[k] * value(scj[k]); // rows longer than the prefetch
#pragma unroll
         for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
         if (lane == 0) sx[cur.pos] += cur.d * (cur.rhs - sum);
      }
      const It nextC = itB;
      if (nextC.s != itC.s)
      {
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

template <int LPR, int NPF>
static void gs_blocks_sorted_t(const DCsr &A, const GsPlan &p, const double *dinv, const double *b, const double *xin, double *xout, bool forward,
                               bool zero_in)
{
   const int    n          = A.nrows;
   const int    lds_levels = std::min(p.blk_max_levels + 1, 12 * 1024);
   const size_t lds        = sizeof(int) * (size_t)lds_levels;
   k_gs_to_sweep_order<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, zero_in ? 1 : 0, p.perm.data(), xin, b, dinv, p.s_x.data(), p.s_b.data(), p.s_d.data());
   k_gs_blocks_sorted<LPR, NPF><<<p.nblk, 1024, lds, STREAM>>>(forward ? 0 : 1, zero_in ? 1 : 0, lds_levels, p.blk_lvl_ptr.data(), p.blk_lvl.data(),
                                                              p.s_rowptr.data(), p.s_col.data(), p.s_val.data(), p.s_d.data(), p.s_b.data(), xin,
                                                              p.s_x.data());
   k_gs_from_sweep_order<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, p.perm.data(), p.s_x.data(), xout);
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
                     bool zero_in)
{
   HDA_REQUIRE(plan.built && plan.nblk > 0, "row-block Gauss-Seidel plan missing");
   HDA_REQUIRE(zero_in || (xin && xin != xout), "row-block Gauss-Seidel sweeps out of place");
   if (A.nrows == 0) return;
   if (plan.span_rp != A.rowptr.data() || plan.span_nnz != A.nnz || plan.span_gen != A.gen)
   { // another matrix behind a kept plan (preconditioner.reuse)
      gs_row_spans(A, plan);
      if (plan.sorted) gs_sorted_copy(A, plan);
   }
   const double a = A.avg_row();
   const bool use_sorted = !(getenv("HDA_GS_SORTED") && atoi(getenv("HDA_GS_SORTED")) == 0);
   if (plan.sorted && use_sorted)
   {
      if (a <= 10.0) gs_blocks_sorted_t<2, 4>(A, plan, dinv, b, xin, xout, forward, zero_in); // (two lanes per row: 42.7 -> 41.1 ms per 128^3 solve against four, eight 48.2)
      else if (a <= 40.0) gs_blocks_sorted_t<8, 8>(A, plan, dinv, b, xin, xout, forward, zero_in);
      else gs_blocks_sorted_t<16, 8>(A, plan, dinv, b, xin, xout, forward, zero_in);
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
         static const int pipe = getenv("HDA_GS_PIPE") ? atoi(getenv("HDA_GS_PIPE")) : 1;
         const size_t     lds  = sizeof(int) * (size_t)(sg.second - sg.first + 1);
         if (pipe && lds <= 48 * 1024)
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
