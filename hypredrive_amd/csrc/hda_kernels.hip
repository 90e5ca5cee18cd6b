// hda_kernels.hip -- solve-phase kernels for CDNA4 (gfx950, wave64).
//
// All of these are HBM-bandwidth bound (SpMV arithmetic intensity ~0.13 flop/B), so the
// design rules are the memory ones: a lane group of LPR lanes walks one CSR row so that a
// wave reads a contiguous run of (col,val) pairs; two independent rows per lane group are
// in flight to cover HBM latency; per-row sums are combined with wave shuffles; every
// dot product is produced as kRedBlocks block partials in a fixed tree (deterministic) and
// fused into the kernel that already streams the operands.
#include "hda_kernels.h"
#include "hda_sort.h"

#include "hda_dist.h"
#include "hda_mpi_join.h"
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace hda {

// ------------------------------------------------------------------ context

Context::Context()
{
   HDA_HIP(hipGetDevice(&device));
   HDA_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
   {
      int lo = 0, hi = 0; // numerically lower = higher priority: the small transfer kernels should not queue behind product grids
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); hi = 0; }
      if (hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, hi) != hipSuccess)
      {
         (void)hipGetLastError();
         HDA_HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
      }
   }
   HDA_HIP(hipMalloc((void **)&partials, sizeof(double) * kNumSlots * kRedBlocks));
   HDA_HIP(hipMemset(partials, 0, sizeof(double) * kNumSlots * kRedBlocks));
   HDA_HIP(hipMalloc((void **)&scalars, sizeof(double) * kNumScalars));
   HDA_HIP(hipMemset(scalars, 0, sizeof(double) * kNumScalars));
   HDA_HIP(hipHostMalloc((void **)&host_scalars, sizeof(double) * kNumScalars, hipHostMallocDefault));
   HDA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
}

// ---- host waits with an optional limit (hda_common.h)
namespace {
const char *g_stage_name = "idle";
template <class Query>
void wait_limited(Query done, const char *what)
{
   const double lim = wait_limit_s();
   const auto   t0  = std::chrono::steady_clock::now();
   for (long spin = 0;; spin++)
   {
      hipError_t e = done();
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) HDA_HIP(e);
      (void)hipGetLastError();
      if (spin > 2000) usleep(20);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim)
      {
         fprintf(stderr, "[hypredrv_amd] rank %d of %d (%s transport): a %s wait did not end within %.1f s (HDA_COMM_TIMEOUT_S) during %s: a peer "
                         "rank has failed or left the collective sequence, or a kernel hangs; ending the job\n", Comm::world().rank, Comm::world().size,
                 Comm::world().name(), what, lim, g_stage_name);
         abort_job(86);
      }
   }
}
} // namespace
double wait_limit_s()
{
   static const double t = [] { const char *e = getenv("HDA_COMM_TIMEOUT_S"); return e ? atof(e) : 0.0; }();
   return t;
}
void set_stage(const char *stage) { g_stage_name = stage; }
const char *current_stage() { return g_stage_name; }
void wait_stream(hipStream_t s)
{
   if (wait_limit_s() <= 0.0) { HDA_HIP(hipStreamSynchronize(s)); return; }
   wait_limited([s] { return hipStreamQuery(s); }, "stream");
}
void wait_event(hipEvent_t e)
{
   if (wait_limit_s() <= 0.0) { HDA_HIP(hipEventSynchronize(e)); return; }
   wait_limited([e] { return hipEventQuery(e); }, "event");
}
void abort_job(int status)
{
   fflush(nullptr);
   mpi_abort(0x44000000 /* MPI_COMM_WORLD of the MPICH ABI; the joined duplicate is used when there is one */, status);
   _exit(status);
}

// ---- process-global state with a private copy per thread rank (hda_common.h)
namespace {
thread_local bool                        tl_rank = false;
thread_local std::vector<void (*)()>    *tl_drops = nullptr;
thread_local Context                    *tl_ctx  = nullptr;
} // namespace
bool in_thread_rank() { return tl_rank; }
void enter_thread_rank()
{
   tl_rank = true;
   if (!tl_drops) tl_drops = new std::vector<void (*)()>();
}
void thread_rank_on_leave(void (*drop)())
{
   if (tl_drops) tl_drops->push_back(drop);
}
void leave_thread_rank()
{
   if (!tl_rank) return;
   Comm::set_world(nullptr);   // this thread's communicator
   Context::release_thread();  // stream syncs, cached blocks back to the driver, reduction scratch
   if (tl_drops)
   {
      for (size_t q = tl_drops->size(); q-- > 0;) (*tl_drops)[q](); // (a dropper must not create state)
      delete tl_drops;
      tl_drops = nullptr;
   }
   delete tl_ctx;
   tl_ctx  = nullptr;
   tl_rank = false;
}

// One context per PROCESS (the reference's contract is one thread at a time, include/HYPREDRV.h:66-70: any thread may make the next
// call); a thread rank of the test seam has its own -- stream, reduction scratch, allocator, communicator.
Context &Context::get()
{
   if (!tl_rank)
   {
      static Context ctx;
      return ctx;
   }
   if (!tl_ctx) tl_ctx = new Context();
   return *tl_ctx;
}
void Context::release_thread()
{ // the thread-rank harness: give back what this thread's context and allocator hold before the thread ends
   if (!tl_rank || !tl_ctx) return;
   Context &c = get();
   (void)hipStreamSynchronize(c.stream);
   (void)hipStreamSynchronize(c.comm_stream);
   pool_trim();
   (void)hipFree(c.partials); (void)hipFree(c.scalars); (void)hipHostFree(c.host_scalars);
   (void)hipEventDestroy(c.ev); (void)hipStreamDestroy(c.comm_stream); (void)hipStreamDestroy(c.stream);
   c.partials = c.scalars = c.host_scalars = nullptr; c.ev = nullptr; c.stream = c.comm_stream = nullptr;
}

// ---------------------------------------------------------------- allocator

namespace {
struct Cached {
   void    *p;
   uint64_t stamp; // release order: the oldest blocks go first when the cache is over its limit
};
struct Pool;
// every live block's pool: a block may be released by another thread than the one that asked for it (a finalizer thread of the
// caller's language runtime; a thread rank's object destroyed after its thread has left) and must find its way home
std::mutex                          g_pool_mutex; // guards every Pool and the two maps below
std::unordered_map<void *, Pool *> &block_owner()
{
   static std::unordered_map<void *, Pool *> m;
   return m;
}
std::vector<Pool *> &all_pools() // the process's pool and every thread rank's (under g_pool_mutex)
{
   static std::vector<Pool *> v;
   return v;
}
struct Pool {
   std::multimap<size_t, Cached>      free_;  // size -> block
   std::unordered_map<void *, size_t> size_;  // every block we own
   size_t                             in_use = 0, cached = 0, peak = 0;
   uint64_t                           clock = 0;
   Pool() { all_pools().push_back(this); } // (made by current_pool() only, whose callers hold g_pool_mutex)
   ~Pool();
   void drop_cached() // caller holds g_pool_mutex; hipFree waits for the device, so a block a queued kernel still reads is safe to return
   {
      for (auto &kv : free_)
      {
         size_.erase(kv.second.p);
         block_owner().erase(kv.second.p);
         (void)hipFree(kv.second.p);
      }
      free_.clear();
      cached = 0;
   }
};
Pool &process_pool()
{
   static Pool *p = new Pool(); // (never destroyed: blocks are released by static destructors of other translation units)
   return *p;
}
thread_local Pool *tl_pool = nullptr; // a thread rank's own pool, like its context whose stream orders its blocks
void drop_tl_pool()
{
   delete tl_pool;
   tl_pool = nullptr;
}
Pool &current_pool()
{
   if (!in_thread_rank()) return process_pool();
   if (!tl_pool)
   {
      tl_pool = new Pool();
      thread_rank_on_leave(&drop_tl_pool);
   }
   return *tl_pool;
}
Pool::~Pool()
{ // a thread rank leaves: cached blocks go back to the driver, blocks still in use move to the process's pool
   std::lock_guard<std::mutex> lk(g_pool_mutex);
   drop_cached();
   auto &all = all_pools();
   all.erase(std::remove(all.begin(), all.end(), this), all.end());
   Pool &g = process_pool();
   if (&g == this) return;
   for (auto &kv : size_)
   {
      g.size_[kv.first] = kv.second;
      g.in_use += kv.second;
      block_owner()[kv.first] = &g;
   }
}
constexpr size_t kAlign = 512;
// what the driver's allocator cost: calls of hipMalloc that reached the driver and the host time spent in them (bench.py reports both
// for the first and the second setup of a process: on some boxes of the pool the first setup pays several hundred ms here)
std::atomic<long long> g_malloc_calls{0}, g_malloc_ns{0}, g_malloc_bytes{0};
hipError_t timed_hipMalloc(void **p, size_t bytes)
{
   const auto       t0 = std::chrono::steady_clock::now();
   const hipError_t e  = hipMalloc(p, bytes);
   g_malloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
   g_malloc_calls++;
   if (e == hipSuccess) g_malloc_bytes += (long long)bytes;
   return e;
}
} // namespace
void pool_driver_stats(double out[3], bool reset)
{
   out[0] = (double)g_malloc_calls.load();
   out[1] = (double)g_malloc_ns.load() * 1e-6;
   out[2] = (double)g_malloc_bytes.load();
   if (reset) { g_malloc_calls = 0; g_malloc_ns = 0; g_malloc_bytes = 0; }
}
#define g_pool (current_pool())

// HDA_POISON=1 (diagnostics): every block handed out is filled with 0xFF bytes (NaN as double, -1 as int) on the library
// stream first, so a kernel that reads memory nobody wrote fails the tests instead of depending on what the block held before
static void poison(void *p, size_t bytes)
{
   static const bool on = getenv("HDA_POISON") && *getenv("HDA_POISON") && *getenv("HDA_POISON") != '0';
   if (on && p) (void)hipMemsetAsync(p, 0xFF, bytes, Context::get().stream);
}

// HDA_GUARD=1 (diagnostics): every block carries a 512-byte tail filled with 0xA5 that is checked when the block is
// released; a kernel that writes past the end of its buffer aborts the process with a message instead of corrupting a neighbour
static bool guard_on()
{
   static const bool on = getenv("HDA_GUARD") && *getenv("HDA_GUARD") && *getenv("HDA_GUARD") != '0';
   return on;
}
constexpr size_t kGuardBytes = 512;
static std::unordered_map<void *, size_t> g_guard_at; // block -> offset of its guard tail (under g_pool_mutex)

void *pool_alloc(size_t bytes)
{
   if (bytes == 0) return nullptr;
   std::unique_lock<std::mutex> lk(g_pool_mutex);
   const size_t user = (bytes + kAlign - 1) / kAlign * kAlign;
   const size_t want = user + (guard_on() ? kGuardBytes : 0);
   // best fit, but never waste more than 25 % (+64 KiB) of a cached block
   auto it = g_pool.free_.lower_bound(want);
   if (it != g_pool.free_.end() && it->first <= want + want / 4 + 65536)
   {
      void *p = it->second.p;
      g_pool.cached -= it->first;
      g_pool.in_use += it->first;
      g_pool.free_.erase(it);
      g_pool.peak = std::max(g_pool.peak, g_pool.in_use);
      poison(p, want);
      if (guard_on())
      {
         (void)hipMemsetAsync((char *)p + user, 0xA5, kGuardBytes, Context::get().stream);
         g_guard_at[p] = user;
      }
      return p;
   }
   void      *p = nullptr;
   hipError_t e = timed_hipMalloc(&p, want);
   if (e != hipSuccess)
   {
      (void)hipGetLastError();
      lk.unlock();
      pool_trim(); // give cached blocks back and retry once
      lk.lock();
      e = timed_hipMalloc(&p, want);
   }
   if (e != hipSuccess)
   { // thread ranks share the device: what is short may sit in the caches of the other ranks' pools (each keeps up to twice its peak)
      (void)hipGetLastError();
      for (Pool *q : all_pools()) q->drop_cached();
      e = hipMalloc(&p, want);
   }
   if (e != hipSuccess)
   {
      char buf[256];
      snprintf(buf, sizeof(buf), "device allocation of %zu bytes failed: %s (in use %zu)", want, hipGetErrorString(e), g_pool.in_use);
      throw Error(buf);
   }
   g_pool.size_[p] = want;
   block_owner()[p] = &g_pool;
   g_pool.in_use += want;
   g_pool.peak = std::max(g_pool.peak, g_pool.in_use);
   poison(p, want);
   if (guard_on())
   {
      (void)hipMemsetAsync((char *)p + user, 0xA5, kGuardBytes, Context::get().stream);
      g_guard_at[p] = user;
   }
   return p;
}

#undef g_pool
void pool_free(void *p)
{
   if (!p) return;
   std::lock_guard<std::mutex> lk(g_pool_mutex);
   auto ow = block_owner().find(p);
   if (ow == block_owner().end())
   { // not a block of this allocator (or released twice): say so instead of leaking or corrupting quietly
      fprintf(stderr, "[hda] pool_free: %p is not a live block of the device allocator (ignored)\n", p);
      return;
   }
   Pool &g_pool = *ow->second; // the pool the block came from, whichever thread releases it
   auto  it     = g_pool.size_.find(p);
   if (it == g_pool.size_.end()) return;
   if (guard_on())
   {
      auto g = g_guard_at.find(p);
      if (g != g_guard_at.end())
      {
         unsigned char tail[kGuardBytes];
         (void)hipStreamSynchronize(Context::get().stream);
         if (hipMemcpy(tail, (char *)p + g->second, kGuardBytes, hipMemcpyDeviceToHost) == hipSuccess)
            for (size_t q = 0; q < kGuardBytes; q++)
               if (tail[q] != 0xA5)
               {
                  fprintf(stderr, "HDA_GUARD: a kernel wrote %zu bytes past the end of a %zu-byte device buffer\n", q + 1, g->second);
                  abort();
               }
         g_guard_at.erase(g);
      }
   }
   g_pool.in_use -= it->second;
   g_pool.cached += it->second;
   g_pool.free_.emplace(it->second, Cached{p, ++g_pool.clock});
   // Released blocks are kept for the next setup / solve of the same shape (on boxes with a slow hipMalloc a cold AMG setup takes three
   // times the warm one), but never more than TWICE what this thread ever had in use at once (at least HDA_POOL_CACHE_MIN_GB, default 4):
   // a long-lived process that has solved many differently sized systems must not sit on the whole of HBM -- other processes and other
   // rank threads allocate from it too.  (Twice, not once: after a setup the allocator owns about 1.2x the peak -- temporaries freed
   // early cannot all be reused later -- and a bound of 1x evicted exactly the big level-0 temporaries the next setup asks for first.)
   static const size_t floor_bytes = [] {
      const char *e = getenv("HDA_POOL_CACHE_MIN_GB");
      return (size_t)((e && *e ? atof(e) : 4.0) * (double)(1ull << 30));
   }();
   const size_t cap = std::max(floor_bytes, 2 * g_pool.peak);
   if (g_pool.cached > cap)
   {
      std::vector<std::multimap<size_t, Cached>::iterator> by_age;
      for (auto q = g_pool.free_.begin(); q != g_pool.free_.end(); ++q) by_age.push_back(q);
      std::sort(by_age.begin(), by_age.end(), [](auto a, auto b) { return a->second.stamp < b->second.stamp; });
      // (hipFree waits for the device, so a block a queued kernel still uses is safe to return)
      for (auto q : by_age)
      {
         if (g_pool.cached <= cap - cap / 4) break; // some slack: not one hipFree per release from here on
         g_pool.cached -= q->first;
         g_pool.size_.erase(q->second.p);
         block_owner().erase(q->second.p);
         (void)hipFree(q->second.p);
         g_pool.free_.erase(q);
      }
   }
}
#define g_pool (current_pool())

void pool_trim()
{
   (void)hipStreamSynchronize(Context::get().stream);
   std::lock_guard<std::mutex> lk(g_pool_mutex);
   g_pool.drop_cached();
}
size_t pool_bytes_in_use() { std::lock_guard<std::mutex> lk(g_pool_mutex); return g_pool.in_use; }
size_t pool_bytes_peak() { std::lock_guard<std::mutex> lk(g_pool_mutex); return g_pool.peak; }
size_t pool_bytes_cached() { std::lock_guard<std::mutex> lk(g_pool_mutex); return g_pool.cached; }
#undef g_pool

#define STREAM (Context::get().stream)

// ------------------------------------------------------- device reductions

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
   for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
   return v;
}

// fixed-order sum over a 256-thread block; every thread gets the result
__device__ __forceinline__ double block_sum(double v)
{
   __shared__ double sm[4];
   v = wave_sum(v);
   if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
   __syncthreads();
   double t = (sm[0] + sm[1]) + (sm[2] + sm[3]);
   __syncthreads();
   return t;
}

template <int LPR>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
   for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
   return v;
}

__global__ __launch_bounds__(kRedThreads) void k_finalize(const double *__restrict__ partials,
                                                           int first_slot, double *scalars,
                                                           int first_scalar)
{
   const double *p = partials + (size_t)(first_slot + blockIdx.x) * kRedBlocks;
   double        s = 0.0;
   for (int i = threadIdx.x; i < kRedBlocks; i += kRedThreads) s += p[i];
   s = block_sum(s);
   if (threadIdx.x == 0) scalars[first_scalar + blockIdx.x] = s;
}

void finalize_n(int first_slot, int nslots, int first_scalar)
{
   Context &c = Context::get();
   k_finalize<<<nslots, kRedThreads, 0, c.stream>>>(c.partials, first_slot, c.scalars, first_scalar);
   // row-partitioned: the scalars are partial sums of this rank's rows (C2: one fused all-reduce)
   Comm &cm = Comm::world();
   if (cm.size > 1) cm.allreduce_sum_dev(c.scalars + first_scalar, nslots);
}
void finalize(int slot, int scalar_idx) { finalize_n(slot, 1, scalar_idx); }

double read_scalar(int idx)
{
   Context &c = Context::get();
   double   v;
   HDA_HIP(hipMemcpyAsync(&v, c.scalars + idx, sizeof(double), hipMemcpyDeviceToHost, c.stream));
   c.sync();
   return v;
}

void read_scalars_async(int first, int count)
{
   Context &c = Context::get();
   HDA_HIP(hipMemcpyAsync(c.host_scalars + first, c.scalars + first, sizeof(double) * count,
                          hipMemcpyDeviceToHost, c.stream));
   HDA_HIP(hipEventRecord(c.ev, c.stream));
}

// -------------------------------------------------------------------- SpMV

enum { MODE_PLAIN = 0, MODE_RESID = 1, MODE_JACOBI = 2 };

// One lane group (LPR lanes) per row, two rows in flight per group, grid-stride.
//  MODE_PLAIN : out = alpha*Ax + beta*yin       DOT: partial += out_i * w_i
//  MODE_RESID : out = b - Ax
//  MODE_JACOBI: out = xin + dinv*(b - A xin)    DOT: partial += b_i * out_i
template <int LPR, int MODE, bool DOT>
__global__ __launch_bounds__(256) void k_spmv(int n, const int *__restrict__ rowptr,
                                              const int *__restrict__ col,
                                              const double *__restrict__ val,
                                              const double *__restrict__ x, double alpha,
                                              double beta, const double *yin,
                                              const double *__restrict__ b,
                                              const double *__restrict__ dinv,
                                              const double *__restrict__ w, double *out,
                                              double *__restrict__ partial, const double *__restrict__ dinv2 = nullptr,
                                              double *__restrict__ out2 = nullptr)
{ // out2 (MODE_PLAIN): a second result out2 = dinv2 .* out, as in k_spmv_stream
   const int  lane = threadIdx.x & (LPR - 1);
   const long G    = (long)gridDim.x * (256 / LPR);
   const long gid  = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
   double     acc  = 0.0;
   for (long r0 = gid; r0 < n; r0 += 2 * G)
   {
      const long r1 = r0 + G;
      const bool h1 = r1 < n;
      int        s0 = rowptr[r0], e0 = rowptr[r0 + 1];
      int        s1 = 0, e1 = 0;
      if (h1) { s1 = rowptr[r1]; e1 = rowptr[r1 + 1]; }
      int    k0 = s0 + lane, k1 = s1 + lane;
      double a0 = 0.0, a1 = 0.0;
      if (k0 < e0) a0 = val[k0] * x[col[k0]];
      if (k1 < e1) a1 = val[k1] * x[col[k1]];
      for (k0 += LPR; k0 < e0; k0 += LPR) a0 += val[k0] * x[col[k0]];
      for (k1 += LPR; k1 < e1; k1 += LPR) a1 += val[k1] * x[col[k1]];
      a0 = group_sum<LPR>(a0);
      a1 = group_sum<LPR>(a1);
      if (lane == 0)
      {
         double o0, o1 = 0.0;
         if (MODE == MODE_PLAIN)
         {
            o0 = (beta == 0.0) ? alpha * a0 : alpha * a0 + beta * yin[r0];
            if (h1) o1 = (beta == 0.0) ? alpha * a1 : alpha * a1 + beta * yin[r1];
            if (DOT) { acc += o0 * w[r0]; if (h1) acc += o1 * w[r1]; }
            if (out2) { out2[r0] = dinv2[r0] * o0; if (h1) out2[r1] = dinv2[r1] * o1; }
         }
         else if (MODE == MODE_RESID)
         {
            o0 = b[r0] - a0;
            if (h1) o1 = b[r1] - a1;
         }
         else
         {
            const double b0 = b[r0];
            o0              = x[r0] + dinv[r0] * (b0 - a0);
            if (DOT) acc += b0 * o0;
            if (h1)
            {
               const double b1 = b[r1];
               o1              = x[r1] + dinv[r1] * (b1 - a1);
               if (DOT) acc += b1 * o1;
            }
         }
         out[r0] = o0;
         if (h1) out[r1] = o1;
      }
   }
   if (DOT)
   {
      acc = block_sum(acc);
      if (threadIdx.x == 0) partial[blockIdx.x] = acc;
   }
}

static int pick_lpr(const DCsr &A)
{
   double a = A.avg_row();
   if (a <= 5.0) return 4;
   if (a <= 10.0) return 8;
   if (a <= 20.0) return 16;
   if (a <= 40.0) return 32;
   return 64;
}

// ---- LDS-staged streaming SpMV ---------------------------------------------------------
// The vector kernel above leaves lanes idle whenever row lengths do not match the lane-group
// width and issues narrow, row-shaped loads.  Here a workgroup streams a contiguous chunk of
// ~kChunk entries with fully coalesced (val, col) loads, multiplies by the gathered x and
// parks the products in LDS; rows are then reduced out of LDS by lane groups sized to the
// number of rows in the chunk.  Chunks are dealt to workgroups so that workgroups sharing an
// XCD (blockIdx % 8) walk a contiguous eighth of the matrix: every XCD's L2 then holds one
// window of x instead of all eight holding the same one.
constexpr int kChunk     = 2048;
constexpr int kMaxRowLds = 1024; // rows longer than this use the vector kernel

__global__ __launch_bounds__(256) void k_chunk_rows(int nchunks, int nrows, const int *__restrict__ rowptr, int *__restrict__ chunk_row)
{
   const int c = blockIdx.x * 256 + threadIdx.x;
   if (c > nchunks) return;
   if (c == nchunks) { chunk_row[c] = nrows; return; }
   const int target = c * kChunk;
   int       lo = 0, hi = nrows; // smallest r with rowptr[r] >= target
   while (lo < hi)
   {
      const int mid = (lo + hi) >> 1;
      if (rowptr[mid] < target) lo = mid + 1;
      else hi = mid;
   }
   chunk_row[c] = lo;
}
__global__ __launch_bounds__(256) void k_max_row(int n, const int *__restrict__ rowptr, int *mx)
{
   int m = 0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = max(m, rowptr[i + 1] - rowptr[i]);
   for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
   if ((threadIdx.x & 63) == 0) atomicMax(mx, m);
}

static void ensure_plan(const DCsr &A)
{
   if (A.maxrow >= 0) return;
   DArray<int> mx(1);
   mx.zero();
   if (A.nrows) k_max_row<<<std::min(ceil_div(A.nrows, 256), 1024), 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), mx.data());
   int m = 0;
   mx.download(&m, 1);
   A.maxrow  = m;
   A.nchunks = std::max(1, ceil_div(A.nnz, kChunk));
   A.chunk_row.alloc((size_t)A.nchunks + 1);
   k_chunk_rows<<<ceil_div(A.nchunks + 1, 256), 256, 0, STREAM>>>(A.nchunks, A.nrows, A.rowptr.data(), A.chunk_row.data());
}

// VC: value-coded operator (see "value-coded SpMV" below) -- the 8-byte value stream is replaced by
// one-byte codes into a 255-entry dictionary held in LDS; code 255 = read the value array
// SPLIT: row-partitioned product overlapped with its halo exchange -- entries whose column is a ghost (>= nown)
// contribute nothing here; k_offd_fix adds them once the ghost values have arrived
template <int MODE, bool DOT, bool VC, bool SPLIT>
__global__ __launch_bounds__(256) void k_spmv_stream(int nchunks, const int *__restrict__ chunk_row,
                                                     const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const double *__restrict__ val, const double *__restrict__ x,
                                                     double alpha, double beta, const double *yin,
                                                     const double *__restrict__ b, const double *__restrict__ dinv,
                                                     const double *__restrict__ w, double *out,
                                                     double *__restrict__ partial, const unsigned char *__restrict__ code,
                                                     const double *__restrict__ dval, int nown, const double *__restrict__ dinv2,
                                                     double *__restrict__ out2)
{ // out2 (MODE_PLAIN only): a second result out2 = dinv2 .* out -- the zero-guess Jacobi sweep of the next coarser level rides
  // on the restriction that produces its right-hand side
   extern __shared__ double prod[];
   __shared__ double sdict[VC ? 256 : 1];
   const int tid  = threadIdx.x;
   if (VC)
   {
      sdict[tid] = dval[tid];
      __syncthreads();
   }
   const int xcd  = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
   const int nper = (nchunks + 7) >> 3;
   double    acc  = 0.0;
   for (int i = slot; i < nper; i += nslot)
   {
      const int c = xcd * nper + i;
      if (c >= nchunks) break;
      const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
      if (r0 == r1) continue;
      const int k0 = rowptr[r0], k1 = rowptr[r1];
      // stage 1: products -> LDS, four independent (val, col, x) triples in flight per lane
      int k = k0 + tid;
for (; k + 768 < k1; k += 1024)
      {
         double v0, v1, v2, v3;
         if (VC)
         {
            const int q0 = code[k], q1 = code[k + 256], q2 = code[k + 512], q3 = code[k + 768];
            v0 = (q0 != 255) ? sdict[q0] : val[k];
            v1 = (q1 != 255) ? sdict[q1] : val[k + 256];
            v2 = (q2 != 255) ? sdict[q2] : val[k + 512];
            v3 = (q3 != 255) ? sdict[q3] : val[k + 768];
         }
         else { v0 = val[k]; v1 = val[k + 256]; v2 = val[k + 512]; v3 = val[k + 768]; }
         const int    c0 = col[k], c1 = col[k + 256], c2 = col[k + 512], c3 = col[k + 768];
         double x0, x1, x2, x3;
         if (SPLIT)
         {
            x0 = (c0 < nown) ? x[c0] : 0.0; x1 = (c1 < nown) ? x[c1] : 0.0;
            x2 = (c2 < nown) ? x[c2] : 0.0; x3 = (c3 < nown) ? x[c3] : 0.0;
         }
         else { x0 = x[c0]; x1 = x[c1]; x2 = x[c2]; x3 = x[c3]; }
         prod[k - k0]       = v0 * x0;
         prod[k - k0 + 256] = v1 * x1;
         prod[k - k0 + 512] = v2 * x2;
         prod[k - k0 + 768] = v3 * x3;
      }
      for (; k < k1; k += 256)
      {
         double vv;
         if (VC) { const int q0 = code[k]; vv = (q0 != 255) ? sdict[q0] : val[k]; }
         else vv = val[k];
         const int cc = col[k];
         prod[k - k0] = vv * ((!SPLIT || cc < nown) ? x[cc] : 0.0);
      }
      __syncthreads();
      // stage 2: L lanes per row, L = largest power of two with rows*L <= 256
      const int nr = r1 - r0;
      int       L  = 1;
      while (L < 64 && nr * (L << 1) <= 256) L <<= 1;
      const int lane = tid & (L - 1);
      for (int rr = tid / L; rr < nr; rr += 256 / L)
      {
         const int r = r0 + rr;
         const int s = rowptr[r] - k0, e = rowptr[r + 1] - k0;
         double    sum = 0.0;
         for (int q = s + lane; q < e; q += L) sum += prod[q];
         for (int o = L >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
         if (lane == 0)
         {
            double o0;
            if (MODE == MODE_PLAIN)
            {
               o0 = (beta == 0.0) ? alpha * sum : alpha * sum + beta * yin[r];
               if (DOT) acc += o0 * w[r];
               if (out2) out2[r] = dinv2[r] * o0;
            }
            else if (MODE == MODE_RESID) o0 = b[r] - sum;
            else
            {
               const double br = b[r];
               o0              = x[r] + dinv[r] * (br - sum);
               if (DOT) acc += br * o0;
            }
            out[r] = o0;
         }
      }
      __syncthreads();
   }
   if (DOT)
   {
      acc = block_sum(acc);
      if (tid == 0)
      {
         partial[blockIdx.x] = acc;
         if (blockIdx.x + gridDim.x < kRedBlocks) partial[blockIdx.x + gridDim.x] = 0.0; // reduced grid (>= kRedBlocks / 2)
      }
   }
}


// ---- stencil-coded SpMV ------------------------------------------------------------------
// HBM-bound kernels get faster only by moving fewer bytes.  A constant-coefficient operator
// (the 7-pt Laplacian of the headline benchmark, hypredrive's ps3d10pt7 examples, any level-0
// stencil matrix) repeats a handful of (column offset, value) pairs: entry k is then stored as
// one byte code[k] naming a dictionary pair, 12 B -> 1 B per entry.  Entries whose pair is not
// in the dictionary (ghost columns of a row block, boundary oddities) carry code 255 and are
// read from the plain CSR arrays, which stay in place for the setup kernels.  The arithmetic
// is unchanged: same doubles, same products, same order.
constexpr int      kDictSlots = 255;
constexpr uint64_t kEmptyKey  = 0xffffffffffffffffull;

__device__ __forceinline__ uint64_t pair_key(double v, int delta)
{
   uint64_t h = (uint64_t)__double_as_longlong(v) ^ ((uint64_t)(uint32_t)delta * 0x9e3779b97f4a7c15ull);
   h ^= h >> 29;
   h *= 0xbf58476d1ce4e5b9ull;
   h ^= h >> 32;
   return h == kEmptyKey ? 0 : h;
}

// pass 1: claim dictionary slots (first come, first served).  Rows i = first, first + stride, ...;
// a pair that finds the table full counts as a failure, and once the failures exceed `limit`
// the remaining rows give up: an operator without a small pair alphabet is rejected in
// microseconds (sampling pass) instead of probing a full table for every entry.
__global__ __launch_bounds__(256) void k_code_collect(int n, int stride, const int *__restrict__ rp, const int *__restrict__ cj,
                                                      const double *__restrict__ v, unsigned long long *keys, double *dval, int *ddelta,
                                                      int *fails, int limit)
{
   const long i = ((long)blockIdx.x * 256 + threadIdx.x) * stride;
   if (i >= n || *(volatile int *)fails > limit) return;
   int bad = 0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      if (bad)
      { // report at once: an operator that does not fit the table is given up as soon as the count passes the limit,
        // not after every sampled row has walked the full table for each of its entries
         if (atomicAdd(fails, bad) + bad > limit) return;
         bad = 0;
      }
      const double   val = v[k];
      const int      d   = cj[k] - (int)i;
      const uint64_t key = pair_key(val, d);
      int            s   = (int)(key % kDictSlots);
      bool           ok  = false;
      for (int probe = 0; probe < kDictSlots; probe++)
      {
         unsigned long long cur = keys[s];
         if (cur == kEmptyKey) cur = atomicCAS(&keys[s], (unsigned long long)kEmptyKey, (unsigned long long)key);
         if (cur == kEmptyKey) { dval[s] = val; ddelta[s] = d; ok = true; break; } // this thread claimed the slot
         if (cur == key) { ok = true; break; }                                    // present (or a hash twin: resolved in pass 2)
         s = (s + 1 == kDictSlots) ? 0 : s + 1;
      }
      bad += !ok;
   }
   if (bad) atomicAdd(fails, bad);
}
// pass 2: encode against the finished dictionary, exact comparison of the pair
__global__ __launch_bounds__(256) void k_code_encode(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                     const unsigned long long *__restrict__ keys, const double *__restrict__ dval,
                                                     const int *__restrict__ ddelta, unsigned char *__restrict__ code, int *escapes)
{
   __shared__ unsigned long long sk[256];
   __shared__ double             sv[256];
   __shared__ int                sd[256];
   if (threadIdx.x < kDictSlots) { sk[threadIdx.x] = keys[threadIdx.x]; sv[threadIdx.x] = dval[threadIdx.x]; sd[threadIdx.x] = ddelta[threadIdx.x]; }
   __syncthreads();
   const int i   = blockIdx.x * 256 + threadIdx.x;
   int       esc = 0;
   if (i < n)
      for (int k = rp[i]; k < rp[i + 1]; k++)
      {
         const double   val = v[k];
         const int      d   = cj[k] - i;
         const uint64_t key = pair_key(val, d);
         int            s   = (int)(key % kDictSlots), c = 255;
         for (int probe = 0; probe < kDictSlots; probe++)
         {
            const unsigned long long cur = sk[s];
            if (cur == kEmptyKey) break;
            if (cur == key)
            {
               if (__double_as_longlong(sv[s]) == __double_as_longlong(val) && sd[s] == d) c = s;
               break;
            }
            s = (s + 1 == kDictSlots) ? 0 : s + 1;
         }
         code[k] = (unsigned char)c;
         esc += (c == 255);
      }
   for (int o = 32; o > 0; o >>= 1) esc += __shfl_xor(esc, o);
   if ((threadIdx.x & 63) == 0 && esc) atomicAdd(escapes, esc);
}
static bool coded_enabled()
{
   static const bool on = !(getenv("HDA_CODED") && atoi(getenv("HDA_CODED")) == 0);
   return on;
}

// ---- row-class coding -----------------------------------------------------------------------------------
// One step further for operators that are already stencil-coded: away from the boundary every row of a
// constant-coefficient discretisation spells the SAME sequence of entry codes.  Rows with equal sequences (<= 8
// entries, no escapes) share a class; the product then reads ONE byte per row instead of a row pointer and 7 entry
// codes (11 -> 1 B per row of the 7-pt operator) and takes offsets and values of the whole row from a class table in
// LDS.  Class 255 = the row is read from the CSR arrays (ghost columns of a row block, rows longer than 8).
// Products and the order of additions are those of the CSR row: bit-identical results.
constexpr int                kRcSlots  = 127;
constexpr unsigned long long kRcNoKey  = 0xffffffffffffffffull; // also the key of an empty row: never a class

__device__ __forceinline__ unsigned long long rc_row_key(int s, int e, const unsigned char *__restrict__ code)
{
   if (e - s > 8 || e == s) return kRcNoKey;
   unsigned long long key = 0;
   for (int u = 0; u < 8; u++)
   {
      const unsigned c = (s + u < e) ? code[s + u] : 254u; // 254 pads short rows ...
      if (c == 255u || (c == 254u && s + u < e)) return kRcNoKey; // ... so a row that really uses dictionary slot 254 stays a CSR row
      key |= (unsigned long long)c << (8 * u);
   }
   return key;
}
__global__ __launch_bounds__(256) void k_rc_collect(int n, int stride, const int *__restrict__ rp, const unsigned char *__restrict__ code,
                                                    unsigned long long *keys, int *fails)
{
   const long i = ((long)blockIdx.x * 256 + threadIdx.x) * stride;
   if (i >= n || *(volatile int *)fails > 0) return;
   const unsigned long long key = rc_row_key(rp[i], rp[i + 1], code);
   if (key == kRcNoKey) return;
   int s = (int)(key % kRcSlots);
   for (int probe = 0; probe < kRcSlots; probe++)
   {
      unsigned long long cur = keys[s];
      if (cur == kRcNoKey) cur = atomicCAS(&keys[s], kRcNoKey, key);
      if (cur == kRcNoKey || cur == key) return;
      s = (s + 1 == kRcSlots) ? 0 : s + 1;
   }
   atomicAdd(fails, 1); // more than 127 different rows: not a stencil operator in this sense
}
__global__ __launch_bounds__(256) void k_rc_encode(int n, const int *__restrict__ rp, const unsigned char *__restrict__ code,
                                                   const unsigned long long *__restrict__ keys, unsigned char *__restrict__ rclass,
                                                   unsigned long long *esc /* [0] rows, [1] entries */)
{
   __shared__ unsigned long long sk[128];
   if (threadIdx.x < 128) sk[threadIdx.x] = threadIdx.x < kRcSlots ? keys[threadIdx.x] : kRcNoKey;
   __syncthreads();
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int                s = rp[i], e = rp[i + 1];
   const unsigned long long key = rc_row_key(s, e, code);
   int                      c = 255;
   if (key != kRcNoKey)
   {
      int q = (int)(key % kRcSlots);
      for (int probe = 0; probe < kRcSlots; probe++)
      {
         if (sk[q] == key) { c = q; break; }
         if (sk[q] == kRcNoKey) break;
         q = (q + 1 == kRcSlots) ? 0 : q + 1;
      }
   }
   rclass[i] = (unsigned char)c;
   if (c == 255 && e > s)
   {
      atomicAdd(&esc[0], 1ull);
      atomicAdd(&esc[1], (unsigned long long)(e - s));
   }
   else if (c == 255) rclass[i] = 254; // empty row: nothing to add, nothing to read
}
static void ensure_rowclass(const DCsr &A)
{
   A.rowcoded = 0;
   const bool on = !(getenv("HDA_ROWCLASS") && atoi(getenv("HDA_ROWCLASS")) == 0); // read per matrix: the parity test builds both forms
   if (!on || A.coded != 1 || A.nrows < 1) return;
   A.rc_keys.alloc(128);
   HDA_HIP(hipMemsetAsync(A.rc_keys.data(), 0xff, 128 * sizeof(unsigned long long), STREAM));
   DArray<int>                fails(1);
   DArray<unsigned long long> esc(2);
   fails.zero();
   esc.zero();
   const int g = ceil_div(A.nrows, 256);
   k_rc_collect<<<g, 256, 0, STREAM>>>(A.nrows, 1, A.rowptr.data(), A.code.data(), A.rc_keys.data(), fails.data());
   int nf = 0;
   fails.download(&nf, 1);
   if (nf > 0) { A.rc_keys.release(); return; }
   A.rclass.alloc(((size_t)A.nrows + 64 + 3) & ~(size_t)3);
   k_rc_encode<<<g, 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.code.data(), A.rc_keys.data(), A.rclass.data(), esc.data());
   unsigned long long he[2] = {0, 0};
   esc.download(he, 2);
   if (he[0] * 8 > (unsigned long long)A.nrows)
   { // more than one row in eight falls back to CSR: the entry-coded kernel serves this operator better
      A.rclass.release();
      A.rc_keys.release();
      return;
   }
   A.rowcoded       = 1;
   A.rc_esc_rows    = (int)he[0];
   A.rc_esc_entries = (long long)he[1];
   HDA_TRACE("row-class coding for %d x %d: %d rows (%.3f %%) read from CSR", A.nrows, A.ncols, A.rc_esc_rows, 100.0 * A.rc_esc_rows / A.nrows);
}

// decide once per matrix whether the coded form pays: big enough to be bandwidth-bound and at
// most 1 entry in 16 escaping
static void ensure_vcoded(const DCsr &A);
static void ensure_rowclass(const DCsr &A);
static void ensure_coded(const DCsr &A)
{
   if (A.coded >= 0) return;
   A.coded = 0;
   if (!coded_enabled() || A.nnz < (1 << 18) || A.maxrow > 64) return;
   DArray<unsigned long long> keys(256);
   DArray<int>                esc(1);
   HDA_HIP(hipMemsetAsync(keys.data(), 0xff, 256 * sizeof(unsigned long long), STREAM));
   esc.zero();
   A.dict_val.alloc(256);
   A.dict_delta.alloc(256);
   A.dict_val.zero();
   A.dict_delta.zero();
   auto reject = [&]() {
      A.code.release();
      A.dict_val.release();
      A.dict_delta.release();
      ensure_vcoded(A); // the weights may still repeat even if the (offset, value) pairs do not
   };
   // sampling pass over ~64K rows: more than 1 sampled entry in 16 outside the table ends the
   // attempt (a few are expected: ghost columns of a row block have no fixed offset)
   DArray<int> fails(1);
   fails.zero();
   const int stride = std::max(1, A.nrows / 65536), ns = ceil_div(A.nrows, stride);
   const int slimit = (int)(ns * A.avg_row() / 16.0);
   k_code_collect<<<ceil_div(ns, 256), 256, 0, STREAM>>>(A.nrows, stride, A.rowptr.data(), A.col.data(), A.val.data(), keys.data(),
                                                         A.dict_val.data(), A.dict_delta.data(), fails.data(), slimit);
   int nf = 0;
   fails.download(&nf, 1);
   if (nf > slimit) return reject();
   fails.zero();
   A.code.alloc(((size_t)A.nnz + 3 + 16) & ~(size_t)3); // row kernels read up to 3 words past an entry
   const int g = ceil_div(A.nrows, 256);
   if (stride > 1)
   {
      k_code_collect<<<g, 256, 0, STREAM>>>(A.nrows, 1, A.rowptr.data(), A.col.data(), A.val.data(), keys.data(), A.dict_val.data(),
                                            A.dict_delta.data(), fails.data(), A.nnz / 16);
      fails.download(&nf, 1);
      if ((long long)nf * 16 > A.nnz) return reject();
   }
   k_code_encode<<<g, 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.col.data(), A.val.data(), keys.data(), A.dict_val.data(), A.dict_delta.data(),
                                        A.code.data(), esc.data());
   int e = 0;
   esc.download(&e, 1);
   if ((long long)e * 16 > A.nnz) return reject();
   A.coded   = 1;
   A.escapes = e;
   HDA_TRACE("coded SpMV for %d x %d, nnz %d: %d escapes (%.3f %%)", A.nrows, A.ncols, A.nnz, e, 100.0 * e / std::max(A.nnz, 1));
   ensure_rowclass(A);
}

// Row-per-lane kernel of the coded product: no LDS staging, no barriers.  A lane fetches the
// (<= 8) code bytes of its row as three aligned words straight from HBM (neighbouring lanes
// read neighbouring bytes), decodes them against the dictionary in LDS and keeps all of the
// row's gathers in flight at once.  Longer rows take further batches of eight.  (Measured on
// 256^3: LDS-staged code chunks 220-260 us, two / four rows per lane 160 / 257 us, this 144 us.)
struct CodedBatch {
   int    j[8], cc[8];
   double v[8];
};
__device__ __forceinline__ void coded_decode(CodedBatch &B, int r, int q, int e, const unsigned int *__restrict__ cwords,
                                             const double *sv, const int *sd, const int *__restrict__ col, const double *__restrict__ val)
{
   const unsigned int *p  = cwords + (q >> 2);
   const unsigned int  w0 = p[0], w1 = p[1], w2 = p[2];
   const int           o  = q & 3;
   const unsigned int  lo = __builtin_amdgcn_alignbyte(w1, w0, o), hi = __builtin_amdgcn_alignbyte(w2, w1, o);
   bool                esc = false;
#pragma unroll
   for (int u = 0; u < 8; u++)
   {
      const int c = (int)(((u < 4 ? lo : hi) >> (8 * (u & 3))) & 255u);
      B.cc[u]     = (q + u < e) ? c : 256;
      B.j[u]      = (q + u < e) ? r + sd[c] : 0;
      B.v[u]      = sv[c];
      esc |= (B.cc[u] == 255);
   }
   if (esc)
   {
#pragma unroll
      for (int u = 0; u < 8; u++)
         if (B.cc[u] == 255) { B.j[u] = col[q + u]; B.v[u] = val[q + u]; }
   }
}
template <int MODE, bool DOT, bool SPLIT>
__global__ __launch_bounds__(256) void k_spmv_coded_row(int nrows, const int *__restrict__ rowptr, const unsigned char *__restrict__ code,
                                                        const double *__restrict__ dval, const int *__restrict__ ddelta,
                                                        const int *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ x, double alpha, double beta, const double *yin,
                                                        const double *__restrict__ b, const double *__restrict__ dinv,
                                                        const double *__restrict__ w, double *out, double *__restrict__ partial, int nown)
{
   __shared__ double sv[256];
   __shared__ int    sd[256];
   const int tid = threadIdx.x;
   sv[tid]       = dval[tid];
   sd[tid]       = ddelta[tid];
   __syncthreads();
   const unsigned int *__restrict__ cwords = (const unsigned int *)code;
   const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
   constexpr int H = 1, T = 256 * H; // rows per lane (more than one lowers occupancy and loses), rows per tile
   const int per = (((nrows + 7) >> 3) + T - 1) / T * T; // rows per XCD, whole tiles
   double    acc = 0.0;
   for (int tile = slot; tile * T < per; tile += nslot)
   {
      const int  base = xcd * per + tile * T;
      int        rr[H], s[H], e[H], q[H];
      double     e0[H], e1[H], e2[H], sum[H];
      CodedBatch B[H];
#pragma unroll
      for (int h = 0; h < H; h++)
      {
         rr[h]           = base + 256 * h + tid;
         const bool live = rr[h] < nrows;
         s[h]            = live ? rowptr[rr[h]] : 0;
         e[h]            = live ? rowptr[rr[h] + 1] : 0;
         e0[h] = e1[h] = e2[h] = 0.0;
         sum[h]          = 0.0;
         if (live)
         {
            const int r = rr[h];
            if (MODE == MODE_PLAIN) { if (beta != 0.0) e0[h] = yin[r]; if (DOT) e1[h] = w[r]; }
            else if (MODE == MODE_RESID) e0[h] = b[r];
            else { e0[h] = b[r]; e1[h] = dinv[r]; e2[h] = x[r]; }
         }
      }
      bool more = false;
#pragma unroll
      for (int h = 0; h < H; h++) { q[h] = s[h]; more |= q[h] < e[h]; }
      while (more)
      {
         double xv[H][8];
#pragma unroll
         for (int h = 0; h < H; h++) coded_decode(B[h], rr[h], q[h], e[h], cwords, sv, sd, col, val);
         if (SPLIT)
         { // ghost columns wait for k_offd_fix.  (Usually escapes, but NOT always: on slab partitions a ghost column has a constant
           // offset and gets a dictionary code -- the j >= nown test on every decoded entry is what keeps them out, keep it)
#pragma unroll
            for (int h = 0; h < H; h++)
#pragma unroll
               for (int u = 0; u < 8; u++)
                  if (B[h].j[u] >= nown) { B[h].j[u] = 0; B[h].v[u] = 0.0; }
         }
#pragma unroll
         for (int h = 0; h < H; h++)
#pragma unroll
            for (int u = 0; u < 8; u++) xv[h][u] = x[B[h].j[u]];
#pragma unroll
         for (int h = 0; h < H; h++)
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
               const double t = sum[h] + B[h].v[u] * xv[h][u];
               sum[h]         = (B[h].cc[u] != 256) ? t : sum[h];
            }
         more = false;
#pragma unroll
         for (int h = 0; h < H; h++) { q[h] = min(q[h] + 8, max(e[h], q[h])); more |= q[h] < e[h]; }
      }
#pragma unroll
      for (int h = 0; h < H; h++)
      {
         const int r = rr[h];
         if (r >= nrows) continue;
         double o0;
         if (MODE == MODE_PLAIN)
         {
            o0 = (beta == 0.0) ? alpha * sum[h] : alpha * sum[h] + beta * e0[h];
            if (DOT) acc += o0 * e1[h];
         }
         else if (MODE == MODE_RESID) o0 = e0[h] - sum[h];
         else
         {
            o0 = e2[h] + e1[h] * (e0[h] - sum[h]);
            if (DOT) acc += e0[h] * o0;
         }
         out[r] = o0;
      }
   }
   if (DOT)
   {
      acc = block_sum(acc);
      if (tid == 0)
      {
         partial[blockIdx.x] = acc;
         if (blockIdx.x + gridDim.x < kRedBlocks) partial[blockIdx.x + gridDim.x] = 0.0;
      }
   }
}

// Row-per-lane product of a row-class coded operator: the lane reads its class byte (neighbouring lanes, neighbouring
// bytes), takes the row's offsets and values from the class table in LDS (lanes of a wave mostly share a class: LDS
// broadcasts) and keeps the row's <= 8 gathers in flight; x[r + offset] of 64 consecutive rows is a coalesced load.
template <int MODE, bool DOT, bool SPLIT>
__global__ __launch_bounds__(256) void k_spmv_rowclass(int nrows, const unsigned char *__restrict__ rclass,
                                                       const unsigned long long *__restrict__ rc_keys, const double *__restrict__ dval,
                                                       const int *__restrict__ ddelta, const int *__restrict__ rowptr,
                                                       const int *__restrict__ col, const double *__restrict__ val,
                                                       const double *__restrict__ x, double alpha, double beta, const double *yin,
                                                       const double *__restrict__ b, const double *__restrict__ dinv,
                                                       const double *__restrict__ w, double *out, double *__restrict__ partial, int nown)
{
   __shared__ double sv[128 * 8];
   __shared__ int    sd[128 * 8];
   __shared__ int    sn[128];
   const int tid = threadIdx.x;
   if (tid < 128)
   {
      const unsigned long long key = tid < kRcSlots ? rc_keys[tid] : kRcNoKey;
      int                      n   = 0;
      for (int u = 0; u < 8; u++)
      {
         const unsigned c = (unsigned)(key >> (8 * u)) & 255u;
         const bool     on = key != kRcNoKey && c < 254u;
         sv[tid * 8 + u]   = on ? dval[c] : 0.0;
         sd[tid * 8 + u]   = on ? ddelta[c] : 0;
         n += on;
      }
      sn[tid] = n;
   }
   __syncthreads();
   const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
   const int per = (((nrows + 7) >> 3) + 255) / 256 * 256; // rows per XCD, whole tiles
   double    acc = 0.0;
   // One row per lane and two dependent memory round trips per row (class byte, then the gathers it names) leave the
   // kernel latency-bound (PMC: 2.5 TB/s of HBM traffic).  The class byte and the row's vector operands of the NEXT tile
   // are therefore requested before the gathers of the current one: a tile then costs one round trip.
   int    n_cls = 254;
   double n_e0 = 0.0, n_e1 = 0.0, n_e2 = 0.0;
   auto fetch = [&](int tile) {
      const int r = xcd * per + tile * 256 + tid;
      n_cls = 254;
      n_e0 = n_e1 = n_e2 = 0.0;
      if (tile * 256 < per && r < nrows)
      {
         n_cls = (int)rclass[r];
         if (MODE == MODE_PLAIN) { if (beta != 0.0) n_e0 = yin[r]; if (DOT) n_e1 = w[r]; }
         else if (MODE == MODE_RESID) n_e0 = b[r];
         else { n_e0 = b[r]; n_e1 = dinv[r]; n_e2 = x[r]; }
      }
   };
   // (runs of 2-16 consecutive tiles per workgroup, for L1 reuse of the neighbouring x lines, measured no faster)
   fetch(slot);
   for (int tile = slot; tile * 256 < per; tile += nslot)
   {
      const int    r    = xcd * per + tile * 256 + tid;
      const bool   live = r < nrows;
      const int    cls  = n_cls;
      const double e0 = n_e0, e1 = n_e1, e2 = n_e2;
      double       sum = 0.0;
      fetch(tile + nslot);
      if (cls < 128)
      {
         const int     n  = sn[cls];
         const double *cv = sv + cls * 8;
         const int    *cd = sd + cls * 8;
         double        xv[8];
#pragma unroll
         for (int u = 0; u < 8; u++)
         {
            const int j = r + cd[u];
            xv[u]       = (u < n && (!SPLIT || j < nown)) ? x[j] : 0.0;
         }
#pragma unroll
         for (int u = 0; u < 8; u++)
         {
            const double t = sum + cv[u] * xv[u];
            sum            = (u < n) ? t : sum;
         }
      }
      else if (cls == 255)
      { // CSR row (ghost columns, long rows): same products, same order
         for (int k = rowptr[r]; k < rowptr[r + 1]; k++)
         {
            const int j = col[k];
            if (!SPLIT || j < nown) sum += val[k] * x[j];
         }
      }
      if (live)
      {
         double o0;
         if (MODE == MODE_PLAIN)
         {
            o0 = (beta == 0.0) ? alpha * sum : alpha * sum + beta * e0;
            if (DOT) acc += o0 * e1;
         }
         else if (MODE == MODE_RESID) o0 = e0 - sum;
         else
         {
            o0 = e2 + e1 * (e0 - sum);
            if (DOT) acc += e0 * o0;
         }
         out[r] = o0;
      }
   }
   if (DOT)
   {
      acc = block_sum(acc);
      if (tid == 0)
      {
         partial[blockIdx.x] = acc;
         if (blockIdx.x + gridDim.x < kRedBlocks) partial[blockIdx.x + gridDim.x] = 0.0;
      }
   }
}

// ---- windowed CSR -----------------------------------------------------------------------------------
// PMC on the plain streaming kernel (profiles/r02_pmc_cache.csv): 1.19 vector-L1 accesses per entry -- the 8-byte x gather
// of every lane is an access of its own -- keep the address unit of a CU busy 74 % of the kernel; it is that unit, not HBM,
// that bounds the kernel.  The rows of a chunk (neighbouring unknowns) name the same few hundred columns again and again, so
// at plan time every chunk of ~1024 entries gets the ascending list of its DISTINCT columns and every entry the 2-byte
// position of its column in it.  The product (0) gathers each distinct x once into LDS, (1) streams (val, position) with
// coalesced loads and multiplies out of LDS, (2) reduces the rows as the plain kernel does: same products, row sums by lane
// groups (agreement with the plain kernel to rounding); 0.4 instead of 1.2 L1 accesses and 10 + 4 d instead of 12 bytes per entry (d = distinct
// columns per entry, 0.2-0.3 on Galerkin operators).  The streams and the distinct-column list of the NEXT chunk are
// requested before the row reduction of the current one.
constexpr int kWChunk = 1024, kWinSort = 2048; // kWinSort >= kWChunk + kMaxRowLds

__global__ __launch_bounds__(256) void k_wchunk_rows(int nw, int nrows, const int *__restrict__ rowptr, int *__restrict__ wmeta)
{
   const int c = blockIdx.x * 256 + threadIdx.x;
   if (c > nw) return;
   int lo = nrows;
   if (c < nw)
   {
      const int target = c * kWChunk;
      int       hi = nrows;
      lo           = 0; // smallest r with rowptr[r] >= target
      while (lo < hi)
      {
         const int mid = (lo + hi) >> 1;
         if (rowptr[mid] < target) lo = mid + 1;
         else hi = mid;
      }
   }
   wmeta[3 * c]     = lo;
   wmeta[3 * c + 1] = rowptr[lo];
}
// sorts the columns of a chunk (in registers, hda_sort.h) and leaves the distinct ones, ascending, in keys[0 .. return value)
template <int PER>
__device__ int win_sort_unique(const int *__restrict__ col, int k0, int n, int *keys, int *scan)
{
   const int tid = threadIdx.x;
   int       k[PER];
#pragma unroll
   for (int m = 0; m < PER; m++)
   {
      const int i = tid + 256 * m; // any initial placement will do: coalesced reads
      k[m]        = (i < n) ? col[k0 + i] : 0x7fffffff;
   }
   block_sort_regs<PER>(k, keys, tid); // element tid*PER + r of the sorted sequence is now k[r]
   __syncthreads();
   scan[tid] = k[PER - 1];
   __syncthreads();
   int prev = (tid > 0) ? scan[tid - 1] : -1; // columns are non-negative
   __syncthreads();
   int      cnt   = 0;
   unsigned heads = 0; // bit r: k[r] is the first of its value
#pragma unroll
   for (int r = 0; r < PER; r++)
   {
      const bool head = (tid * PER + r < n) && (k[r] != prev);
      prev            = k[r];
      heads |= (unsigned)head << r;
      cnt += head;
   }
   scan[tid] = cnt;
   __syncthreads();
   for (int o = 1; o < 256; o <<= 1)
   {
      const int v = (tid >= o) ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
   }
   const int total = scan[255];
   int       pos   = scan[tid] - cnt;
#pragma unroll
   for (int r = 0; r < PER; r++)
      if ((heads >> r) & 1u) keys[pos++] = k[r]; // the sort's exchange area is free again (barriers above)
   __syncthreads();
   return total;
}
// pass 1: sort a chunk's columns once; its distinct columns go to tcol[k0 ..) (scratch as long as the column array), their number to ucount
__global__ __launch_bounds__(256) void k_win_count(int nw, const int *__restrict__ wmeta, const int *__restrict__ col, int *__restrict__ ucount,
                                                   int *__restrict__ tcol)
{
   __shared__ int keys[kWinSort];
   __shared__ int scan[256];
   for (int c = blockIdx.x; c < nw; c += gridDim.x)
   {
      const int k0 = wmeta[3 * c + 1], n = wmeta[3 * c + 4] - k0;
      const int t  = (n <= 0) ? 0 : (n <= 1024) ? win_sort_unique<4>(col, k0, n, keys, scan) : win_sort_unique<kWinSort / 256>(col, k0, n, keys, scan);
      for (int j = threadIdx.x; j < t; j += 256) tcol[k0 + j] = keys[j];
      if (threadIdx.x == 0) ucount[c] = t;
      __syncthreads();
   }
}
// pass 2: the distinct columns move to their final place; every entry gets the position of its column (binary search in LDS)
__global__ __launch_bounds__(256) void k_win_fill(int nw, int *__restrict__ wmeta, const int *__restrict__ col, const int *__restrict__ uoff,
                                                  const int *__restrict__ tcol, int *__restrict__ ucol, unsigned short *__restrict__ lidx)
{
   __shared__ int keys[kWinSort];
   for (int c = blockIdx.x; c <= nw; c += gridDim.x)
   {
      if (threadIdx.x == 0) wmeta[3 * c + 2] = uoff[c];
      if (c == nw) break;
      const int k0 = wmeta[3 * c + 1], n = wmeta[3 * c + 4] - k0, u0 = uoff[c], t = uoff[c + 1] - u0;
      for (int j = threadIdx.x; j < t; j += 256)
      {
         const int v = tcol[k0 + j];
         keys[j]     = v;
         ucol[u0 + j] = v;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += 256)
      {
         const int cc = col[k0 + i];
         int       lo = 0, hi = t - 1;
         while (lo < hi)
         {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < cc) lo = mid + 1;
            else hi = mid;
         }
         lidx[k0 + i] = (unsigned short)lo;
      }
      __syncthreads();
   }
}
// run form: number of runs of consecutive indices among a chunk's distinct columns (tcol[k0 .. k0 + ucount))
constexpr int kWinRuns = 16;
__global__ __launch_bounds__(256) void k_win_run_count(int nw, const int *__restrict__ wmeta, const int *__restrict__ ucount, const int *__restrict__ tcol,
                                                       int *__restrict__ rcount)
{
   __shared__ int cnt;
   for (int c = blockIdx.x; c < nw; c += gridDim.x)
   {
      if (threadIdx.x == 0) cnt = 0;
      __syncthreads();
      const int k0 = wmeta[3 * c + 1], t = ucount[c];
      int       mine = 0;
      for (int j = threadIdx.x; j < t; j += 256) mine += (j == 0) || (tcol[k0 + j] != tcol[k0 + j - 1] + 1);
      if (mine) atomicAdd(&cnt, mine);
      __syncthreads();
      if (threadIdx.x == 0) rcount[c] = cnt;
      __syncthreads();
   }
}
// the run table of every chunk: (first position, first column) of its runs in ascending order, INT_MAX beyond the last
__global__ __launch_bounds__(64) void k_win_run_fill(int nw, const int *__restrict__ wmeta, const int *__restrict__ ucount, const int *__restrict__ tcol,
                                                     int *__restrict__ wrun)
{
   const int lane = threadIdx.x;
   for (int c = blockIdx.x; c < nw; c += gridDim.x)
   {
      const int k0 = wmeta[3 * c + 1], t = ucount[c];
      if (lane < 2 * kWinRuns) wrun[(size_t)c * 2 * kWinRuns + lane] = 0x7fffffff;
      int nr = 0;
      for (int base = 0; base < t; base += 64)
      {
         const int  j    = base + lane;
         const int  v    = (j < t) ? tcol[k0 + j] : 0;
         const bool head = (j < t) && (j == 0 || v != tcol[k0 + j - 1] + 1);
         const unsigned long long bal = __ballot(head);
         if (head)
         {
            const int r = nr + __popcll(bal & ((1ULL << lane) - 1));
            if (r < kWinRuns)
            {
               wrun[(size_t)c * 2 * kWinRuns + 2 * r]     = j;
               wrun[(size_t)c * 2 * kWinRuns + 2 * r + 1] = v;
            }
         }
         nr += __popcll(bal);
      }
   }
}
__global__ __launch_bounds__(256) void k_max_int(int n, const int *__restrict__ v, int *mx)
{
   int m = 0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = max(m, v[i]);
   for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
   if ((threadIdx.x & 63) == 0) atomicMax(mx, m);
}
// decide once per matrix: worth it when the chunks name clearly fewer distinct columns than entries
static void ensure_window(const DCsr &A)
{
   if (A.win >= 0) return;
   A.win = 0;
   const bool on      = !(getenv("HDA_WINDOW") && atoi(getenv("HDA_WINDOW")) == 0); // read per matrix (the parity test builds both forms)
   const long min_nnz = getenv("HDA_WINDOW_MIN_NNZ") ? atol(getenv("HDA_WINDOW_MIN_NNZ")) : (1L << 20);
   if (!on || A.coded == 1 || A.coded < 0 || A.nnz < min_nnz || A.nnz < 1 || A.maxrow > kMaxRowLds) return;
   const int nw = std::max(1, ceil_div(A.nnz, kWChunk));
   A.wmeta.alloc(3 * ((size_t)nw + 2));
   A.wmeta.zero();
   k_wchunk_rows<<<ceil_div(nw + 1, 256), 256, 0, STREAM>>>(nw, A.nrows, A.rowptr.data(), A.wmeta.data());
   DArray<int> ucount((size_t)nw + 1), uoff((size_t)nw + 1), mx(1), tcol((size_t)A.nnz + 1);
   const int   g = std::min(nw, 256 * 8);
   k_win_count<<<g, 256, 0, STREAM>>>(nw, A.wmeta.data(), A.col.data(), ucount.data(), tcol.data());
   exclusive_scan(nw, ucount.data(), uoff.data(), nullptr);
   mx.zero();
   k_max_int<<<std::min(ceil_div(nw, 256), 256), 256, 0, STREAM>>>(nw, ucount.data(), mx.data());
   int total = 0, m = 0;
   HDA_HIP(hipMemcpyAsync(&total, uoff.data() + nw, 4, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&m, mx.data(), 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   const double ratio = (double)total / std::max(A.nnz, 1);
   // 0.6 since round 3: the level-0 restriction of the benchmark (0.59 distinct columns per entry) is windowed as well -- with the row
   // operands one chunk ahead it gains 8 % (0.228 -> 0.210 ms per apply, solve 34.25 -> 33.6 ms; tools/gpurun/r03_q.sh, three rounds;
   // 0.65 and 0.8 no better: a 7-point operator in lexicographic order, 0.71, does not profit)
   const double limit = getenv("HDA_WINDOW_RATIO") ? atof(getenv("HDA_WINDOW_RATIO")) : 0.6;
   // Run form (round 3): an operator on a structured grid in lexicographic order names, per chunk, a handful of RUNS of consecutive
   // columns (a 7-point operator: the lines below, beside and above a chunk's rows -- 5 to 10 runs, 0.71 distinct columns per entry, which
   // is why the list form does not pay for it).  With at most kWinRuns runs in every chunk the list is replaced by a 128-byte run table per
   // chunk: 8 + 2 bytes per entry instead of 12, and the gather of the chunk's x values becomes a few contiguous reads.  HDA_WINDOW_RUNS=0: off.
   const bool runs_on = !(getenv("HDA_WINDOW_RUNS") && atoi(getenv("HDA_WINDOW_RUNS")) == 0);
   if (runs_on && A.coded != 2)
   {
      DArray<int> rcount((size_t)nw + 1);
      k_win_run_count<<<g, 256, 0, STREAM>>>(nw, A.wmeta.data(), ucount.data(), tcol.data(), rcount.data());
      mx.zero();
      k_max_int<<<std::min(ceil_div(nw, 256), 256), 256, 0, STREAM>>>(nw, rcount.data(), mx.data());
      int mr = 0;
      HDA_HIP(hipMemcpyAsync(&mr, mx.data(), 4, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
      if (mr <= kWinRuns)
      {
         A.ucol.alloc((size_t)nw * 2 * kWinRuns + 64);
         k_win_run_fill<<<g, 64, 0, STREAM>>>(nw, A.wmeta.data(), ucount.data(), tcol.data(), A.ucol.data());
         // positions: k_win_fill's second half (it also writes a list, into scratch here)
         DArray<int> scratch((size_t)std::max(total, 1) + 1024);
         A.lidx.alloc((size_t)std::max(A.nnz, 1) + 1024);
         k_win_fill<<<g, 256, 0, STREAM>>>(nw, A.wmeta.data(), A.col.data(), uoff.data(), tcol.data(), scratch.data(), A.lidx.data());
         A.win       = 1;
         A.win_runs  = true;
         A.nwin      = nw;
         A.win_maxu  = m;
         A.win_total = total;
         HDA_TRACE("windowed CSR (run form) for %d x %d, nnz %d: %.3f distinct columns per entry in at most %d runs per chunk", A.nrows, A.ncols, A.nnz, ratio, mr);
         return;
      }
   }
   if (ratio > limit)
   {
      A.wmeta.release();
      HDA_TRACE("windowed CSR rejected for %d x %d, nnz %d: %.2f distinct columns per entry", A.nrows, A.ncols, A.nnz, ratio);
      return;
   }
   A.ucol.alloc((size_t)std::max(total, 1) + 1024); // the kernel's prefetch may read up to 768 entries past a chunk's list
   A.lidx.alloc((size_t)std::max(A.nnz, 1) + 1024);
   k_win_fill<<<g, 256, 0, STREAM>>>(nw, A.wmeta.data(), A.col.data(), uoff.data(), tcol.data(), A.ucol.data(), A.lidx.data());
   A.win       = 1;
   A.nwin      = nw;
   A.win_maxu  = m;
   A.win_total = total;
   HDA_TRACE("windowed CSR for %d x %d, nnz %d: %.3f distinct columns per entry, at most %d in a chunk", A.nrows, A.ncols, A.nnz, ratio, m);
}

template <int MODE, bool DOT, bool VC, bool SPLIT, bool RUNS = false>
__global__ __launch_bounds__(256) void k_spmv_win(int nw, const int *__restrict__ wmeta, const int *__restrict__ rowptr,
                                                  const unsigned short *__restrict__ lidx, const int *__restrict__ ucol,
                                                  const double *__restrict__ val, const double *__restrict__ x, double alpha, double beta,
                                                  const double *yin, const double *__restrict__ b, const double *__restrict__ dinv,
                                                  const double *__restrict__ w, double *out, double *__restrict__ partial, int nown,
                                                  int prod_len, const unsigned char *__restrict__ code, const double *__restrict__ dval, int pf,
                                                  const double *__restrict__ dinv2 = nullptr, double *__restrict__ out2 = nullptr)
{ // out2 (MODE_PLAIN only): a second result out2 = dinv2 .* out, as in k_spmv_stream (the next level's zero-guess Jacobi sweep)
   extern __shared__ double smem[];
   double *prod = smem, *xs = smem + prod_len;
   __shared__ double sdict[VC ? 256 : 1];
   if (VC)
   {
      sdict[threadIdx.x] = dval[threadIdx.x];
      __syncthreads();
   }
   auto value = [&](int k) -> double { // value-coded operator: one-byte code into the dictionary, 255 = read the value array
      if (VC) { const int q = code[k]; return (q != 255) ? sdict[q] : val[k]; }
      return val[k];
   };
   constexpr int E = 4, U = 3; // entries / distinct columns per thread requested one chunk ahead
   const int tid = threadIdx.x;
   const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
   const int nper = (nw + 7) >> 3;
   double    acc  = 0.0;
   struct Chunk { int r0, r1, k0, k1, u0, nu, q; bool ok; };
   auto meta = [&](int i) {
      Chunk     c;
      const int q = xcd * nper + i;
      c.q         = q;
      c.ok        = i < nper && q < nw;
      if (c.ok)
      {
         c.r0 = wmeta[3 * q]; c.k0 = wmeta[3 * q + 1]; c.u0 = wmeta[3 * q + 2];
         c.r1 = wmeta[3 * q + 3]; c.k1 = wmeta[3 * q + 4]; c.nu = wmeta[3 * q + 5] - c.u0;
      }
      else c.r0 = c.r1 = c.k0 = c.k1 = c.u0 = c.nu = 0;
      return c;
   };
   double pv[E];
   int    pl[E], pu[U];
   // per-row operands of the lane's FIRST row of a chunk, requested with the chunk's streams (one chunk ahead): the row reduction
   // then starts on LDS products at once instead of on a round trip for its row pointers, and ends on registers instead of on a
   // round trip for b, dinv and x[r]  (pf: HDA_WIN_PF, A/B)
   int    ps = 0, pe = 0;
   double pb = 0.0, pd = 0.0, px = 0.0;
   auto lanes_per_row = [](int nr) {
      int L = 1;
      while (L < 64 && nr * (L << 1) <= 256) L <<= 1;
      return L;
   };
   auto request_rows = [&](const Chunk &c) {
      const int nr = c.r1 - c.r0, L = lanes_per_row(nr), rr = tid / L;
      if (rr < nr)
      {
         const int r = c.r0 + rr;
         ps          = rowptr[r] - c.k0;
         pe          = rowptr[r + 1] - c.k0;
         if (MODE == MODE_PLAIN) { if (beta != 0.0) pb = yin[r]; if (DOT) pd = w[r]; }
         else if (MODE == MODE_RESID) pb = b[r];
         else { pb = b[r]; pd = dinv[r]; px = x[r]; }
      }
   };
   auto request = [&](const Chunk &c) {
      if (pf) request_rows(c);
#pragma unroll
      for (int e = 0; e < E; e++)
      {
         const int k = c.k0 + tid + 256 * e;
         pv[e]       = (k < c.k1) ? value(k) : 0.0;
         pl[e]       = (k < c.k1) ? (int)lidx[k] : 0;
      }
      if (RUNS) pu[0] = ((tid & 63) < 2 * kWinRuns) ? ucol[(size_t)c.q * 2 * kWinRuns + (tid & 63)] : 0x7fffffff; // the chunk's run table, once per wave
      else
      {
#pragma unroll
         for (int u = 0; u < U; u++)
         {
            const int j = tid + 256 * u;
            pu[u]       = (j < c.nu) ? ucol[c.u0 + j] : 0;
         }
      }
   };
   // run form: column of the distinct-column position j = first column of its run + distance from the run's first position; the table
   // (lanes 2r, 2r + 1 of every wave: first position and first column of run r, ascending, INT_MAX beyond the last) is searched by shuffles
   auto run_col = [&](int j) {
      int lo = 0;
#pragma unroll
      for (int step = kWinRuns / 2; step > 0; step >>= 1)
      {
         const int p = __shfl(pu[0], 2 * (lo + step));
         if (p <= j) lo += step;
      }
      return __shfl(pu[0], 2 * lo + 1) + (j - __shfl(pu[0], 2 * lo));
   };
   int   i   = slot;
   Chunk cur = meta(i);
   if (cur.ok) request(cur);
   int    cps = ps, cpe = pe; // row operands of the chunk being reduced (the request for the next one overwrites ps .. px)
   double cpb = pb, cpd = pd, cpx = px;
   while (cur.ok)
   {
      const int ne = cur.k1 - cur.k0;
      // stage 0: every distinct x of the chunk once, ascending addresses
      if (RUNS)
      {
         for (int base = 0; base < cur.nu; base += 256) // (uniform trip count: the shuffles of run_col need whole waves)
         {
            const int j  = base + tid;
            const int uc = run_col(j);
            if (j < cur.nu) xs[j] = (SPLIT && uc >= nown) ? 0.0 : x[uc];
         }
      }
      else
      {
#pragma unroll
         for (int u = 0; u < U; u++)
         {
            const int j = tid + 256 * u;
            if (j < cur.nu) xs[j] = (SPLIT && pu[u] >= nown) ? 0.0 : x[pu[u]];
         }
         for (int j = tid + 256 * U; j < cur.nu; j += 256)
         {
            const int uc = ucol[cur.u0 + j];
            xs[j]        = (SPLIT && uc >= nown) ? 0.0 : x[uc];
         }
      }
      __syncthreads();
      // stage 1: products -> LDS
#pragma unroll
      for (int e = 0; e < E; e++)
      {
         const int k = tid + 256 * e;
         if (k < ne) prod[k] = pv[e] * xs[pl[e]];
      }
      for (int k = tid + 256 * E; k < ne; k += 256) prod[k] = value(cur.k0 + k) * xs[lidx[cur.k0 + k]];
      // the next chunk's streams and distinct-column list leave now and travel during the reduction
      const Chunk nxt = meta(i + nslot);
      if (nxt.ok) request(nxt);
      __syncthreads();
      // stage 2: L lanes per row, L = largest power of two with rows*L <= 256
      // (the row operands prefetched for `cur` were overwritten by the request for `nxt` above: keep them)
      const int    cs = cps, ce = cpe;
      const double cb = cpb, cd = cpd, cx = cpx;
      cps = ps; cpe = pe; cpb = pb; cpd = pd; cpx = px;
      const int nr = cur.r1 - cur.r0;
      const int L  = lanes_per_row(nr);
      const int lane = tid & (L - 1);
      bool first = pf;
      for (int rr = tid / L; rr < nr; rr += 256 / L)
      {
         const int r = cur.r0 + rr;
         int       s, e;
         if (first) { s = cs; e = ce; }
         else { s = rowptr[r] - cur.k0; e = rowptr[r + 1] - cur.k0; }
         double    sum = 0.0;
         for (int q = s + lane; q < e; q += L) sum += prod[q];
         for (int o = L >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
         if (lane == 0)
         {
            double o0;
            if (MODE == MODE_PLAIN)
            {
               o0 = (beta == 0.0) ? alpha * sum : alpha * sum + beta * (first ? cb : yin[r]);
               if (DOT) acc += o0 * (first ? cd : w[r]);
               if (out2) out2[r] = dinv2[r] * o0;
            }
            else if (MODE == MODE_RESID) o0 = (first ? cb : b[r]) - sum;
            else
            {
               const double br = first ? cb : b[r];
               o0              = (first ? cx : x[r]) + (first ? cd : dinv[r]) * (br - sum);
               if (DOT) acc += br * o0;
            }
            out[r] = o0;
         }
         first = false;
      }
      __syncthreads();
      cur = nxt;
      i += nslot;
   }
   if (DOT)
   {
      acc = block_sum(acc);
      if (tid == 0)
      {
         partial[blockIdx.x] = acc;
         if (blockIdx.x + gridDim.x < kRedBlocks) partial[blockIdx.x + gridDim.x] = 0.0;
      }
   }
}

// ---- ghost-column part of a row-partitioned product ------------------------------------------------
// The entries with ghost columns of the rows that have any ("boundary rows"), as a compressed-row list built
// once per operator.  k_offd_fix runs after the halo exchange has landed and adds them to what the SPLIT
// main kernel left: out_r (+)= the mode's linear function of sum_r = sum_k val_k * x_ghost[col_k]; fused dots
// are linear in out, so their block partials get the matching correction (same slot, entries 0..grid-1).
__global__ __launch_bounds__(256) void k_offd_count(int n, int nown, const int *__restrict__ rp, const int *__restrict__ cj, int *__restrict__ cnt,
                                                    int *__restrict__ flag)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int c = 0;
   for (int k = rp[i]; k < rp[i + 1]; k++) c += (cj[k] >= nown);
   cnt[i]  = c;
   flag[i] = c > 0;
}
__global__ __launch_bounds__(256) void k_offd_fill(int n, int nown, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                   const int *__restrict__ pos, const int *__restrict__ off, int *__restrict__ brow,
                                                   int *__restrict__ orp, int *__restrict__ ocol, double *__restrict__ oval)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n || pos[i + 1] == pos[i]) return;
   const int b = pos[i];
   int       o = off[i];
   brow[b]     = i;
   orp[b]      = o;
   for (int k = rp[i]; k < rp[i + 1]; k++)
      if (cj[k] >= nown) { ocol[o] = cj[k]; oval[o] = v[k]; o++; }
}
template <int MODE, bool DOT>
__global__ __launch_bounds__(256) void k_offd_fix(int nb, const int *__restrict__ brow, const int *__restrict__ rp, const int *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ x, double alpha,
                                                  const double *__restrict__ b, const double *__restrict__ dinv, const double *__restrict__ w,
                                                  double *out, double *__restrict__ partial)
{
   double acc = 0.0;
   for (int q = blockIdx.x * 256 + threadIdx.x; q < nb; q += gridDim.x * 256)
   {
      const int r   = brow[q];
      double    sum = 0.0;
      for (int k = rp[q]; k < rp[q + 1]; k++) sum += val[k] * x[col[k]];
      if (MODE == MODE_PLAIN)
      {
         const double d = alpha * sum;
         out[r] += d;
         if (DOT) acc += d * w[r];
      }
      else if (MODE == MODE_RESID) out[r] -= sum;
      else
      {
         const double d = dinv[r] * sum;
         out[r] -= d;
         if (DOT) acc -= b[r] * d;
      }
   }
   if (DOT)
   {
      acc = block_sum(acc);
      if (threadIdx.x == 0) partial[blockIdx.x] += acc;
   }
}

static void ensure_offd(const DCsr &A, int nown)
{
   if (A.offd && A.offd->nown == nown) return;
   auto        P = std::make_unique<OffdPart>();
   const int   n = A.nrows;
   DArray<int> cnt((size_t)n + 1), flag((size_t)n + 1), pos((size_t)n + 1), off((size_t)n + 1);
   P->nown = nown;
   if (n)
   {
      k_offd_count<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, nown, A.rowptr.data(), A.col.data(), cnt.data(), flag.data());
      exclusive_scan(n, flag.data(), pos.data(), nullptr);
      exclusive_scan(n, cnt.data(), off.data(), nullptr);
      int tot[2] = {0, 0};
      HDA_HIP(hipMemcpyAsync(&tot[0], pos.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
      HDA_HIP(hipMemcpyAsync(&tot[1], off.data() + n, 4, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
      P->nbrows = tot[0];
      P->nnz    = tot[1];
      P->brow.alloc((size_t)std::max(P->nbrows, 1));
      P->rp.alloc((size_t)P->nbrows + 1);
      P->col.alloc((size_t)std::max(P->nnz, 1));
      P->val.alloc((size_t)std::max(P->nnz, 1));
      if (P->nbrows)
         k_offd_fill<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, nown, A.rowptr.data(), A.col.data(), A.val.data(), pos.data(), off.data(),
                                                           P->brow.data(), P->rp.data(), P->col.data(), P->val.data());
      HDA_HIP(hipMemcpyAsync(P->rp.data() + P->nbrows, &tot[1], 4, hipMemcpyHostToDevice, STREAM));
      Context::get().sync();
   }
   else P->rp.alloc(1), P->rp.zero();
   A.offd = std::move(P);
}

// ---- value-coded SpMV ----------------------------------------------------------------------
// Interpolation operators of a constant-coefficient problem repeat their WEIGHTS even though the
// column offsets are irregular (level-0 P of the 7-pt benchmark: 1300 distinct values, the 255
// most frequent cover 91 % of the entries).  Such an operator keeps its column array and replaces
// the 8-byte value by a one-byte code: 12 B -> 5 B per entry; rarer values carry code 255 and are
// read from the value array.  Same products, same order.  The codes are consumed by the LDS-staged
// streaming kernel (a row-per-lane kernel like the stencil one was 1.5x SLOWER here: irregular rows).
constexpr int kVHashSlots = 8192;

__device__ __forceinline__ unsigned vhash(unsigned long long key)
{
   key ^= key >> 31;
   key *= 0x9e3779b97f4a7c15ull;
   key ^= key >> 29;
   return (unsigned)key & (kVHashSlots - 1);
}
// histogram of a 1-in-`stride` sample of the values.  Every workgroup counts its samples in a table of its own in LDS and adds the
// table to the one in memory at the end: one atomic per (workgroup, value) instead of one per sample -- an operator whose values
// repeat sends its 2.2 M samples (256^3 series-B level 1) to a few hundred addresses, and atomics on one address are served one
// after the other at the L2 (5.6 ms per launch in the round-5 trace)
__global__ __launch_bounds__(256) void k_vhist(long nnz, int stride, const double *__restrict__ v, unsigned long long *keys, int *counts, int *distinct)
{
   extern __shared__ unsigned long long lkeys[]; // kVHashSlots keys, then as many counts
   int                                *lcounts = (int *)(lkeys + kVHashSlots);
   __shared__ int                      ldistinct, crowded;
   for (int s = threadIdx.x; s < kVHashSlots; s += 256) { lkeys[s] = kEmptyKey; lcounts[s] = 0; }
   if (threadIdx.x == 0) ldistinct = crowded = 0;
   __syncthreads();
   for (long k = ((long)blockIdx.x * 256 + threadIdx.x) * stride; k < nnz; k += (long)gridDim.x * 256 * stride)
   {
      if (*(volatile int *)&ldistinct > kVHashSlots / 2 || *(volatile int *)&crowded) break; // too many different values: give up early
      const unsigned long long key = (unsigned long long)__double_as_longlong(v[k]);
      unsigned                 s   = vhash(key);
      bool placed = false;
      for (int probe = 0; probe < 64; probe++)
      {
         unsigned long long cur = lkeys[s];
         if (cur == kEmptyKey)
         {
            cur = atomicCAS(&lkeys[s], (unsigned long long)kEmptyKey, key);
            if (cur == kEmptyKey) { atomicAdd(&ldistinct, 1); cur = key; }
         }
         if (cur == key) { atomicAdd(&lcounts[s], 1); placed = true; break; }
         s = (s + 1) & (kVHashSlots - 1);
      }
      if (!placed) { crowded = 1; break; } // 64 occupied slots in a row: far more different values than a dictionary holds
   }
   __syncthreads();
   if (ldistinct > kVHashSlots / 2 || crowded)
   {
      if (threadIdx.x == 0) atomicAdd(distinct, kVHashSlots);
      return;
   }
   for (int t = threadIdx.x; t < kVHashSlots; t += 256)
   {
      const unsigned long long key = lkeys[t];
      if (key == kEmptyKey) continue;
      if (*(volatile int *)distinct > kVHashSlots / 2) return;
      unsigned s      = vhash(key);
      bool     placed = false;
      for (int probe = 0; probe < 64; probe++)
      {
         unsigned long long cur = keys[s];
         if (cur == kEmptyKey)
         {
            cur = atomicCAS(&keys[s], (unsigned long long)kEmptyKey, key);
            if (cur == kEmptyKey) { atomicAdd(distinct, 1); cur = key; }
         }
         if (cur == key) { atomicAdd(&counts[s], lcounts[t]); placed = true; break; }
         s = (s + 1) & (kVHashSlots - 1);
      }
      if (!placed)
      { // the table in memory is crowded
         atomicAdd(distinct, kVHashSlots);
         return;
      }
   }
}
__global__ __launch_bounds__(256) void k_vencode(long nnz, const double *__restrict__ v, const unsigned long long *__restrict__ keys,
                                                 const int *__restrict__ slot_code, unsigned char *__restrict__ code, int *escapes)
{
   int esc = 0;
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256)
   {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v[k]);
      unsigned                 s   = vhash(key);
      int                      c   = 255;
      for (int probe = 0; probe < kVHashSlots; probe++)
      {
         const unsigned long long cur = keys[s];
         if (cur == kEmptyKey) break;
         if (cur == key) { c = slot_code[s]; break; }
         s = (s + 1) & (kVHashSlots - 1);
      }
      code[k] = (unsigned char)c;
      esc += (c == 255);
   }
   for (int o = 32; o > 0; o >>= 1) esc += __shfl_xor(esc, o);
   if ((threadIdx.x & 63) == 0 && esc) atomicAdd(escapes, esc);
}

static void ensure_vcoded(const DCsr &A)
{ // called when the (offset, value) coding was rejected; leaves A.coded = 2 on success
   if (A.nnz < (1 << 20) || A.maxrow > 64) return;
   DArray<unsigned long long> keys((size_t)kVHashSlots);
   DArray<int>                counts((size_t)kVHashSlots), distinct(1), esc(1);
   HDA_HIP(hipMemsetAsync(keys.data(), 0xff, sizeof(unsigned long long) * kVHashSlots, STREAM));
   counts.zero();
   distinct.zero();
   esc.zero();
   // a small strided probe rejects operators with irregular values (every Galerkin coarse operator)
   // before the larger sample that ranks the values by frequency
   int nd = 0;
   constexpr size_t vhist_lds = (size_t)kVHashSlots * (sizeof(unsigned long long) + sizeof(int));
   HDA_HIP(hipFuncSetAttribute((const void *)k_vhist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vhist_lds));
   k_vhist<<<64, 256, vhist_lds, STREAM>>>(A.nnz, std::max(1, A.nnz / 65536), A.val.data(), keys.data(), counts.data(), distinct.data());
   distinct.download(&nd, 1);
   if (nd > 2048) return;
   HDA_HIP(hipMemsetAsync(keys.data(), 0xff, sizeof(unsigned long long) * kVHashSlots, STREAM));
   counts.zero();
   distinct.zero();
   k_vhist<<<256, 256, vhist_lds, STREAM>>>(A.nnz, 64, A.val.data(), keys.data(), counts.data(), distinct.data());
   distinct.download(&nd, 1);
   if (nd > kVHashSlots / 2) return;
   std::vector<unsigned long long> hk = keys.to_host();
   std::vector<int>                hc = counts.to_host();
   std::vector<int>                order;
   long                            total = 0;
   for (int s = 0; s < kVHashSlots; s++)
      if (hk[(size_t)s] != kEmptyKey) { order.push_back(s); total += hc[(size_t)s]; }
   std::sort(order.begin(), order.end(), [&](int a, int b) { return hc[(size_t)a] != hc[(size_t)b] ? hc[(size_t)a] > hc[(size_t)b] : hk[(size_t)a] < hk[(size_t)b]; });
   long                covered = 0;
   std::vector<int>    slot_code((size_t)kVHashSlots, 255);
   std::vector<double> dict(256, 0.0);
   for (size_t q = 0; q < order.size() && q < 255; q++)
   {
      slot_code[(size_t)order[q]] = (int)q;
      long long bits              = (long long)hk[(size_t)order[q]];
      memcpy(&dict[q], &bits, 8);
      covered += hc[(size_t)order[q]];
   }
   if (total == 0 || covered * 10 < total * 7) return; // less than 70 % of the sample covered: not worth it
   DArray<int> dsc;
   dsc.upload(slot_code.data(), slot_code.size());
   A.dict_val.upload(dict.data(), dict.size());
   A.code.alloc(((size_t)A.nnz + 3 + 16) & ~(size_t)3);
   k_vencode<<<2048, 256, 0, STREAM>>>(A.nnz, A.val.data(), keys.data(), dsc.data(), A.code.data(), esc.data());
   int e = 0;
   esc.download(&e, 1);
   if ((long long)e * 10 > (long long)A.nnz * 3)
   {
      A.code.release();
      A.dict_val.release();
      return;
   }
   A.coded   = 2;
   A.escapes = e;
   HDA_TRACE("value-coded SpMV for %d x %d, nnz %d: %d distinct values sampled, %d escapes (%.2f %%)", A.nrows, A.ncols, A.nnz, nd, e,
             100.0 * e / std::max(A.nnz, 1));
}

static int spmv_mode()
{
   static int m = -1;
   if (m < 0)
   {
      m = 0; // (1 forced the lane-group kernel everywhere: rounds 1-2)
   }
   return m;
}

// row operands of the windowed kernel requested one chunk ahead (HDA_WIN_PF=0: loaded in the row reduction).  Same-box A/B, three
// rounds (tools/gpurun/r03_f.sh): level-1 Jacobi sweep at 256^3 0.4562 -> 0.4512 ms (-1.1 %), solve 34.53 -> 34.40 ms.
static int win_pf()
{
   constexpr int v = 1;
   return v;
}
// entries below which a product runs on the lane-group kernel (HDA_SMALL_NNZ; 0 = never).  Same-box A/B, two rounds each
// (tools/gpurun/r03_e.sh): 1 000 000 entries: 64^3 2.12 -> 1.96 ms per solve, 128^3 5.50 -> 5.25, 256^3 34.66 -> 34.33;
// 100 000 and 4 000 000 both a little behind it.
static long small_nnz()
{
   constexpr long v = 1000000;
   return v;
}

// grid of a product that shares the chip with its own halo transfer: one workgroup slot per CU is left free
// (7 of 8 resident 256-thread workgroups), or the transfer kernel would only start when the product ends
static int overlap_grid()
{
   static const int g = [] {
      int         v = 1792;
      v             = std::max(kRedBlocks / 2, std::min(kRedBlocks, v));
      return v / 8 * 8;
   }();
   return g;
}

// nown < 0: the whole product.  nown >= 0: SPLIT -- entries with ghost columns (>= nown) are left to k_offd_fix
// y = A x with a second result y2 = dinv2 .* y from the same kernel when the operator runs on the streaming kernel (else the
// caller does the scaling itself): set by spmv_with_scaled_copy for the duration of one spmv() call
struct SpmvEpilogue {
   const double *dinv2 = nullptr;
   double       *out2  = nullptr;
   bool          done  = false;
};
#define g_epilogue (RankState<SpmvEpilogue>::get())

template <int MODE, bool DOT>
static bool launch_spmv_impl(const DCsr &A, const double *x, double alpha, double beta,
                        const double *yin, const double *b, const double *dinv, const double *w,
                        double *out, double *partial, int nown = -1)
{
   if (A.nrows == 0 && !DOT) return true;
   ensure_plan(A);
   ensure_coded(A);
   const bool split = nown >= 0;
   const int  gmax  = split ? overlap_grid() : kRedBlocks;
   if (A.coded == 1 && spmv_mode() == 0)
   {
      const int per  = (((A.nrows + 7) >> 3) + 255) / 256 * 256; // rows per XCD (as in the kernel)
      const int grid = DOT ? gmax : std::min(gmax, 8 * (per / 256));
      if (A.rowcoded == 1)
      {
#define HDA_RC(SPF)                                                                                                                              \
   k_spmv_rowclass<MODE, DOT, SPF><<<grid, 256, 0, STREAM>>>(A.nrows, A.rclass.data(), A.rc_keys.data(), A.dict_val.data(), A.dict_delta.data(),   \
                                                             A.rowptr.data(), A.col.data(), A.val.data(), x, alpha, beta, yin, b, dinv, w, out,    \
                                                             partial, nown)
         if (split) HDA_RC(true);
         else HDA_RC(false);
#undef HDA_RC
         return true;
      }
      if (split)
         k_spmv_coded_row<MODE, DOT, true><<<grid, 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.code.data(), A.dict_val.data(), A.dict_delta.data(),
                                                                     A.col.data(), A.val.data(), x, alpha, beta, yin, b, dinv, w, out, partial, nown);
      else
         k_spmv_coded_row<MODE, DOT, false><<<grid, 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.code.data(), A.dict_val.data(), A.dict_delta.data(),
                                                                      A.col.data(), A.val.data(), x, alpha, beta, yin, b, dinv, w, out, partial, 0);
      return true;
   }
   // Small operators (the coarse levels of a hierarchy; every level of a small problem) are latency-bound, not bandwidth-bound: what
   // a product costs there is its chain of dependent memory round trips.  The lane-group kernel has three (row pointer -> entries ->
   // x); the chunked LDS kernels five and two barriers (chunk table -> row pointers -> entries -> x -> products in LDS -> row
   // pointers again).  Below small_nnz() entries the lane-group kernel runs (whole products only: it has no owned-column form).
   const bool small = !split && A.coded != 1 && A.nnz <= small_nnz() && A.nnz > 0;
   if (spmv_mode() == 0 && A.maxrow <= kMaxRowLds && !small)
   {
      ensure_window(A);
      if (A.win == 1)
      {
         const int    plen = kWChunk + A.maxrow;
         const size_t wlds = sizeof(double) * (size_t)(plen + A.win_maxu);
         const int    wg   = DOT ? gmax : std::min(gmax, ((A.nwin + 7) / 8) * 8);
         // a scaled second result asked for by spmv_with_scaled_copy (whole products only, not the owned-column half)
         const double *wepi_d = nullptr;
         double       *wepi_o = nullptr;
         if (MODE == MODE_PLAIN && !DOT && !split && g_epilogue.out2)
         {
            wepi_d = g_epilogue.dinv2;
            wepi_o = g_epilogue.out2;
            g_epilogue.done = true;
         }
#define HDA_WIN(VCF, SPF, CODE, DICT)                                                                                                   \
   k_spmv_win<MODE, DOT, VCF, SPF><<<wg, 256, wlds, STREAM>>>(A.nwin, A.wmeta.data(), A.rowptr.data(), A.lidx.data(), A.ucol.data(),     \
                                                              A.val.data(), x, alpha, beta, yin, b, dinv, w, out, partial, nown, plen,  \
                                                              CODE, DICT, win_pf(), wepi_d, wepi_o)
         if (A.win_runs)
         { // run form (never value-coded)
            if (split) k_spmv_win<MODE, DOT, false, true, true><<<wg, 256, wlds, STREAM>>>(A.nwin, A.wmeta.data(), A.rowptr.data(), A.lidx.data(), A.ucol.data(), A.val.data(), x, alpha, beta, yin, b, dinv, w, out, partial, nown, plen, nullptr, nullptr, win_pf());
            else k_spmv_win<MODE, DOT, false, false, true><<<wg, 256, wlds, STREAM>>>(A.nwin, A.wmeta.data(), A.rowptr.data(), A.lidx.data(), A.ucol.data(), A.val.data(), x, alpha, beta, yin, b, dinv, w, out, partial, nown, plen, nullptr, nullptr, win_pf(), wepi_d, wepi_o);
         }
         else if (A.coded == 2)
         {
            if (split) { HDA_WIN(true, true, A.code.data(), A.dict_val.data()); }
            else { HDA_WIN(true, false, A.code.data(), A.dict_val.data()); }
         }
         else
         {
            if (split) { HDA_WIN(false, true, nullptr, nullptr); }
            else { HDA_WIN(false, false, nullptr, nullptr); }
         }
#undef HDA_WIN
         return true;
      }
      const int    grid = DOT ? gmax : std::min(gmax, ((A.nchunks + 7) / 8) * 8);
      const size_t lds  = sizeof(double) * (size_t)(kChunk + A.maxrow);
      // a scaled second result asked for by spmv_with_scaled_copy: taken here (whole products only, not the owned-column half)
      const double *epi_d = nullptr;
      double       *epi_o = nullptr;
      if (MODE == MODE_PLAIN && !DOT && !split && g_epilogue.out2)
      {
         epi_d = g_epilogue.dinv2;
         epi_o = g_epilogue.out2;
         g_epilogue.done = true;
      }
#define HDA_STREAM(VCF, SPF, CODE, DICT)                                                                                                  \
   k_spmv_stream<MODE, DOT, VCF, SPF><<<grid, 256, lds, STREAM>>>(A.nchunks, A.chunk_row.data(), A.rowptr.data(), A.col.data(), A.val.data(), x, \
                                                                  alpha, beta, yin, b, dinv, w, out, partial, CODE, DICT, nown, epi_d, epi_o)
      if (A.coded == 2)
      {
         if (split) HDA_STREAM(true, true, A.code.data(), A.dict_val.data());
         else HDA_STREAM(true, false, A.code.data(), A.dict_val.data());
      }
      else
      {
         if (split) HDA_STREAM(false, true, nullptr, nullptr);
         else HDA_STREAM(false, false, nullptr, nullptr);
      }
#undef HDA_STREAM
      return true;
   }
   if (split) return false; // the lane-group kernel (very long rows, HDA_SPMV=vector) has no split form: caller exchanges first
   const int lpr  = pick_lpr(A);
   long      need = ((long)A.nrows * lpr + 511) / 512; // two rows per group
   int       grid = DOT ? kRedBlocks : (int)std::min<long>(std::max<long>(need, 1), kRedBlocks);
   const double *vepi_d = nullptr;
   double       *vepi_o = nullptr;
   if (MODE == MODE_PLAIN && !DOT && g_epilogue.out2)
   { // the scaled second result of spmv_with_scaled_copy
      vepi_d = g_epilogue.dinv2;
      vepi_o = g_epilogue.out2;
      g_epilogue.done = true;
   }
#define HDA_LAUNCH(L)                                                                         \
   k_spmv<L, MODE, DOT><<<grid, 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.col.data(),      \
                                                  A.val.data(), x, alpha, beta, yin, b, dinv,  \
                                                  w, out, partial, vepi_d, vepi_o)
   switch (lpr)
   {
      case 4: HDA_LAUNCH(4); break;
      case 8: HDA_LAUNCH(8); break;
      case 16: HDA_LAUNCH(16); break;
      case 32: HDA_LAUNCH(32); break;
      default: HDA_LAUNCH(64); break;
   }
#undef HDA_LAUNCH
   return true;
}

// Default: products overlap their ghost refresh on transports whose exchange only ENQUEUES work (RCCL).  On the host-staged
// transport the exchange blocks the host and several ranks share one GPU: measured 17 % slower with the split (4 ranks x 128^3 on
// one MI355X: 124.1 vs 106.6 ms per solve, gpurun_out/r02c), so it stays serial there.  HDA_OVERLAP=1 / 0 forces either (tests).
static int g_overlap_forced = -2; // -2: not read yet; -1: by transport; 0 / 1: forced (HDA_OVERLAP, or set_overlap_mode below)
void set_overlap_mode(int mode) { g_overlap_forced = (mode == 0 || mode == 1) ? mode : (getenv("HDA_OVERLAP") ? (atoi(getenv("HDA_OVERLAP")) != 0 ? 1 : 0) : -1); }
static bool overlap_enabled()
{
   if (g_overlap_forced == -2) set_overlap_mode(-1);
   return g_overlap_forced >= 0 ? g_overlap_forced == 1 : Comm::world().async_exchange();
}

// Row-partitioned product with the ghost refresh of x under it (SURVEY 2.4 C1 "overlapped with the diag-block SpMV"):
//   pack the send buffer | rows' owned-column part (SPLIT kernel) || transfer on the communication stream | ghost-column part
// halo == nullptr or a one-rank run: the plain product.
template <int MODE, bool DOT>
static void launch_spmv_halo(const DCsr &A, const HaloPlan *halo, double *x, double alpha, double beta, const double *yin, const double *b,
                             const double *dinv, const double *w, double *out, double *partial)
{
   const bool active = halo && halo_active(*halo);
   if (active && overlap_enabled() && !(MODE == MODE_PLAIN && beta != 0.0 && out == x))
   {
      ensure_plan(A);
      ensure_coded(A);
      const bool splittable = spmv_mode() == 0 && (A.coded == 1 || A.maxrow <= kMaxRowLds);
      if (splittable)
      {
         ensure_offd(A, halo->nloc);
         const OffdPart &O = *A.offd;
         halo_pack(*halo, x);
         launch_spmv_impl<MODE, DOT>(A, x, alpha, beta, yin, b, dinv, w, out, partial, halo->nloc);
         halo_transfer(*halo, x); // waits for the pack only; RCCL: enqueued, runs beside the kernel above; staged: host-side while it runs
         halo_wait(*halo);
         if (O.nbrows)
         {
            const int g = std::min(ceil_div(O.nbrows, 256), DOT ? kRedBlocks / 2 : 4096);
            k_offd_fix<MODE, DOT><<<g, 256, 0, STREAM>>>(O.nbrows, O.brow.data(), O.rp.data(), O.col.data(), O.val.data(), x, alpha, b, dinv, w, out,
                                                         partial);
         }
         Comm::world().stats.overlapped++;
         return;
      }
   }
   if (active) halo_exchange(*halo, x);
   launch_spmv_impl<MODE, DOT>(A, x, alpha, beta, yin, b, dinv, w, out, partial);
}

namespace {
struct SpmvProbe {
   const DCsr             *A    = nullptr;
   int                     mode = -1;
   std::vector<hipEvent_t> evs;
};
#define g_probes (RankState<std::vector<SpmvProbe>>::get())
} // namespace
void spmv_prepare(const DCsr &A)
{
   if (A.nrows == 0) return;
   ensure_plan(A);
   ensure_coded(A);
   if (spmv_mode() == 0) ensure_window(A); // at setup, so that the "prec" timer pays for it, not the first solve
}
void spmv_probe_clear()
{
   for (SpmvProbe &p : g_probes)
      for (hipEvent_t e : p.evs) (void)hipEventDestroy(e);
   g_probes.clear();
}
int spmv_probe_add(const DCsr *A, int mode)
{
   SpmvProbe p;
   p.A    = A;
   p.mode = mode;
   g_probes.push_back(p);
   return (int)g_probes.size() - 1;
}
void spmv_probe_set(const DCsr *A, int mode)
{
   spmv_probe_clear();
   if (A) spmv_probe_add(A, mode);
}
void spmv_probe_read(int id, double *avg_ms, int *count)
{
   Context::get().sync();
   double s = 0.0;
   int    c = 0;
   if (id >= 0 && id < (int)g_probes.size())
   {
      const SpmvProbe &p = g_probes[(size_t)id];
      for (size_t e = 0; e + 1 < p.evs.size(); e += 2)
      {
         float ms = 0.f;
         if (hipEventElapsedTime(&ms, p.evs[e], p.evs[e + 1]) == hipSuccess) { s += ms; c++; }
      }
   }
   if (avg_ms) *avg_ms = c ? s / c : 0.0;
   if (count) *count = c;
}
void spmv_probe_read(double *avg_ms, int *count) { spmv_probe_read(0, avg_ms, count); }
double rowptr_stream_bytes(const DCsr &A, bool format)
{
   if (format)
   {
      ensure_plan(A);
      ensure_coded(A);
      if (A.coded == 1 && A.rowcoded == 1) return 8.0 * A.rc_esc_rows; // class rows never touch the row pointer
   }
   return 4.0 * (A.nrows + 1.0);
}
double matrix_stream_bytes(const DCsr &A, bool format)
{
   if (format)
   {
      ensure_plan(A);
      ensure_coded(A);
      if (A.coded == 1 && A.rowcoded == 1) return 1.0 * A.nrows + 12.0 * (double)A.rc_esc_entries; // one class byte per row; CSR for the rest
      if (A.coded == 1) return 1.0 * A.nnz + 12.0 * A.escapes;
      if (spmv_mode() == 0) ensure_window(A);
      const double idx = (A.win == 1) ? 2.0 * A.nnz + (A.win_runs ? 8.0 * kWinRuns * A.nwin : 4.0 * (double)A.win_total) + 12.0 * A.nwin
                                      : 4.0 * A.nnz; // 2-byte positions + distinct columns (or run tables) + chunk table, or columns
      if (A.coded == 2) return idx + 1.0 * A.nnz + 8.0 * A.escapes;
      if (A.win == 1) return idx + 8.0 * A.nnz;
   }
   return 12.0 * A.nnz;
}

template <int MODE, bool DOT>
static void launch_spmv(const DCsr &A, const HaloPlan *halo, const double *x, double alpha, double beta, const double *yin, const double *b,
                        const double *dinv, const double *w, double *out, double *partial)
{
   double *xw = const_cast<double *>(x); // with a halo plan the ghost tail of x is refreshed (callers pass writable vectors there)
   for (SpmvProbe &p : g_probes)
      if (p.A == &A && p.mode == MODE && p.evs.size() < 8192)
      {
         hipEvent_t e0, e1;
         HDA_HIP(hipEventCreate(&e0));
         HDA_HIP(hipEventCreate(&e1));
         HDA_HIP(hipEventRecord(e0, STREAM));
         launch_spmv_halo<MODE, DOT>(A, halo, xw, alpha, beta, yin, b, dinv, w, out, partial);
         HDA_HIP(hipEventRecord(e1, STREAM));
         p.evs.push_back(e0);
         p.evs.push_back(e1);
         return;
      }
   launch_spmv_halo<MODE, DOT>(A, halo, xw, alpha, beta, yin, b, dinv, w, out, partial);
}

void spmv(const DCsr &A, double alpha, const double *x, double beta, const double *y_in, double *y_out, const HaloPlan *halo)
{
   launch_spmv<MODE_PLAIN, false>(A, halo, x, alpha, beta, y_in, nullptr, nullptr, nullptr, y_out, nullptr);
}
bool spmv_with_scaled_copy(const DCsr &A, const double *x, double *y, const double *dinv2, double *y2, const HaloPlan *halo)
{
   struct Armed { // disarmed on every way out: a throwing launch must not leave a stale out2 behind for the next plain product
      Armed(const double *d, double *o) { g_epilogue = SpmvEpilogue{d, o, false}; }
      ~Armed() { g_epilogue = SpmvEpilogue{}; }
   } armed(dinv2, y2);
   launch_spmv<MODE_PLAIN, false>(A, halo, x, 1.0, 0.0, nullptr, nullptr, nullptr, nullptr, y, nullptr);
   return g_epilogue.done;
}
void spmv_dot(const DCsr &A, const double *x, double *y, const double *w, int slot, const HaloPlan *halo)
{
   launch_spmv<MODE_PLAIN, true>(A, halo, x, 1.0, 0.0, nullptr, nullptr, nullptr, w, y, Context::get().slot(slot));
}
void residual(const DCsr &A, const double *x, const double *b, double *out, const HaloPlan *halo)
{
   launch_spmv<MODE_RESID, false>(A, halo, x, 1.0, 0.0, nullptr, b, nullptr, nullptr, out, nullptr);
}
void jacobi(const DCsr &A, const double *dinv, const double *b, const double *x_in, double *x_out, int dot_slot, const HaloPlan *halo)
{
   if (dot_slot >= 0)
      launch_spmv<MODE_JACOBI, true>(A, halo, x_in, 1.0, 0.0, nullptr, b, dinv, nullptr, x_out, Context::get().slot(dot_slot));
   else
      launch_spmv<MODE_JACOBI, false>(A, halo, x_in, 1.0, 0.0, nullptr, b, dinv, nullptr, x_out, nullptr);
}

// ------------------------------------------------------------------ BLAS-1

static inline int ew_grid(long n) { return (int)std::min<long>(std::max<long>((n + 511) / 512, 1), kRedBlocks); }

__global__ __launch_bounds__(256) void k_mul(int n, const double *__restrict__ a,
                                             const double *__restrict__ b, double *__restrict__ o)
{
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) o[i] = a[i] * b[i];
}
void mul(int n, const double *a, const double *b, double *out)
{
   if (n) k_mul<<<ew_grid(n), 256, 0, STREAM>>>(n, a, b, out);
}
void jacobi_zero_guess(int n, const double *dinv, const double *b, double *x) { mul(n, dinv, b, x); }

__global__ __launch_bounds__(256) void k_dot(int n, const double *__restrict__ x,
                                             const double *__restrict__ y, double *__restrict__ partial)
{
   double acc = 0.0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += x[i] * y[i];
   acc = block_sum(acc);
   if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
void dot(int n, const double *x, const double *y, int slot)
{
   k_dot<<<kRedBlocks, 256, 0, STREAM>>>(n, x, y, Context::get().slot(slot));
}

// sum of one slot of block partials, every thread of a 256-thread block gets it: the arithmetic of k_finalize (same per-thread
// strides, same block tree), so a kernel that finishes a dot product itself gets the bits the finalize kernel would have stored
__device__ __forceinline__ double slot_sum_256(const double *__restrict__ p)
{
   double s = 0.0;
   for (int i = threadIdx.x; i < kRedBlocks; i += kRedThreads) s += p[i];
   return block_sum(s);
}

// FIN: one rank -- <s,p> is finished here from its block partials (sp_partials) instead of by a finalize launch of its own; every
// block computes the same sum, block 0 stores it for the host's breakdown test.
// Z0: the preconditioner's cycle opens with the zero-guess Jacobi sweep z0 = dinv0 .* r (Amg::first_sweep_*): written here, on
// the r this kernel has just produced, instead of by a pass of its own over r and dinv0 (z0 may alias s: same index, read first)
template <bool FIN, bool Z0>
__global__ __launch_bounds__(256) void k_cg_update(int n, double *scalars, int gamma_idx, const double *__restrict__ p,
                                                   const double *s, double *__restrict__ x, double *__restrict__ r,
                                                   double *__restrict__ partial, const double *__restrict__ sp_partials,
                                                   const double *__restrict__ dinv0, double *z0)
{
   double sp;
   if (FIN)
   {
      sp = slot_sum_256(sp_partials);
      if (blockIdx.x == 0 && threadIdx.x == 0) scalars[S_SP] = sp;
   }
   else sp = scalars[S_SP];
   const double alpha = scalars[gamma_idx] / sp;
   double       acc   = 0.0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
   {
      x[i] += alpha * p[i];
      const double ri = r[i] - alpha * s[i];
      r[i]            = ri;
      if (Z0) z0[i] = dinv0[i] * ri;
      acc += ri * ri;
   }
   acc = block_sum(acc);
   if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
void cg_update(int n, int gamma_idx, const double *p, const double *s, double *x, double *r, int rr_slot, int sp_slot, const double *dinv0,
               double *z0)
{
   Context &c = Context::get();
   const double *spp = sp_slot >= 0 ? c.slot(sp_slot) : nullptr;
   if (spp && z0) k_cg_update<true, true><<<kRedBlocks, 256, 0, c.stream>>>(n, c.scalars, gamma_idx, p, s, x, r, c.slot(rr_slot), spp, dinv0, z0);
   else if (spp) k_cg_update<true, false><<<kRedBlocks, 256, 0, c.stream>>>(n, c.scalars, gamma_idx, p, s, x, r, c.slot(rr_slot), spp, dinv0, z0);
   else if (z0) k_cg_update<false, true><<<kRedBlocks, 256, 0, c.stream>>>(n, c.scalars, gamma_idx, p, s, x, r, c.slot(rr_slot), spp, dinv0, z0);
   else k_cg_update<false, false><<<kRedBlocks, 256, 0, c.stream>>>(n, c.scalars, gamma_idx, p, s, x, r, c.slot(rr_slot), spp, dinv0, z0);
}

// FIN: one rank -- gamma_new = <r,z> (and <r,r> beside it) are finished here from block partials slots first_slot, first_slot + 1
// into scalars gn, gn + 1, instead of by a finalize launch
template <bool FIN>
__global__ __launch_bounds__(256) void k_cg_dir(int n, double *scalars, int go, int gn, const double *__restrict__ z, double *__restrict__ p,
                                                const double *__restrict__ partials2)
{
   double g;
   if (FIN)
   {
      g = slot_sum_256(partials2);
      if (blockIdx.x == 0)
      {
         const double rr = slot_sum_256(partials2 + kRedBlocks);
         if (threadIdx.x == 0) { scalars[gn] = g; scalars[gn + 1] = rr; }
      }
   }
   else g = scalars[gn];
   const double beta = g / scalars[go];
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = z[i] + beta * p[i];
}
void cg_direction(int n, int gamma_old_idx, int gamma_new_idx, const double *z, double *p, int first_slot)
{
   Context &c = Context::get();
   if (first_slot >= 0) k_cg_dir<true><<<std::max(ew_grid(n), 1), 256, 0, STREAM>>>(n, c.scalars, gamma_old_idx, gamma_new_idx, z, p, c.slot(first_slot));
   else if (n) k_cg_dir<false><<<ew_grid(n), 256, 0, STREAM>>>(n, c.scalars, gamma_old_idx, gamma_new_idx, z, p, nullptr);
}

// One step of the single-reduction (Chronopoulos-Gear) form of PCG, all vector work of an iteration in one pass:
//   beta = gamma_new / gamma_old (0 on the first step);  alpha = gamma_new / (delta - beta gamma_new / alpha_old)
//   p = u + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s;  partials of <r,r>
// scalars[t_new + {0, 2}] = gamma_new = <r,u>, delta = <w,u>; scalars[t_old] = gamma_old; scalars[alpha_idx] = the previous alpha
__global__ __launch_bounds__(256) void k_cg_single(int n, double *scalars, int t_new, int t_old, int alpha_idx, int first, const double *__restrict__ u,
                                                   const double *__restrict__ w, double *__restrict__ p, double *__restrict__ s,
                                                   double *__restrict__ x, double *__restrict__ r, double *__restrict__ partial)
{
   const double gn = scalars[t_new], delta = scalars[t_new + 2];
   double       beta = 0.0, alpha;
   if (first) alpha = gn / delta;
   else
   {
      beta  = gn / scalars[t_old];
      alpha = gn / (delta - beta * gn / scalars[alpha_idx]);
   }
   double acc = 0.0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
   {
      const double pi = u[i] + beta * p[i], si = w[i] + beta * s[i];
      p[i]            = pi;
      s[i]            = si;
      x[i] += alpha * pi;
      const double ri = r[i] - alpha * si;
      r[i]            = ri;
      acc += ri * ri;
   }
   acc = block_sum(acc);
   if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
// (alpha of the step is published by a one-thread kernel AFTER the step, so that no block of the step can read the new value)
__global__ void k_cg_single_alpha(double *scalars, int t_new, int t_old, int alpha_idx, int first)
{
   const double gn = scalars[t_new], delta = scalars[t_new + 2];
   if (first) scalars[alpha_idx] = gn / delta;
   else
   {
      const double beta   = gn / scalars[t_old];
      scalars[alpha_idx] = gn / (delta - beta * gn / scalars[alpha_idx]);
   }
}
void cg_single_step(int n, int t_new, int t_old, int alpha_idx, bool first, const double *u, const double *w, double *p, double *s, double *x,
                    double *r, int rr_slot)
{
   Context &c = Context::get();
   k_cg_single<<<kRedBlocks, 256, 0, c.stream>>>(n, c.scalars, t_new, t_old, alpha_idx, first ? 1 : 0, u, w, p, s, x, r, c.slot(rr_slot));
   k_cg_single_alpha<<<1, 1, 0, c.stream>>>(c.scalars, t_new, t_old, alpha_idx, first ? 1 : 0);
}

__global__ __launch_bounds__(256) void k_axpy(int n, double a, const double *__restrict__ x, double *__restrict__ y)
{
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}
void axpy(int n, double a, const double *x, double *y)
{
   if (n) k_axpy<<<ew_grid(n), 256, 0, STREAM>>>(n, a, x, y);
}
__global__ __launch_bounds__(256) void k_axpy_dev(int n, const double *__restrict__ scalars, int idx,
                                                  double sign, const double *__restrict__ x, double *__restrict__ y)
{
   const double a = sign * scalars[idx];
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}
void axpy_dev(int n, int scalar_idx, double sign, const double *x, double *y)
{
   if (n) k_axpy_dev<<<ew_grid(n), 256, 0, STREAM>>>(n, Context::get().scalars, scalar_idx, sign, x, y);
}
__global__ __launch_bounds__(256) void k_scale(int n, double a, double *__restrict__ x)
{
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= a;
}
void scale(int n, double a, double *x)
{
   if (n) k_scale<<<ew_grid(n), 256, 0, STREAM>>>(n, a, x);
}
__global__ __launch_bounds__(256) void k_scale_isq(int n, const double *__restrict__ scalars, int idx, double *__restrict__ x)
{
   const double t = scalars[idx];
   const double a = (t > 0.0) ? 1.0 / sqrt(t) : 0.0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= a;
}
void scale_inv_sqrt_dev(int n, int scalar_idx, double *x)
{
   if (n) k_scale_isq<<<ew_grid(n), 256, 0, STREAM>>>(n, Context::get().scalars, scalar_idx, x);
}
void copy(int n, const double *x, double *y)
{
   if (n) HDA_HIP(hipMemcpyAsync(y, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, STREAM));
}
__global__ __launch_bounds__(256) void k_fill(int n, double v, double *__restrict__ x)
{
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = v;
}
void fill(int n, double v, double *x)
{
   if (n == 0) return;
   if (v == 0.0) HDA_HIP(hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, STREAM));
   else k_fill<<<ew_grid(n), 256, 0, STREAM>>>(n, v, x);
}

// ------------------------------------------------------- dense coarse solve

// x = inv * b ; invT is the inverse stored column-major (invT[j*n+i] = inv[i][j]) so that
// consecutive threads read consecutive addresses.
__global__ __launch_bounds__(256) void k_dense_apply(int n, const double *__restrict__ invT,
                                                     const double *__restrict__ b, double *__restrict__ x)
{
   extern __shared__ double sb[];
   for (int j = threadIdx.x; j < n; j += 256) sb[j] = b[j];
   __syncthreads();
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n)
   {
      double s = 0.0;
      for (int j = 0; j < n; j++) s += invT[(size_t)j * n + i] * sb[j];
      x[i] = s;
   }
}
void dense_apply(int n, const double *invT, const double *b, double *x)
{
   if (n) k_dense_apply<<<ceil_div(n, 256), 256, sizeof(double) * (size_t)n, STREAM>>>(n, invT, b, x);
}

__global__ void k_csr_to_dense(int n, const int *__restrict__ rp, const int *__restrict__ cj,
                               const double *__restrict__ v, double *__restrict__ d)
{
   int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n)
      for (int k = rp[i]; k < rp[i + 1]; k++) d[(size_t)i * n + cj[k]] = v[k];
}
void csr_to_dense(const DCsr &A, double *dense)
{
   int n = A.nrows;
   if (!n) return;
   HDA_HIP(hipMemsetAsync(dense, 0, sizeof(double) * (size_t)n * n, STREAM));
   k_csr_to_dense<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), dense);
}

// Gauss-Jordan on [a | I] by one workgroup, no pivoting (hypre_gselim does none either);
// result written transposed (column-major inverse) for k_dense_apply.
__global__ __launch_bounds__(1024) void k_dense_invert(int n, double *a, double *inv, double *invT)
{
   const int t = threadIdx.x, nt = blockDim.x;
   for (int q = t; q < n * n; q += nt) inv[q] = ((q / n) == (q % n)) ? 1.0 : 0.0;
   __syncthreads();
   for (int k = 0; k < n; k++)
   {
      const double piv  = a[(size_t)k * n + k];
      const double ipiv = (piv != 0.0) ? 1.0 / piv : 0.0;
      __syncthreads();
      for (int j = t; j < n; j += nt)
      {
         a[(size_t)k * n + j] *= ipiv;
         inv[(size_t)k * n + j] *= ipiv;
      }
      __syncthreads();
      for (int q = t; q < n * n; q += nt)
      {
         const int i = q / n, j = q % n;
         if (i == k) continue;
         const double f = a[(size_t)i * n + k];
         if (f != 0.0 && j != k)
         {
            a[(size_t)i * n + j] -= f * a[(size_t)k * n + j];
         }
      }
      for (int q = t; q < n * n; q += nt)
      {
         const int i = q / n, j = q % n;
         if (i == k) continue;
         const double f = a[(size_t)i * n + k];
         if (f != 0.0) inv[(size_t)i * n + j] -= f * inv[(size_t)k * n + j];
      }
      __syncthreads();
      for (int i = t; i < n; i += nt)
         if (i != k) a[(size_t)i * n + k] = 0.0;
      __syncthreads();
   }
   for (int q = t; q < n * n; q += nt) invT[(size_t)(q % n) * n + (q / n)] = inv[q];
}
void dense_invert(int n, double *a, double *invT)
{
   if (!n) return;
   DArray<double> tmp((size_t)n * n);
   k_dense_invert<<<1, 1024, 0, STREAM>>>(n, a, tmp.data(), invT);
}

// -------------------------------------------------------------------- scan

template <class T>
__global__ __launch_bounds__(256) void k_scan_blocksum(long n, const int *__restrict__ in, T *__restrict__ bsum)
{
   const long base = (long)blockIdx.x * 1024;
   T          s    = 0;
   for (int q = 0; q < 4; q++)
   {
      long i = base + threadIdx.x * 4 + q;
      if (i < n) s += in[i];
   }
   // integer block sum
   __shared__ T sm[256];
   sm[threadIdx.x] = s;
   __syncthreads();
   for (int o = 128; o > 0; o >>= 1)
   {
      if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
   }
   if (threadIdx.x == 0) bsum[blockIdx.x] = sm[0];
}

template <class T>
__global__ __launch_bounds__(256) void k_scan_apply(long n, const int *__restrict__ in,
                                                    const T *__restrict__ bofs, T *__restrict__ out)
{
   const long base = (long)blockIdx.x * 1024;
   T          v[4], s = 0;
   for (int q = 0; q < 4; q++)
   {
      long i = base + threadIdx.x * 4 + q;
      v[q]   = (i < n) ? (T)in[i] : (T)0;
      s += v[q];
   }
   __shared__ T sm[256];
   sm[threadIdx.x] = s;
   __syncthreads();
   // Hillis-Steele inclusive scan over 256 thread sums
   for (int o = 1; o < 256; o <<= 1)
   {
      T add = (threadIdx.x >= o) ? sm[threadIdx.x - o] : (T)0;
      __syncthreads();
      sm[threadIdx.x] += add;
      __syncthreads();
   }
   T run = bofs[blockIdx.x] + sm[threadIdx.x] - s;
   for (int q = 0; q < 4; q++)
   {
      long i = base + threadIdx.x * 4 + q;
      if (i < n) out[i] = run;
      run += v[q];
      if (i == n - 1) out[n] = run;
   }
}

// serial scan of a short array by one thread (top of the recursion)
template <class T>
__global__ void k_scan_serial(long n, const T *__restrict__ in, T *__restrict__ out)
{
   T run = 0;
   for (long i = 0; i < n; i++)
   {
      T v    = in[i];
      out[i] = run;
      run += v;
   }
   out[n] = run;
}

template <class T>
__global__ __launch_bounds__(256) void k_scan_blocksum_T(long n, const T *__restrict__ in, T *__restrict__ bsum)
{
   const long base = (long)blockIdx.x * 1024;
   T          s    = 0;
   for (int q = 0; q < 4; q++)
   {
      long i = base + threadIdx.x * 4 + q;
      if (i < n) s += in[i];
   }
   __shared__ T sm[256];
   sm[threadIdx.x] = s;
   __syncthreads();
   for (int o = 128; o > 0; o >>= 1)
   {
      if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
   }
   if (threadIdx.x == 0) bsum[blockIdx.x] = sm[0];
}
template <class T>
__global__ __launch_bounds__(256) void k_scan_apply_T(long n, const T *__restrict__ in,
                                                      const T *__restrict__ bofs, T *__restrict__ out)
{
   const long base = (long)blockIdx.x * 1024;
   T          v[4], s = 0;
   for (int q = 0; q < 4; q++)
   {
      long i = base + threadIdx.x * 4 + q;
      v[q]   = (i < n) ? in[i] : (T)0;
      s += v[q];
   }
   __shared__ T sm[256];
   sm[threadIdx.x] = s;
   __syncthreads();
   for (int o = 1; o < 256; o <<= 1)
   {
      T add = (threadIdx.x >= o) ? sm[threadIdx.x - o] : (T)0;
      __syncthreads();
      sm[threadIdx.x] += add;
      __syncthreads();
   }
   T run = bofs[blockIdx.x] + sm[threadIdx.x] - s;
   for (int q = 0; q < 4; q++)
   {
      long i = base + threadIdx.x * 4 + q;
      if (i < n) out[i] = run;
      run += v[q];
      if (i == n - 1) out[n] = run;
   }
}

// exclusive scan of block sums (recursive); in/out type T
template <class T>
static void scan_T(long n, const T *in, T *out)
{
   if (n <= 4096)
   {
      k_scan_serial<T><<<1, 1, 0, STREAM>>>(n, in, out);
      return;
   }
   long      nb = (n + 1023) / 1024;
   DArray<T> bsum((size_t)nb), bofs((size_t)nb + 1);
   k_scan_blocksum_T<T><<<(int)nb, 256, 0, STREAM>>>(n, in, bsum.data());
   scan_T<T>(nb, bsum.data(), bofs.data());
   k_scan_apply_T<T><<<(int)nb, 256, 0, STREAM>>>(n, in, bofs.data(), out);
}

template <class T>
static void scan_from_int(long n, const int *in, T *out)
{
   if (n == 0)
   {
      HDA_HIP(hipMemsetAsync(out, 0, sizeof(T), STREAM));
      return;
   }
   long      nb = (n + 1023) / 1024;
   DArray<T> bsum((size_t)nb), bofs((size_t)nb + 1);
   k_scan_blocksum<T><<<(int)nb, 256, 0, STREAM>>>(n, in, bsum.data());
   scan_T<T>(nb, bsum.data(), bofs.data());
   k_scan_apply<T><<<(int)nb, 256, 0, STREAM>>>(n, in, bofs.data(), out);
}

void exclusive_scan(int n, const int *in, int *out, int *total_out_dev)
{
   scan_from_int<int>(n, in, out);
   if (total_out_dev)
      HDA_HIP(hipMemcpyAsync(total_out_dev, out + n, sizeof(int), hipMemcpyDeviceToDevice, STREAM));
}
void exclusive_scan64(long n, const int *in, long long *out) { scan_from_int<long long>(n, in, out); }

__global__ __launch_bounds__(256) void k_sum64(long n, const int *__restrict__ in, unsigned long long *out)
{
   unsigned long long s = 0;
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (unsigned long long)in[i];
   for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
   if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}
// matrices are indexed with int32: a row-length array whose sum does not fit is a clear error, not a wrap-around
void require_int32_total(long n, const int *counts, const char *what)
{
   DArray<unsigned long long> t(1);
   t.zero();
   if (n) k_sum64<<<(int)std::min<long>((n + 255) / 256, 1024), 256, 0, STREAM>>>(n, counts, t.data());
   unsigned long long h = 0;
   t.download(&h, 1);
   if (h >= (1ull << 31))
      throw Error(std::string(what) + " would have " + std::to_string(h) + " entries: more than int32 indexing holds (2^31-1); "
                  "partition the problem over more GPUs");
}

// ------------------------------------------------------------- row utilities

// hypre_ParCSRComputeL1Norms: option 1 = sum_j |a_ij| (sign of a_ii); option 4 = |a_ii| + half the absolute sum of the entries that
// leave the row's block, truncated to |a_ii| when within 4/3 of it -- "leaving" = ghost (off-rank) columns and, with row blocks
// (part != nullptr: nb + 1 row starts), the columns outside the row's own block; option 0 = the plain diagonal.  One thread per
// row, ascending k: bit-identical to the oracle.
__device__ __forceinline__ void l1_block_range(const int *__restrict__ part, int nb, int i, int n, int &lo, int &hi)
{
   lo = 0;
   hi = n;
   if (!part) return;
   int a = 0, b = nb; // part[a] <= i < part[b]
   while (b - a > 1)
   {
      const int m = (a + b) >> 1;
      if (part[m] <= i) a = m;
      else b = m;
   }
   lo = part[a];
   hi = part[a + 1];
}
__global__ __launch_bounds__(256) void k_l1(int n, const int *__restrict__ rp, const int *__restrict__ cj,
                                            const double *__restrict__ v, int option, double *__restrict__ l1,
                                            const int *__restrict__ part, int nb)
{
   int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   int lo, hi;
   l1_block_range(part, nb, i, n, lo, hi);
   double s = 0.0, d = 0.0, off = 0.0;
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      s += fabs(v[k]);
      if (cj[k] == i) d = v[k];
      if (cj[k] < lo || cj[k] >= hi) off += fabs(v[k]); // ghost (off-rank) and off-block columns
   }
   if (option == 1) l1[i] = (d < 0.0) ? -s : s;
   else if (option == 0) l1[i] = d;
   else
   { // option 4: a_ii + 0.5 * sum_offd |a_ij|, truncated to a_ii when within 4/3 of it
      double t = fabs(d) + 0.5 * off;
      if (t <= (4.0 / 3.0) * fabs(d)) t = fabs(d);
      l1[i] = (d < 0.0) ? -t : t;
   }
}
// G lanes per row (coalesced reads of long rows); the sums keep the sequential order: the lanes hand their terms round in
// ascending k and every lane adds them in that order
template <int G>
__global__ __launch_bounds__(256) void k_l1_grp(int n, const int *__restrict__ rp, const int *__restrict__ cj, const double *__restrict__ v,
                                                int option, double *__restrict__ l1, const int *__restrict__ part, int nb)
{
   const int  gl  = threadIdx.x & (G - 1);
   const long row = ((long)blockIdx.x * 256 + threadIdx.x) / G;
   const bool in  = row < n;
   const int  i   = in ? (int)row : 0;
   const int  k0 = in ? rp[i] : 0, k1 = in ? rp[i + 1] : 0;
   int        lo, hi;
   l1_block_range(part, nb, i, n, lo, hi);
   double     s = 0.0, d = 0.0, off = 0.0;
   for (int base = k0; base < k1; base += G)
   {
      const int k = base + gl;
      double    a = 0.0, o = 0.0;
      if (k < k1)
      {
         const int    c  = cj[k];
         const double vk = v[k];
         a = fabs(vk);
         if (c == i) d = vk;
         if (c < lo || c >= hi) o = a;
      }
      const int m = min(G, k1 - base);
      for (int l = 0; l < m; l++)
      {
         s += __shfl(a, l, G);
         off += __shfl(o, l, G); // (entries that stay inside the block add 0.0)
      }
   }
   for (int o = G >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, G); // one lane holds the diagonal, the others 0
   if (!in || gl) return;
   if (option == 1) l1[i] = (d < 0.0) ? -s : s;
   else if (option == 0) l1[i] = d;
   else
   {
      double t = fabs(d) + 0.5 * off;
      if (t <= (4.0 / 3.0) * fabs(d)) t = fabs(d);
      l1[i] = (d < 0.0) ? -t : t;
   }
}
void l1_row_norms(const DCsr &A, int option, double *l1, const int *part, int nb)
{
   if (!A.nrows) return;
   const int    n   = A.nrows;
   const double avg = A.avg_row();
   if (nb <= 1) part = nullptr;
   if (avg <= 12.0) k_l1<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), option, l1, part, nb);
   else if (avg <= 24.0) k_l1_grp<16><<<ceil_div((long long)n * 16, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), option, l1, part, nb);
   else if (avg <= 48.0) k_l1_grp<32><<<ceil_div((long long)n * 32, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), option, l1, part, nb);
   else k_l1_grp<64><<<ceil_div((long long)n * 64, 256), 256, 0, STREAM>>>(n, A.rowptr.data(), A.col.data(), A.val.data(), option, l1, part, nb);
}
// the plain diagonal (hypre's Jacobi / Gauss-Seidel types 0, 7, 3, 4, 6 divide by a_ii whatever leaves the rank)
void extract_diag(const DCsr &A, double *d) { l1_row_norms(A, 0, d); }

__global__ __launch_bounds__(256) void k_dinv(int n, const double *__restrict__ d, double w, double *__restrict__ o)
{
   int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) o[i] = (d[i] != 0.0) ? w / d[i] : 0.0;
}
void make_dinv(int n, const double *d, double weight, double *dinv)
{
   if (n) k_dinv<<<ceil_div(n, 256), 256, 0, STREAM>>>(n, d, weight, dinv);
}

__global__ __launch_bounds__(256) void k_sort_rows(int n, const int *__restrict__ rp, int *cj, double *v)
{
   int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   const int s = rp[i], e = rp[i + 1];
   for (int a = s + 1; a < e; a++)
   {
      int    c  = cj[a];
      double x  = v[a];
      int    bq = a - 1;
      while (bq >= s && cj[bq] > c)
      {
         cj[bq + 1] = cj[bq];
         v[bq + 1]  = v[bq];
         bq--;
      }
      cj[bq + 1] = c;
      v[bq + 1]  = x;
   }
}
// one wavefront per row of up to 64 entries: lane k holds entry k, bitonic network over the lanes
// (columns of a row are distinct, so the order is unique); longer rows fall back to insertion
__global__ __launch_bounds__(256) void k_sort_rows_wave(int n, const int *__restrict__ rp, int *cj, double *v)
{
   const int lane = threadIdx.x & 63;
   for (long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 6; i < n; i += ((long)gridDim.x * 256) >> 6)
   {
      const int s = rp[i], len = rp[i + 1] - s;
      if (len > 64)
      {
         if (lane == 0)
            for (int a = s + 1; a < s + len; a++)
            {
               const int    c = cj[a];
               const double x = v[a];
               int          bq = a - 1;
               while (bq >= s && cj[bq] > c) { cj[bq + 1] = cj[bq]; v[bq + 1] = v[bq]; bq--; }
               cj[bq + 1] = c;
               v[bq + 1]  = x;
            }
         continue;
      }
      if (len < 2) continue;
      int    c = (lane < len) ? cj[s + lane] : 0x7fffffff;
      double x = (lane < len) ? v[s + lane] : 0.0;
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
         for (int j = k >> 1; j > 0; j >>= 1)
         {
            const int    oc = __shfl_xor(c, j);
            const double ox = __shfl_xor(x, j);
            const bool   up = ((lane & k) == 0);          // ascending block
            const bool   lo = ((lane & j) == 0);          // this lane keeps the smaller of the pair when ascending
            const bool   take = (lo == up) ? (oc < c) : (oc > c);
            if (take) { c = oc; x = ox; }
         }
      if (lane < len) { cj[s + lane] = c; v[s + lane] = x; }
   }
}
// every row column-sorted by ONE segmented radix sort, whatever its length (columns of a row are distinct: the order is unique)
void sort_rows_segmented(DCsr &C)
{
   if (C.nrows == 0 || C.nnz == 0) return;
   DArray<int>    k2((size_t)C.nnz);
   DArray<double> v2((size_t)C.nnz);
   int            bits = 1;
   while (bits < 31 && (std::max(C.ncols, C.nrows) >> bits)) bits++;
   size_t tmp_bytes = 0;
   HDA_ROCPRIM(rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, C.col.data(), k2.data(), C.val.data(), v2.data(), (unsigned)C.nnz, (unsigned)C.nrows,
                                               C.rowptr.data(), C.rowptr.data() + 1, 0, bits, STREAM));
   DArray<char> tmp(std::max<size_t>(tmp_bytes, 1));
   HDA_ROCPRIM(rocprim::segmented_radix_sort_pairs(tmp.data(), tmp_bytes, C.col.data(), k2.data(), C.val.data(), v2.data(), (unsigned)C.nnz, (unsigned)C.nrows,
                                               C.rowptr.data(), C.rowptr.data() + 1, 0, bits, STREAM));
   C.col = std::move(k2);
   C.val = std::move(v2);
}
void sort_rows(DCsr &A)
{
   if (!A.nrows) return;
   // long rows (the operators of coarse levels: 50-100 entries and more): the wave kernel below sorts rows of more than 64 entries by
   // insertion on ONE lane -- 50 ms per transpose of a 198 k-row level with 80-entry rows in the round-5 series-B trace
   if (A.avg_row() > 40.0) return sort_rows_segmented(A);
   if (A.avg_row() > 12.0)
      k_sort_rows_wave<<<std::min(ceil_div((long long)A.nrows * 64, 256), 1 << 16), 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.col.data(),
                                                                                                 A.val.data());
   else k_sort_rows<<<ceil_div(A.nrows, 256), 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.col.data(), A.val.data());
}

__global__ __launch_bounds__(256) void k_count_cols(int nnz, const int *__restrict__ cj, int *cnt)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256) atomicAdd(&cnt[cj[k]], 1);
}
__global__ __launch_bounds__(256) void k_transpose_fill(int n, const int *__restrict__ rp, const int *__restrict__ cj,
                                                        const double *__restrict__ v, int *cursor,
                                                        int *__restrict__ tj, double *__restrict__ tv)
{
   int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   for (int k = rp[i]; k < rp[i + 1]; k++)
   {
      int q = atomicAdd(&cursor[cj[k]], 1);
      tj[q] = i;
      tv[q] = v[k];
   }
}
// T = A^T.  Atomic scatter then per-row sort => deterministic, rows ascending.
__global__ __launch_bounds__(256) void k_transpose_fill_pattern(int n, const int *__restrict__ rp, const int *__restrict__ cj, int *cursor,
                                                                int *__restrict__ tj)
{
   int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= n) return;
   for (int k = rp[i]; k < rp[i + 1]; k++) tj[atomicAdd(&cursor[cj[k]], 1)] = i;
}
// the pattern of A^T alone, rows in NO particular order (for consumers that only enumerate a column's rows: no values moved, no sort)
void transpose_pattern_unsorted(const DCsr &A, DArray<int> &trp, DArray<int> &tcj)
{
   trp.alloc((size_t)A.ncols + 1);
   tcj.alloc((size_t)std::max(A.nnz, 1));
   DArray<int> cnt((size_t)A.ncols + 1);
   cnt.zero();
   if (A.nnz) k_count_cols<<<ew_grid(A.nnz), 256, 0, STREAM>>>(A.nnz, A.col.data(), cnt.data());
   exclusive_scan(A.ncols, cnt.data(), trp.data(), nullptr);
   cnt.copy_from(trp); // cursor
   if (A.nrows) k_transpose_fill_pattern<<<ceil_div(A.nrows, 256), 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.col.data(), cnt.data(), tcj.data());
}
void transpose(const DCsr &A, DCsr &T)
{
   T.nrows = A.ncols;
   T.ncols = A.nrows;
   T.nnz   = A.nnz;
   T.rowptr.alloc((size_t)T.nrows + 1);
   T.col.alloc((size_t)std::max(T.nnz, 1));
   T.val.alloc((size_t)std::max(T.nnz, 1));
   DArray<int> cnt((size_t)T.nrows + 1);
   cnt.zero();
   if (A.nnz) k_count_cols<<<ew_grid(A.nnz), 256, 0, STREAM>>>(A.nnz, A.col.data(), cnt.data());
   exclusive_scan(T.nrows, cnt.data(), T.rowptr.data(), nullptr);
   cnt.copy_from(T.rowptr); // cursor
   if (A.nrows)
      k_transpose_fill<<<ceil_div(A.nrows, 256), 256, 0, STREAM>>>(A.nrows, A.rowptr.data(), A.col.data(), A.val.data(),
                                                                 cnt.data(), T.col.data(), T.val.data());
   sort_rows(T);
}

// ------------------------------------------------------ 7-pt generator

struct LapGeom {
   int       n[3];     // global grid
   int       ln[3];    // local block dims
   long long st[3];    // local block starts (global coords)
   long long ilower;   // first global row of this block
   int       P[3], pc[3];
   long long ps[3][9]; // partition starts per dim (P <= 8 per dim)
   double    c[3];
};

__device__ inline long long lap_gidx(const LapGeom &g, long long x, long long y, long long z)
{
   // owner block of (x,y,z), then laplacian.c:504-520 numbering
   int b[3];
   long long q[3] = {x, y, z};
   for (int d = 0; d < 3; d++)
   {
      int bb = 0;
      while (bb + 1 < g.P[d] && q[d] >= g.ps[d][bb + 1]) bb++;
      b[d] = bb;
   }
   long long lx = g.ps[0][b[0] + 1] - g.ps[0][b[0]];
   long long ly = g.ps[1][b[1] + 1] - g.ps[1][b[1]];
   return g.ps[0][b[0]] * g.n[1] * g.n[2] + g.ps[1][b[1]] * g.n[2] * lx + g.ps[2][b[2]] * lx * ly +
          ((z - g.ps[2][b[2]]) * ly + (y - g.ps[1][b[1]])) * lx + (x - g.ps[0][b[0]]);
}

__global__ __launch_bounds__(256) void k_lap7_count(LapGeom g, int local_n, int *cnt)
{
   int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= local_n) return;
   long long x = g.st[0] + i % g.ln[0], y = g.st[1] + (i / g.ln[0]) % g.ln[1], z = g.st[2] + i / (g.ln[0] * g.ln[1]);
   cnt[i] = 1 + (x > 0) + (x < g.n[0] - 1) + (y > 0) + (y < g.n[1] - 1) + (z > 0) + (z < g.n[2] - 1);
}
__global__ __launch_bounds__(256) void k_lap7_fill(LapGeom g, int local_n, const int *__restrict__ rp,
                                                   long long *__restrict__ cols, double *__restrict__ vals,
                                                   double *__restrict__ rhs)
{
   int i = blockIdx.x * 256 + threadIdx.x;
   if (i >= local_n) return;
   long long x = g.st[0] + i % g.ln[0], y = g.st[1] + (i / g.ln[0]) % g.ln[1], z = g.st[2] + i / (g.ln[0] * g.ln[1]);
   int       q = rp[i];
   cols[q]     = g.ilower + i;
   vals[q++]   = 2.0 * (g.c[0] + g.c[1] + g.c[2]);
   if (z > 0) { cols[q] = lap_gidx(g, x, y, z - 1); vals[q++] = -g.c[2]; }
   if (y > 0) { cols[q] = lap_gidx(g, x, y - 1, z); vals[q++] = -g.c[1]; }
   if (x > 0) { cols[q] = lap_gidx(g, x - 1, y, z); vals[q++] = -g.c[0]; }
   if (x < g.n[0] - 1) { cols[q] = lap_gidx(g, x + 1, y, z); vals[q++] = -g.c[0]; }
   if (y < g.n[1] - 1) { cols[q] = lap_gidx(g, x, y + 1, z); vals[q++] = -g.c[1]; }
   if (z < g.n[2] - 1) { cols[q] = lap_gidx(g, x, y, z + 1); vals[q++] = -g.c[2]; }
   rhs[i] = (y == 0) ? 1.0 : 0.0;
}

void lap7_generate(const int n[3], const int P[3], const int pc[3], const double c[3], int rowptr_out[],
                   long long cols_out[], double vals_out[], double rhs_out[], int local_n)
{
   LapGeom g;
   for (int d = 0; d < 3; d++)
   {
      HDA_REQUIRE(P[d] >= 1 && P[d] <= 8, "lap7: at most 8 blocks per dimension");
      g.n[d] = n[d]; g.P[d] = P[d]; g.pc[d] = pc[d]; g.c[d] = c[d];
      int size = n[d] / P[d], rest = n[d] - size * P[d];
      for (int j = 0; j <= P[d]; j++) g.ps[d][j] = (long long)size * j + (j < rest ? j : rest);
      g.st[d] = g.ps[d][pc[d]];
      g.ln[d] = (int)(g.ps[d][pc[d] + 1] - g.ps[d][pc[d]]);
   }
   long long lx = g.ln[0], ly = g.ln[1];
   g.ilower = g.ps[0][pc[0]] * n[1] * n[2] + g.ps[1][pc[1]] * n[2] * lx + g.ps[2][pc[2]] * lx * ly;
   HDA_REQUIRE((long long)g.ln[0] * g.ln[1] * g.ln[2] == local_n, "lap7: local size mismatch");
   DArray<int> cnt((size_t)local_n + 1);
   k_lap7_count<<<ceil_div(local_n, 256), 256, 0, STREAM>>>(g, local_n, cnt.data());
   exclusive_scan(local_n, cnt.data(), rowptr_out, nullptr);
   k_lap7_fill<<<ceil_div(local_n, 256), 256, 0, STREAM>>>(g, local_n, rowptr_out, cols_out, vals_out, rhs_out);
}

} // namespace hda
