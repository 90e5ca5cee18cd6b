// hda_common.h -- shared host-side plumbing of the MI355X (gfx950) solve path.
// Device memory comes from the stream-ordered HIP pool (hipMallocAsync) so AMG setup does
// not pay a device-wide sync per temporary; everything runs on one stream per context.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace hda {

struct Error : std::runtime_error {
   using std::runtime_error::runtime_error;
};

#define HDA_HIP(expr)                                                                    \
   do {                                                                                  \
      hipError_t _e = (expr);                                                            \
      if (_e != hipSuccess)                                                              \
      {                                                                                  \
         char _buf[512];                                                                 \
         snprintf(_buf, sizeof(_buf), "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), \
                  __FILE__, __LINE__, #expr);                                            \
         throw ::hda::Error(_buf);                                                       \
      }                                                                                  \
   } while (0)
// rocPRIM's device functions return hipGetLastError() after every launch of theirs: whatever an EARLIER launch of this thread left there
// comes back as their result -- and kernel launches of this library are not checked one by one (a launch over an empty range, grid 0,
// on a rank or level without rows leaves "invalid configuration argument" behind and does nothing else).  Round 5: the partitioned
// Galerkin product on a rank with an empty coarse level failed in segmented_radix_sort_pairs that way.  The thread's last error is read
// (which clears it) before every rocPRIM call.
#define HDA_ROCPRIM(expr)          \
   do {                            \
      (void)hipGetLastError();     \
      HDA_HIP(expr);               \
   } while (0)

#define HDA_REQUIRE(cond, msg)                                                     \
   do {                                                                            \
      if (!(cond))                                                                 \
      {                                                                            \
         char _buf[512];                                                           \
         snprintf(_buf, sizeof(_buf), "%s (%s) at %s:%d", msg, #cond, __FILE__, __LINE__); \
         throw ::hda::Error(_buf);                                                 \
      }                                                                            \
   } while (0)

// Number of per-block partial sums every fused reduction writes (fixed so that the
// summation tree -- and therefore every dot product -- is run-to-run deterministic).
constexpr int kRedBlocks  = 2048; // 8 blocks (32 waves) per CU on 256 CUs
constexpr int kRedThreads = 256;

// ---- where the library's state lives.  It is PROCESS-global: the reference's contract (include/HYPREDRV.h:66-70) is "one thread at a
// time", so initialising on one thread and solving on another is legal, and a handle freed by a finalizer thread must find its
// allocator.  The thread-rank test seam (hda_thread_ranks.hip: the ranks of a row partition as threads of one process) is the one
// place where a thread owns state: a thread that has joined a thread world gets private instances of everything below, installed by
// enter_thread_rank() and destroyed by leave_thread_rank().
bool in_thread_rank();
void enter_thread_rank();
void leave_thread_rank();                     // releases this thread's context, allocator, communicator and the rest of its private state
void thread_rank_on_leave(void (*drop)());    // (RankState registers the destruction of a private instance)
template <class T, int Tag = 0>
struct RankState {
   static T &get()
   {
      if (!in_thread_rank())
      {
         static T global;
         return global;
      }
      if (!mine_)
      {
         mine_ = new T();
         thread_rank_on_leave(&drop);
      }
      return *mine_;
   }

 private:
   static void drop()
   {
      delete mine_;
      mine_ = nullptr;
   }
   static thread_local T *mine_;
};
template <class T, int Tag>
thread_local T *RankState<T, Tag>::mine_ = nullptr;

// Host waits of the library.  With HDA_COMM_TIMEOUT_S=<seconds> (default: none, the wait blocks as hipStreamSynchronize does) a wait
// that does not end in time -- a peer rank that failed or left the collective sequence leaves RCCL / the staged transport waiting
// silently -- reports rank and stage and ends the job non-zero (MPI_Abort under MPI, _exit(86) otherwise); nothing is re-exec'd.
void        wait_stream(hipStream_t s);
void        wait_event(hipEvent_t e);
double      wait_limit_s();
void        set_stage(const char *stage); // what the library is doing, for that report ("matrix assembly", "preconditioner setup", ...)
const char *current_stage();
[[noreturn]] void abort_job(int status);  // MPI_Abort when the process runs under MPI (hda_mpi.cpp), else _exit

// One per process (per GPU): stream, reduction scratch, pinned scalars.
struct Context {
   hipStream_t stream      = nullptr;
   hipStream_t comm_stream = nullptr; // halo transfers that run under a product kernel (row-partitioned runs)
   double     *partials    = nullptr; // [kNumSlots][kRedBlocks] block partials
   double     *scalars     = nullptr; // device scalars (gamma, alpha, ...)
   double     *host_scalars = nullptr; // pinned host mirror for async read-back
   hipEvent_t  ev           = nullptr;
   int         device       = 0;
   static constexpr int kNumSlots   = 8;
   static constexpr int kNumScalars = 320; // recurrence scalars + GMRES Gram-Schmidt coefficients (krylov_dim up to ~300)

   static Context &get();
   static void     release_thread(); // (thread ranks: called by leave_thread_rank)
   void            sync() { wait_stream(stream); }
   double         *slot(int s) { return partials + (size_t)s * kRedBlocks; }

 private:
   Context();
};

// Caching device allocator.  Everything in this library runs on ONE stream, so a block
// released by the host can be handed to the next request immediately: any kernel that
// still reads it was enqueued earlier on the same stream than any kernel of the new owner.
// (hipMallocAsync/hipFreeAsync pools were tried first and deadlocked inside the runtime on
// ROCm 7.2 / gfx950 when a freed 50 MB block was re-requested at another size.)
void *pool_alloc(size_t bytes);
void  pool_free(void *p);
void  pool_trim(); // sync + return all cached blocks to the driver
size_t pool_bytes_in_use();
size_t pool_bytes_peak();
size_t pool_bytes_cached(); // released blocks kept for reuse: at most max(2 x peak, HDA_POOL_CACHE_MIN_GB)
void   pool_driver_stats(double out[3], bool reset); // hipMalloc calls that reached the driver, host ms spent in them, bytes they returned

// Stream-ordered device array.
template <class T>
class DArray {
 public:
   DArray() = default;
   explicit DArray(size_t n) { alloc(n); }
   DArray(const DArray &) = delete;
   DArray &operator=(const DArray &) = delete;
   DArray(DArray &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
   DArray &operator=(DArray &&o) noexcept
   {
      if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
      return *this;
   }
   ~DArray() { release(); }

   void alloc(size_t n)
   {
      release();
      n_ = n;
      if (n == 0) return;
      p_ = (T *)pool_alloc(n * sizeof(T));
   }
   void release()
   {
      if (p_) pool_free(p_);
      p_ = nullptr;
      n_ = 0;
   }
   void zero()
   {
      if (n_) HDA_HIP(hipMemsetAsync(p_, 0, n_ * sizeof(T), Context::get().stream));
   }
   void upload(const T *h, size_t n)
   {
      if (n_ != n) alloc(n);
      if (n) HDA_HIP(hipMemcpyAsync(p_, h, n * sizeof(T), hipMemcpyHostToDevice, Context::get().stream));
      Context::get().sync(); // host buffer may be pageable / transient
   }
   void download(T *h, size_t n) const
   {
      HDA_REQUIRE(n <= n_, "download larger than array");
      if (n) HDA_HIP(hipMemcpyAsync(h, p_, n * sizeof(T), hipMemcpyDeviceToHost, Context::get().stream));
      Context::get().sync();
   }
   std::vector<T> to_host() const
   {
      std::vector<T> v(n_);
      download(v.data(), n_);
      return v;
   }
   void copy_from(const DArray<T> &o)
   {
      if (n_ != o.n_) alloc(o.n_);
      if (n_) HDA_HIP(hipMemcpyAsync(p_, o.p_, n_ * sizeof(T), hipMemcpyDeviceToDevice, Context::get().stream));
   }
   T       *data() { return p_; }
   const T *data() const { return p_; }
   size_t   size() const { return n_; }

 private:
   T     *p_ = nullptr;
   size_t n_ = 0;
};

// Ghost-column entries of a row block, by boundary row (built lazily by the first overlapped product; hda_kernels.hip)
struct OffdPart {
   int            nown = -1, nbrows = 0, nnz = 0; // columns >= nown are ghosts; rows with any; their entries
   DArray<int>    brow, rp, col;
   DArray<double> val;
};

// identity stamp of a matrix's arrays: every DCsr is born with a fresh one and reset_plan() (arrays replaced) draws another, so
// a cache keyed on it (the Gauss-Seidel row spans) cannot mistake a later matrix at the same address and size for its own
inline unsigned long long next_csr_gen()
{
   static std::atomic<unsigned long long> g{1};
   return g.fetch_add(1, std::memory_order_relaxed);
}

// Local CSR block resident in HBM. int32 indices, fp64 values, rows column-sorted.
// ncols may exceed nrows: columns >= nrows address the ghost tail of an extended vector
// (row-partitioned case: [owned | halo]).
struct DCsr {
   int            nrows = 0, ncols = 0, nnz = 0;
   mutable unsigned long long gen = next_csr_gen();
   DArray<int>    rowptr; // nrows+1
   DArray<int>    col;    // nnz
   DArray<double> val;    // nnz
   double         avg_row() const { return nrows ? (double)nnz / nrows : 0.0; }
   // streaming plan of the LDS-staged SpMV (built lazily by the first product): chunk c owns
   // the rows whose first entry lies in [c*kChunk, (c+1)*kChunk)
   mutable DArray<int> chunk_row; // nchunks+1
   mutable int         nchunks = 0, maxrow = -1;
   // stencil-coded shadow of (col, val) for operators with few distinct (col - row, value)
   // pairs (constant-coefficient discretisations): one byte per entry indexing a dictionary of
   // <= 255 pairs, 255 = "escape, read (col, val) from the plain arrays".  Built lazily next to
   // the streaming plan; products and their order are those of the plain arrays.
   mutable DArray<unsigned char> code;       // nnz (padded to a multiple of 4)
   mutable DArray<double>        dict_val;   // 256
   mutable DArray<int>           dict_delta; // 256
   mutable int                   coded = -1, escapes = 0; // -1 not examined, 0 plain, 1 coded; entries outside the dictionary
   // row classes of a stencil-coded operator (coded == 1): rows that spell the same sequence of <= 8 entry codes share a
   // class; one byte per ROW then replaces row pointer and entry codes (hda_kernels.hip "row-class coding")
   mutable DArray<unsigned char>      rclass;  // nrows (+ padding); 255 = read the row from the CSR arrays
   mutable DArray<unsigned long long> rc_keys; // 128 slots: the packed code sequence of every class
   mutable int                        rowcoded = 0, rc_esc_rows = 0;
   mutable long long                  rc_esc_entries = 0;
   mutable std::unique_ptr<OffdPart> offd; // ghost-column part, for products overlapped with their halo exchange
   // windowed form of the column indices (hda_kernels.hip "windowed CSR"): chunks of ~1024 entries (whole rows); per chunk the
   // ascending list of its DISTINCT columns (ucol[uoff[c] .. uoff[c+1])) and per entry the 2-byte position of its column in it
   mutable DArray<unsigned short> lidx;          // nnz
   mutable DArray<int>            ucol, wmeta;   // distinct columns; 3 ints per chunk + sentinel: first row, first entry, first distinct
   mutable int                    win = -1, nwin = 0, win_maxu = 0; // -1 not examined, 0 plain, 1 windowed; chunks; most distinct columns in a chunk
   mutable long long              win_total = 0;
   // run form of the windowed operator: where every chunk's distinct columns are at most kWinRuns runs of consecutive indices (a
   // structured-grid operator in lexicographic order) ucol holds, per chunk, 2 * kWinRuns ints (first position, first column of
   // every run; unused runs start at INT_MAX) instead of the list: the kernel computes a position's column instead of loading it
   mutable bool                   win_runs = false;
   void reset_plan() const
   {
      chunk_row.release(); code.release(); dict_val.release(); dict_delta.release(); offd.reset();
      rclass.release(); rc_keys.release(); lidx.release(); ucol.release(); wmeta.release();
      gen = next_csr_gen();
      nchunks = 0; maxrow = -1; coded = -1; rowcoded = 0; rc_esc_rows = 0; rc_esc_entries = 0; win = -1; nwin = 0; win_maxu = 0; win_total = 0; win_runs = false;
   }
};

// HDA_VERBOSE=1: phase trace on stderr (each trace point synchronises the stream)
inline bool verbose()
{
   static const bool v = getenv("HDA_VERBOSE") != nullptr;
   return v;
}
inline double trace_ms() // milliseconds since the first trace point
{
   static const auto t0 = std::chrono::steady_clock::now();
   return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define HDA_TRACE(...)                                 \
   do {                                                \
      if (::hda::verbose())                            \
      {                                                \
         ::hda::Context::get().sync();                 \
         fprintf(stderr, "[hda %9.2f] ", ::hda::trace_ms()); \
         fprintf(stderr, __VA_ARGS__);                 \
         fprintf(stderr, "\n");                        \
         fflush(stderr);                               \
      }                                                \
   } while (0)

// Host read-back ordered on the library stream.  The stream is created non-blocking, so a
// null-stream hipMemcpy does NOT wait for kernels enqueued on it: every device -> host copy
// of data the library's kernels produce goes through here.
inline void download_sync(void *host, const void *dev, size_t bytes)
{
   if (bytes) HDA_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, Context::get().stream));
   Context::get().sync();
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

} // namespace hda
