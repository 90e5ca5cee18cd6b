// hda_testranks_comm.hip -- TEST transport: the ranks of a row partition as threads of one process.
// Built into libhypredrv_amd_testranks.so together with hda_thread_ranks.hip; the product library libhypredrv_amd.so holds neither
// (it keeps the per-thread state seam they stand on: enter_thread_rank / leave_thread_rank, hda_common.h).
#include "hda_testranks.h"

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>

namespace hda {

#define STREAM (Context::get().stream)

// ------------------------------------------------------------ ranks as threads of one process
//
// Test transport: the ranks of a row partition are THREADS of one process (each with its own context, stream, allocator and
// this communicator; hda_thread_ranks.hip).  It exists because a GPU box admits few processes on its card: the 2x2x2 layout
// of BASELINE config 3 needs eight ranks.  Messages are staged through host memory like the callback transport; collectives
// meet at a generation barrier; sums run in rank order on every rank (deterministic, identical everywhere).
namespace {
struct ThreadWorld {
   explicit ThreadWorld(int n) : size(n), ptr((size_t)n, nullptr), cnt((size_t)n, nullptr), kind((size_t)n, 0), len((size_t)n, 0) {}
   virtual ~ThreadWorld() = default;
   int                       size;
   std::mutex                mu;
   std::condition_variable   cv;
   int                       arrived = 0;
   unsigned long             gen     = 0;
   bool                      failed  = false; // a rank died: release everybody instead of deadlocking the process
   std::vector<const void *> ptr;
   std::vector<const long *> cnt;
   std::vector<int>          kind; // which collective a rank has entered (checked by all after the first barrier of a call:
   std::vector<long>         len;  //  ranks that disagree would read each other's stale pointers) and its operand's length
   // called between the two barriers of a collective, before anybody dereferences a peer's pointers
   void same_call(int rank, int k, long n, const char *what) const
   {
      for (int p = 0; p < size; p++)
         if (kind[(size_t)p] != k || (n >= 0 && len[(size_t)p] != n))
            throw Error(std::string("thread ranks: ranks are in different collectives: rank ") + std::to_string(rank) + " in " + what + "(" +
                        std::to_string(n) + "), rank " + std::to_string(p) + " in kind " + std::to_string(kind[(size_t)p]) + "(" +
                        std::to_string(len[(size_t)p]) + ")");
   }
   void barrier()
   {
      std::unique_lock<std::mutex> lk(mu);
      if (failed) throw Error("thread ranks: another rank failed");
      const unsigned long g = gen;
      if (++arrived == size)
      {
         arrived = 0;
         gen++;
         cv.notify_all();
         return;
      }
      cv.wait(lk, [&] { return gen != g || failed; });
      if (failed && gen == g) throw Error("thread ranks: another rank failed");
   }
   void fail()
   {
      std::lock_guard<std::mutex> lk(mu);
      failed = true;
      cv.notify_all();
   }
};

class ThreadComm : public Comm {
 public:
   ThreadComm(int r, std::shared_ptr<ThreadWorld> w) : w_(std::move(w))
   {
      rank = r;
      size = w_->size;
   }
   void allreduce_sum_dev(double *d, int n) override
   {
      stats.allreduce++;
      stats.allreduce_doubles += n;
      hbuf_.resize((size_t)std::max(n, 1));
      HDA_HIP(hipMemcpyAsync(hbuf_.data(), d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
      reduce(hbuf_.data(), n, 0);
      HDA_HIP(hipMemcpyAsync(d, hbuf_.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, STREAM));
      Context::get().sync();
   }
   void exchange_dev(const double *send, const int *sc, double *recv, const int *rc, hipStream_t strm) override
   {
      stats.exchange++;
      long              st = 0, rt = 0;
      std::vector<long> sb((size_t)size), rb((size_t)size);
      for (int p = 0; p < size; p++)
      {
         sb[(size_t)p] = 8L * sc[p];
         rb[(size_t)p] = 8L * rc[p];
         st += sc[p];
         rt += rc[p];
      }
      stats.exchange_doubles += st;
      hs_.resize((size_t)std::max<long>(st, 1));
      hr_.resize((size_t)std::max<long>(rt, 1));
      if (st) HDA_HIP(hipMemcpyAsync(hs_.data(), send, sizeof(double) * (size_t)st, hipMemcpyDeviceToHost, strm));
      HDA_HIP(hipStreamSynchronize(strm));
      alltoallv_host(hs_.data(), sb.data(), hr_.data(), rb.data());
      if (rt) HDA_HIP(hipMemcpyAsync(recv, hr_.data(), sizeof(double) * (size_t)rt, hipMemcpyHostToDevice, strm));
      HDA_HIP(hipStreamSynchronize(strm));
   }
   void allreduce_host(long long *v, int n, int op) override
   {
      std::vector<long long> t((size_t)std::max(n, 1));
      w_->ptr[(size_t)rank]  = v;
      w_->kind[(size_t)rank] = 2;
      w_->len[(size_t)rank]  = n;
      w_->barrier();
      w_->same_call(rank, 2, n, "allreduce_host");
      for (int i = 0; i < n; i++)
      {
         long long a = ((const long long *)w_->ptr[0])[i];
         for (int p = 1; p < size; p++)
         {
            const long long b = ((const long long *)w_->ptr[(size_t)p])[i];
            a = op ? std::max(a, b) : a + b;
         }
         t[(size_t)i] = a;
      }
      w_->barrier();
      if (n) memcpy(v, t.data(), sizeof(long long) * (size_t)n);
   }
   void alltoallv_host(const void *send, const long *sb, void *recv, const long *rb) override
   {
      w_->ptr[(size_t)rank]  = send;
      w_->cnt[(size_t)rank]  = sb;
      w_->kind[(size_t)rank] = 3;
      w_->barrier();
      w_->same_call(rank, 3, -1, "alltoallv_host");
      long ro = 0;
      for (int p = 0; p < size; p++)
      {
         const long *psb = w_->cnt[(size_t)p];
         long        so  = 0;
         for (int q = 0; q < rank; q++) so += psb[q];
         const long nb = psb[rank];
         if (nb != rb[p])
            throw Error("thread ranks: alltoallv_host: rank " + std::to_string(p) + " sends " + std::to_string(nb) + " bytes to rank " +
                        std::to_string(rank) + ", which expects " + std::to_string(rb[p]));
         if (nb) memcpy((char *)recv + ro, (const char *)w_->ptr[(size_t)p] + so, (size_t)nb);
         ro += rb[p];
      }
      w_->barrier();
   }
   const char *name() const override { return "threads"; }

 protected:
   void reduce(double *v, int n, int)
   {
      std::vector<double> t((size_t)std::max(n, 1));
      w_->ptr[(size_t)rank]  = v;
      w_->kind[(size_t)rank] = 1;
      w_->len[(size_t)rank]  = n;
      w_->barrier();
      w_->same_call(rank, 1, n, "allreduce_sum_dev");
      for (int i = 0; i < n; i++)
      {
         double a = ((const double *)w_->ptr[0])[i];
         for (int p = 1; p < size; p++) a += ((const double *)w_->ptr[(size_t)p])[i];
         t[(size_t)i] = a;
      }
      w_->barrier();
      if (n) memcpy(v, t.data(), sizeof(double) * (size_t)n);
   }
   std::shared_ptr<ThreadWorld> w_;
   std::vector<double>          hbuf_, hs_, hr_;
};

// The same ranks-as-threads world with a transport that behaves like RCCL towards the caller (HDA_THREAD_TRANSPORT=device): an
// exchange or a device all-reduce only ENQUEUES work on the caller's stream -- device-to-device copies straight out of the peers'
// buffers, ordered by events -- and returns; nothing waits for the GPU on the host.  (The host threads still meet at the world's barrier
// while they enqueue, which RCCL does not need; what matters is that the DEVICE side is asynchronous.)  This is the one-GPU rehearsal
// of everything the library does around an asynchronous transport: the overlapped products (pack | owned-column part || transfer |
// ghost-column part, hda_kernels.hip launch_spmv_halo), buffers reused while a peer may still read them, results read back by the host
// without a transport-side synchronisation to lean on.
constexpr int kMaxThreadRanks = 64;
struct PtrTable {
   const double *p[kMaxThreadRanks];
};
__global__ __launch_bounds__(256) void k_sum_ranks(int n, int nranks, PtrTable t, double *__restrict__ out)
{
   for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
   {
      double a = t.p[0][i];
      for (int r = 1; r < nranks; r++) a += t.p[r][i]; // rank order on every rank: identical, deterministic sums
      out[i] = a;
   }
}
// HDA_THREAD_JITTER=<max microseconds>: a rank-and-call dependent delay in front of every send-ready / operand-ready event, so the
// ranks' device timelines drift apart the way eight GPUs' do; results must not depend on it (a missing event wait would)
__global__ void k_spin(long long cycles)
{
   const long long t0 = wall_clock64();
   while (wall_clock64() - t0 < cycles) {}
}
static void jitter(hipStream_t st, int rank, unsigned long &calls)
{
   static const int max_us = getenv("HDA_THREAD_JITTER") ? atoi(getenv("HDA_THREAD_JITTER")) : 0;
   if (max_us <= 0) return;
   unsigned long long h = (unsigned long long)(rank + 1) * 0x9E3779B97F4A7C15ull + (++calls) * 0xC2B2AE3D27D4EB4Full;
   h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
   const long long us = (long long)(h % (unsigned long long)(max_us + 1));
   if (us) k_spin<<<1, 1, 0, st>>>(us * 100); // wall_clock64 ticks at 100 MHz on gfx950
}
struct DeviceThreadWorld : ThreadWorld {
   explicit DeviceThreadWorld(int n)
      : ThreadWorld(n), xsend((size_t)n, nullptr), xcnt((size_t)n, nullptr), arptr((size_t)n, nullptr), ev_ready((size_t)n, nullptr),
        ev_done((size_t)n, nullptr), ar_ready((size_t)n, nullptr), ar_done((size_t)n, nullptr)
   {
   }
   std::vector<const double *> xsend;  // rank -> its packed send buffer (device)
   std::vector<const int *>    xcnt;   // rank -> its send counts by destination (host, valid between the two barriers of a call)
   std::vector<const double *> arptr;  // rank -> its staged all-reduce operand (device)
   std::vector<hipEvent_t>     ev_ready, ev_done, ar_ready, ar_done;
};

class DeviceThreadComm : public ThreadComm {
 public:
   DeviceThreadComm(int r, std::shared_ptr<DeviceThreadWorld> w) : ThreadComm(r, w), d_(std::move(w))
   {
      HDA_REQUIRE(size <= kMaxThreadRanks, "thread ranks: the device transport handles at most 64 ranks");
      for (hipEvent_t *e : {&d_->ev_ready[(size_t)r], &d_->ev_done[(size_t)r], &d_->ar_ready[(size_t)r], &d_->ar_done[(size_t)r]})
         HDA_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
      d_->barrier(); // every rank's events exist before anybody waits on one
   }
   ~DeviceThreadComm() override
   {
      (void)hipDeviceSynchronize(); // peers may still be reading this rank's buffers
      for (hipEvent_t e : {d_->ev_ready[(size_t)rank], d_->ev_done[(size_t)rank], d_->ar_ready[(size_t)rank], d_->ar_done[(size_t)rank]})
         if (e) (void)hipEventDestroy(e);
   }
   bool async_exchange() const override { return true; }
   const char *name() const override { return "threads-device"; }
   void exchange_dev(const double *send, const int *sc, double *recv, const int *rc, hipStream_t st) override
   {
      stats.exchange++;
      for (int p = 0; p < size; p++) stats.exchange_doubles += sc[p];
      d_->xsend[(size_t)rank] = send;
      d_->xcnt[(size_t)rank]  = sc;
      d_->kind[(size_t)rank]  = 4;
      jitter(st, rank, calls_);
      HDA_HIP(hipEventRecord(d_->ev_ready[(size_t)rank], st)); // my send buffer is packed once `st` gets here
      d_->barrier();
      d_->same_call(rank, 4, -1, "exchange_dev");
      size_t ro = 0;
      for (int p = 0; p < size; p++)
      {
         if (!rc[p]) continue;
         const int *pc = d_->xcnt[(size_t)p];
         size_t     so = 0;
         for (int q = 0; q < rank; q++) so += (size_t)pc[q];
         HDA_REQUIRE(pc[rank] == rc[p], "thread ranks: send and receive counts of a neighbour exchange disagree");
         HDA_HIP(hipStreamWaitEvent(st, d_->ev_ready[(size_t)p], 0));
         HDA_HIP(hipMemcpyAsync(recv + ro, d_->xsend[(size_t)p] + so, sizeof(double) * (size_t)rc[p], hipMemcpyDeviceToDevice, st));
         ro += (size_t)rc[p];
      }
      HDA_HIP(hipEventRecord(d_->ev_done[(size_t)rank], st)); // I have read what I needed from my peers
      d_->barrier();
      for (int p = 0; p < size; p++) // like a send that has completed: later work on `st` may overwrite the send buffer
         if (sc[p]) HDA_HIP(hipStreamWaitEvent(st, d_->ev_done[(size_t)p], 0));
   }
   void allreduce_sum_dev(double *d, int n) override
   {
      stats.allreduce++;
      stats.allreduce_doubles += n;
      if (n <= 0) return;
      hipStream_t st = STREAM;
      for (int p = 0; p < size; p++) // peers have finished reading my previous operand (recorded before the last barrier of that call)
         if (p != rank) HDA_HIP(hipStreamWaitEvent(st, d_->ar_done[(size_t)p], 0));
      if (stage_.size() < (size_t)n) stage_.alloc((size_t)n);
      HDA_HIP(hipMemcpyAsync(stage_.data(), d, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
      jitter(st, rank, calls_);
      HDA_HIP(hipEventRecord(d_->ar_ready[(size_t)rank], st));
      d_->arptr[(size_t)rank] = stage_.data();
      d_->kind[(size_t)rank]  = 5;
      d_->len[(size_t)rank]   = n;
      d_->barrier();
      d_->same_call(rank, 5, n, "allreduce_sum_dev");
      PtrTable t;
      for (int p = 0; p < size; p++)
      {
         t.p[p] = d_->arptr[(size_t)p];
         if (p != rank) HDA_HIP(hipStreamWaitEvent(st, d_->ar_ready[(size_t)p], 0));
      }
      k_sum_ranks<<<std::min(ceil_div(n, 256), 1024), 256, 0, st>>>(n, size, t, d);
      HDA_HIP(hipEventRecord(d_->ar_done[(size_t)rank], st));
      d_->barrier();
   }

 private:
   std::shared_ptr<DeviceThreadWorld> d_;
   DArray<double>                     stage_;
   unsigned long                      calls_ = 0;
};
} // namespace
std::shared_ptr<void> make_thread_world(int size)
{
   const char *t = getenv("HDA_THREAD_TRANSPORT");
   if (t && !strcmp(t, "device")) return std::static_pointer_cast<ThreadWorld>(std::make_shared<DeviceThreadWorld>(size));
   return std::make_shared<ThreadWorld>(size);
}
Comm *make_thread_comm(int rank, const std::shared_ptr<void> &world)
{
   auto w = std::static_pointer_cast<ThreadWorld>(world);
   if (auto d = std::dynamic_pointer_cast<DeviceThreadWorld>(w)) return new DeviceThreadComm(rank, d);
   return new ThreadComm(rank, w);
}
void  thread_world_fail(const std::shared_ptr<void> &world) { std::static_pointer_cast<ThreadWorld>(world)->fail(); }

} // namespace hda
