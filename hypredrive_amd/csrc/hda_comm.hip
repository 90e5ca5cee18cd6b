// hda_comm.hip -- transports behind hda::Comm (see hda_comm.h).
#include "hda_comm.h"

#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>

// The host driver of this pool supports dmabuf IPC only: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL (and any device memory shared
// across processes) fails with hipIpcGetMemHandle: invalid argument.  The HIP runtime reads it when it starts, so it goes into the
// environment when this library is loaded -- before any GPU call of ours -- unless the caller has chosen a value (C callers such
// as hypredrive-cli and the reference's drivers have no other place to do it; bench.py and hypredrive_amd/dist.py also set it).
__attribute__((constructor)) static void hda_default_environment() { setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); }

namespace hda {

#define STREAM (Context::get().stream)

// the process's communicator (a thread rank of the test seam has its own, like its context: hda_common.h RankState)
#define g_world (RankState<std::unique_ptr<Comm>>::get())

Comm &Comm::world()
{
   if (!g_world) g_world.reset(make_self_comm());
   return *g_world;
}
void Comm::set_world(Comm *c) { g_world.reset(c); }
#define g_explicit (RankState<bool, 41>::get())
bool Comm::explicitly_joined() { return g_explicit; }
void Comm::set_explicitly_joined(bool v) { g_explicit = v; }

void Comm::allgather_ll(long long mine, std::vector<long long> &all)
{
   all.assign((size_t)size, 0);
   all[(size_t)rank] = mine;
   allreduce_host(all.data(), size, 0);
}

void Comm::allgatherv_bytes(const void *mine, long nbytes, std::vector<char> &out, std::vector<long> &counts)
{
   std::vector<long long> c;
   allgather_ll(nbytes, c);
   counts.assign(c.begin(), c.end());
   long total = 0;
   for (long v : counts) total += v;
   out.resize((size_t)std::max<long>(total, 1));
   // everybody sends its block to everybody
   std::vector<long> sb((size_t)size, nbytes);
   std::vector<char> send((size_t)std::max<long>(nbytes * size, 1));
   for (int r = 0; r < size; r++)
      if (nbytes) memcpy(send.data() + (size_t)r * nbytes, mine, (size_t)nbytes);
   alltoallv_host(send.data(), sb.data(), out.data(), counts.data());
}

// ------------------------------------------------------------------- self

namespace {
class SelfComm : public Comm {
 public:
   void allreduce_sum_dev(double *, int) override {}
   void exchange_dev(const double *, const int *, double *, const int *, hipStream_t) override {}
   void allreduce_host(long long *, int, int) override {}
   void alltoallv_host(const void *send, const long *sb, void *recv, const long *rb) override
   {
      if (sb[0] && rb[0]) memcpy(recv, send, (size_t)std::min(sb[0], rb[0]));
   }
   const char *name() const override { return "self"; }
};
} // namespace
Comm *make_self_comm() { return new SelfComm(); }

// ------------------------------------------------------------------- RCCL

namespace {
// the few RCCL entry points used, resolved at run time so the library itself has no
// link-time dependency (torch ships its own librccl; whichever is loaded first is used)
struct Rccl {
   void *lib = nullptr;
   int (*GetUniqueId)(void *)                                                     = nullptr;
   int (*CommDestroy)(void *)                                                     = nullptr;
   int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t)  = nullptr;
   int (*Send)(const void *, size_t, int, int, void *, hipStream_t)               = nullptr;
   int (*Recv)(void *, size_t, int, int, void *, hipStream_t)                     = nullptr;
   int (*GroupStart)()                                                            = nullptr;
   int (*GroupEnd)()                                                              = nullptr;
   const char *(*GetErrorString)(int)                                             = nullptr;
   void *CommInitRank                                                             = nullptr;
};
struct Uid128 {
   char b[128];
};
Rccl &rccl()
{
   static Rccl r;
   if (r.lib) return r;
   const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
   for (const char *n : names)
   {
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
   }
   if (!r.lib) throw Error(std::string("cannot load RCCL: ") + dlerror());
#define HDA_SYM(field, sym)                                             \
   *(void **)(&r.field) = dlsym(r.lib, sym);                            \
   if (!r.field) throw Error(std::string("RCCL symbol missing: ") + sym)
   HDA_SYM(GetUniqueId, "ncclGetUniqueId");
   HDA_SYM(CommDestroy, "ncclCommDestroy");
   HDA_SYM(AllReduce, "ncclAllReduce");
   HDA_SYM(Send, "ncclSend");
   HDA_SYM(Recv, "ncclRecv");
   HDA_SYM(GroupStart, "ncclGroupStart");
   HDA_SYM(GroupEnd, "ncclGroupEnd");
   HDA_SYM(GetErrorString, "ncclGetErrorString");
   HDA_SYM(CommInitRank, "ncclCommInitRank");
#undef HDA_SYM
   return r;
}
#define HDA_NCCL(expr)                                                                          \
   do {                                                                                         \
      int _e = (expr);                                                                          \
      if (_e != 0) throw Error(std::string("RCCL error: ") + rccl().GetErrorString(_e) + " in " #expr); \
   } while (0)

// ncclDataType_t / ncclRedOp_t values (rccl.h): ncclInt8 0, ncclInt64 4, ncclFloat64 8; ncclSum 0, ncclMax 2
enum { kInt8 = 0, kInt64 = 4, kF64 = 8, kSum = 0, kMax = 2 };

class RcclComm : public Comm {
 public:
   RcclComm(int r, int s, const void *uid)
   {
      rank = r;
      size = s;
      Uid128 id;
      memcpy(id.b, uid, 128);
      typedef int (*Init_t)(void **, int, Uid128, int);
      HDA_NCCL(((Init_t)rccl().CommInitRank)(&comm_, s, id, r));
      // A second communicator carries the neighbour exchanges: they run on the communication stream beside product kernels
      // and beside the all-reduces of the main stream.  Two communicators driven from two streams in the same order on every
      // rank is the documented-safe way to have RCCL operations in flight concurrently.  Its id is made by rank 0 and handed
      // round through the first communicator (a byte-wise sum with zeros from everybody else).
      if (s > 1)
      {
         Uid128 id2;
         memset(&id2, 0, sizeof(id2));
         if (r == 0) HDA_NCCL(rccl().GetUniqueId(&id2));
         DArray<char> d(128);
         HDA_HIP(hipMemcpyAsync(d.data(), id2.b, 128, hipMemcpyHostToDevice, STREAM));
         HDA_NCCL(rccl().AllReduce(d.data(), d.data(), 128, kInt8, kSum, comm_, STREAM));
         HDA_HIP(hipMemcpyAsync(id2.b, d.data(), 128, hipMemcpyDeviceToHost, STREAM));
         Context::get().sync();
         HDA_NCCL(((Init_t)rccl().CommInitRank)(&halo_comm_, s, id2, r));
      }
   }
   ~RcclComm() override
   {
      if (halo_comm_) rccl().CommDestroy(halo_comm_);
      if (comm_) rccl().CommDestroy(comm_);
   }
   void allreduce_sum_dev(double *d, int n) override
   {
      stats.allreduce++;
      stats.allreduce_doubles += n;
      HDA_NCCL(rccl().AllReduce(d, d, (size_t)n, kF64, kSum, comm_, STREAM));
   }
   void exchange_dev(const double *send, const int *sc, double *recv, const int *rc, hipStream_t st) override
   {
      stats.exchange++;
      void *cm = halo_comm_ ? halo_comm_ : comm_;
      HDA_NCCL(rccl().GroupStart());
      size_t so = 0, ro = 0;
      for (int p = 0; p < size; p++)
      {
         if (sc[p]) HDA_NCCL(rccl().Send(send + so, (size_t)sc[p], kF64, p, cm, st));
         if (rc[p]) HDA_NCCL(rccl().Recv(recv + ro, (size_t)rc[p], kF64, p, cm, st));
         so += (size_t)sc[p];
         ro += (size_t)rc[p];
      }
      HDA_NCCL(rccl().GroupEnd());
      stats.exchange_doubles += (long)so;
   }
   bool async_exchange() const override { return true; }
   void allreduce_host(long long *v, int n, int op) override
   {
      DArray<long long> d;
      d.upload(v, (size_t)n);
      HDA_NCCL(rccl().AllReduce(d.data(), d.data(), (size_t)n, kInt64, op ? kMax : kSum, comm_, STREAM));
      d.download(v, (size_t)n);
   }
   void alltoallv_host(const void *send, const long *sb, void *recv, const long *rb) override
   {
      long st = 0, rt = 0;
      for (int p = 0; p < size; p++) { st += sb[p]; rt += rb[p]; }
      DArray<char> ds((size_t)std::max<long>(st, 1)), dr((size_t)std::max<long>(rt, 1));
      if (st) HDA_HIP(hipMemcpyAsync(ds.data(), send, (size_t)st, hipMemcpyHostToDevice, STREAM));
      HDA_NCCL(rccl().GroupStart());
      long so = 0, ro = 0;
      for (int p = 0; p < size; p++)
      {
         if (p == rank)
         { // own block: plain device copy, no self send/recv
            if (sb[p]) HDA_HIP(hipMemcpyAsync(dr.data() + ro, ds.data() + so, (size_t)std::min(sb[p], rb[p]), hipMemcpyDeviceToDevice, STREAM));
         }
         else
         {
            if (sb[p]) HDA_NCCL(rccl().Send(ds.data() + so, (size_t)sb[p], kInt8, p, comm_, STREAM));
            if (rb[p]) HDA_NCCL(rccl().Recv(dr.data() + ro, (size_t)rb[p], kInt8, p, comm_, STREAM));
         }
         so += sb[p];
         ro += rb[p];
      }
      HDA_NCCL(rccl().GroupEnd());
      if (rt) HDA_HIP(hipMemcpyAsync(recv, dr.data(), (size_t)rt, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
   }
   const char *name() const override { return "rccl"; }

 private:
   void *comm_ = nullptr, *halo_comm_ = nullptr; // collectives on the library stream / neighbour exchanges (any stream)
};

// ------------------------------------------------------------ host-staged callbacks

class CallbackComm : public Comm {
 public:
   CallbackComm(int r, int s, hda_allreduce_cb ar, hda_alltoallv_cb a2a) : ar_(ar), a2a_(a2a)
   {
      rank = r;
      size = s;
   }
   ~CallbackComm() override
   {
      for (double *p : pin_)
         if (p) (void)hipHostFree(p);
   }
   void allreduce_sum_dev(double *d, int n) override
   {
      stats.allreduce++;
      stats.allreduce_doubles += n;
      double *h = pinned(0, (size_t)std::max(n, 1));
      HDA_HIP(hipMemcpyAsync(h, d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, STREAM));
      Context::get().sync();
      check(ar_(h, n, 0, 0), "all-reduce");
      HDA_HIP(hipMemcpyAsync(d, h, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, STREAM));
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
      // pinned staging buffers: the copies go through the DMA engines and do not queue behind a product grid
      double *hs = pinned(0, (size_t)std::max<long>(st, 1)), *hr = pinned(1, (size_t)std::max<long>(rt, 1));
      if (st) HDA_HIP(hipMemcpyAsync(hs, send, sizeof(double) * (size_t)st, hipMemcpyDeviceToHost, strm));
      wait_stream(strm);
      check(a2a_(hs, sb.data(), hr, rb.data()), "neighbour exchange");
      if (rt) HDA_HIP(hipMemcpyAsync(recv, hr, sizeof(double) * (size_t)rt, hipMemcpyHostToDevice, strm));
      wait_stream(strm);
   }
   void allreduce_host(long long *v, int n, int op) override { check(ar_(v, n, 1, op), "host all-reduce"); }
   void alltoallv_host(const void *send, const long *sb, void *recv, const long *rb) override { check(a2a_(send, sb, recv, rb), "host all-to-all"); }
   const char *name() const override { return "host-callbacks"; }

 private:
   double *pinned(int which, size_t n)
   {
      if (pin_n_[which] < n)
      {
         if (pin_[which]) (void)hipHostFree(pin_[which]);
         pin_[which] = nullptr;
         const size_t m = std::max<size_t>(n + n / 2, 4096);
         HDA_HIP(hipHostMalloc((void **)&pin_[which], sizeof(double) * m, hipHostMallocDefault));
         pin_n_[which] = m;
      }
      return pin_[which];
   }
   double *pin_[2]   = {nullptr, nullptr};
   size_t  pin_n_[2] = {0, 0};
   static void check(int rc, const char *what)
   {
      if (rc != 0) throw Error(std::string("staged transport: the launcher's ") + what + " callback failed (code " + std::to_string(rc) + ")");
   }
   hda_allreduce_cb ar_;
   hda_alltoallv_cb a2a_;
};
} // namespace

// (the test transport "ranks as threads of one process" lives in hda_testranks_comm.hip: libhypredrv_amd_testranks.so, not this library)

Comm *make_rccl_comm(int rank, int size, const void *uid) { return new RcclComm(rank, size, uid); }
Comm *make_callback_comm(int rank, int size, hda_allreduce_cb ar, hda_alltoallv_cb a2a)
{
   return new CallbackComm(rank, size, ar, a2a);
}
void rccl_get_unique_id(void *out_128)
{
   Uid128 id;
   memset(&id, 0, sizeof(id));
   HDA_NCCL(rccl().GetUniqueId(&id));
   memcpy(out_128, id.b, 128);
}

} // namespace hda
