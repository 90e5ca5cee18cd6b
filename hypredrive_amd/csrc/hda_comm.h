// hda_comm.h -- rank-to-rank communication of the row-partitioned solve path.
//
// One process drives one MI355X; ranks are connected with RCCL over xGMI (C1 halo exchange
// = grouped ncclSend/ncclRecv between neighbours, C2 = one small ncclAllReduce per fused
// group of dot products).  RCCL refuses two ranks on one device, so for
// multi-rank tests on a single GPU (and for gloo-only hosts) a second transport stages the
// same messages through host callbacks supplied by the launcher (torch.distributed gloo).
// Both transports sit behind one interface; everything above it is transport agnostic.
#pragma once

#include "hda_common.h"

namespace hda {

// host callbacks of the staged transport (see include/HYPREDRV.h, HYPREDRV_AMD_CommInitCallbacks)
// Both return 0 on success; anything else makes the library raise an error on this rank (a callback that failed
// silently would leave unfilled receive buffers behind: wrong halos and dot products, or deadlocked peers).
typedef int (*hda_allreduce_cb)(void *buf, long count, int dtype /*0 f64, 1 i64*/, int op /*0 sum, 1 max*/);
// send/recv are packed by ascending peer rank; counts in BYTES, arrays of length world_size
typedef int (*hda_alltoallv_cb)(const void *send, const long *send_bytes, void *recv, const long *recv_bytes);

class Comm {
 public:
   int rank = 0, size = 1;
   // traffic counters of the solve path (bench.py: collectives per iteration)
   struct Stats {
      long allreduce = 0, exchange = 0, allreduce_doubles = 0, exchange_doubles = 0, overlapped = 0;
   } stats;
   virtual ~Comm() = default;
   // in-place sum of n doubles living in HBM, ordered on the library stream
   virtual void allreduce_sum_dev(double *dbuf, int n) = 0;
   // neighbour exchange of doubles in HBM: send_dev is packed by ascending destination rank,
   // recv_dev by ascending source rank; counts in elements, host arrays of length `size`.
   // Ordered on stream `s` (the library stream, or the communication stream of an overlapped exchange);
   // the staged transport blocks the host until the messages have arrived.
   virtual void exchange_dev(const double *send_dev, const int *send_counts, double *recv_dev, const int *recv_counts, hipStream_t s) = 0;
   void exchange_dev(const double *send_dev, const int *send_counts, double *recv_dev, const int *recv_counts)
   {
      exchange_dev(send_dev, send_counts, recv_dev, recv_counts, Context::get().stream);
   }
   // true when exchange_dev only enqueues work (RCCL); false when it blocks the host (staged callbacks)
   virtual bool async_exchange() const { return false; }
   // setup-time host collectives
   virtual void allreduce_host(long long *v, int n, int op /*0 sum 1 max*/) = 0;
   virtual void alltoallv_host(const void *send, const long *send_bytes, void *recv, const long *recv_bytes) = 0;
   virtual const char *name() const = 0;

   // helpers built on the above
   void allgather_ll(long long mine, std::vector<long long> &all);
   // variable-size all-gather of bytes (every rank receives everybody's block, rank order)
   void allgatherv_bytes(const void *mine, long nbytes, std::vector<char> &out, std::vector<long> &counts);

   static Comm &world();
   static void  set_world(Comm *c); // takes ownership
   // true once the launcher has called HYPREDRV_AMD_CommInit / CommInitCallbacks: the MPI join (hda_mpi.cpp) then stays out
   static bool  explicitly_joined();
   static void  set_explicitly_joined(bool v);
};

Comm *make_self_comm();
Comm *make_rccl_comm(int rank, int size, const void *unique_id_128);
Comm *make_callback_comm(int rank, int size, hda_allreduce_cb ar, hda_alltoallv_cb a2a);
void  rccl_get_unique_id(void *out_128);

} // namespace hda
