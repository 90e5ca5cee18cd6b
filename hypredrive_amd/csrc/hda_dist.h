// hda_dist.h -- row-block partition plumbing (hypre ParCSR "diag + offd" in one extended
// CSR): local blocks address an extended vector [owned | ghosts], a HaloPlan refreshes the
// ghost tail before the operator is applied (hypre's ParCSRCommPkg).
#pragma once

#include "hda_comm.h"

namespace hda {

struct HaloPlan {
   int              nloc = 0, nghost = 0, send_total = 0;
   std::vector<int> send_counts, recv_counts; // per peer rank (length comm size)
   DArray<int>      send_idx;                 // owned indices to pack, grouped by ascending destination
   DArray<double>   send_buf;
   // events of an exchange that runs beside a product: send buffer packed (library stream) / ghosts landed (communication stream)
   mutable hipEvent_t ev_packed = nullptr, ev_landed = nullptr;
   HaloPlan() = default;
   HaloPlan(const HaloPlan &) = delete;
   HaloPlan &operator=(const HaloPlan &) = delete;
   HaloPlan(HaloPlan &&o) noexcept { *this = std::move(o); }
   HaloPlan &operator=(HaloPlan &&o) noexcept
   {
      if (this != &o)
      {
         drop_events();
         nloc = o.nloc; nghost = o.nghost; send_total = o.send_total;
         send_counts = std::move(o.send_counts); recv_counts = std::move(o.recv_counts);
         send_idx = std::move(o.send_idx); send_buf = std::move(o.send_buf);
         ev_packed = o.ev_packed; ev_landed = o.ev_landed;
         o.ev_packed = o.ev_landed = nullptr;
      }
      return *this;
   }
   ~HaloPlan() { drop_events(); }
   void drop_events()
   {
      if (ev_packed) (void)hipEventDestroy(ev_packed);
      if (ev_landed) (void)hipEventDestroy(ev_landed);
      ev_packed = ev_landed = nullptr;
   }
};

// Rows [row_lo,row_hi) of the replicated global matrix G as a local block whose owned
// columns are [col_lo,col_hi): owned columns -> [0, ncol_loc), other columns -> ncol_loc +
// position in the ascending list ghost_gids (returned on the host).
void localize(const DCsr &G, long long row_lo, long long row_hi, long long col_lo, long long col_hi, DCsr &L,
              std::vector<long long> &ghost_gids);

// part: row starts of every rank (length size+1) of the vector the plan exchanges.
HaloPlan make_halo_plan(int nloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids);
// its host half (collective over Comm::world(), no device call): per-peer counts and the owned rows to pack, by ascending destination
void halo_plan_host(int nloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids, std::vector<int> &send_counts,
                    std::vector<int> &recv_counts, std::vector<int> &send_idx);

// x_ext[nloc .. nloc+nghost) <- owners' values.  Collective over Comm::world().
void halo_exchange(const HaloPlan &h, double *x_ext);
// The same exchange in three steps, for a product that runs while the values travel (hda_kernels.hip launch_spmv_halo):
bool halo_active(const HaloPlan &h);                // more than one rank and a plan that exchanges something somewhere
void halo_pack(const HaloPlan &h, const double *x); // pack the send buffer on the library stream, mark "packed"
void halo_transfer(const HaloPlan &h, double *x);   // communication stream: wait for "packed", exchange, mark "landed"
void halo_wait(const HaloPlan &h);                  // library stream waits for "landed"

// Gather the row-partitioned local blocks into the full matrix on every rank (replicated
// AMG setup).  Local columns: < nloc owned (global = part[rank] + c), else ghost_gids[c - nloc].
void gather_global(const DCsr &Aloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids,
                   DCsr &G);

} // namespace hda
