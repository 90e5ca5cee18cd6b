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
};

// Rows [row_lo,row_hi) of the replicated global matrix G as a local block whose owned
// columns are [col_lo,col_hi): owned columns -> [0, ncol_loc), other columns -> ncol_loc +
// position in the ascending list ghost_gids (returned on the host).
void localize(const DCsr &G, long long row_lo, long long row_hi, long long col_lo, long long col_hi, DCsr &L,
              std::vector<long long> &ghost_gids);

// part: row starts of every rank (length size+1) of the vector the plan exchanges.
HaloPlan make_halo_plan(int nloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids);

// x_ext[nloc .. nloc+nghost) <- owners' values.  Collective over Comm::world().
void halo_exchange(const HaloPlan &h, double *x_ext);

// Gather the row-partitioned local blocks into the full matrix on every rank (replicated
// AMG setup).  Local columns: < nloc owned (global = part[rank] + c), else ghost_gids[c - nloc].
void gather_global(const DCsr &Aloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids,
                   DCsr &G);

} // namespace hda
