// hda_dist.hip -- see hda_dist.h
#include "hda_dist.h"

#include "hda_kernels.h"

#include <algorithm>
#include <cstring>

namespace hda {

#define STREAM (Context::get().stream)

__global__ __launch_bounds__(256) void k_slice_rowptr(int n, const int *__restrict__ grp, long long row_lo, int *__restrict__ rp)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i <= n) rp[i] = grp[row_lo + i] - grp[row_lo];
}
__global__ __launch_bounds__(256) void k_flag_ghost(long nnz, const int *__restrict__ gcol, long long col_lo, long long col_hi,
                                                    int *__restrict__ flags)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256)
   {
      const int c = gcol[k];
      if (c < col_lo || c >= col_hi) flags[c] = 1;
   }
}
__global__ __launch_bounds__(256) void k_remap_cols(long nnz, const int *__restrict__ gcol, const double *__restrict__ gval,
                                                    long long col_lo, long long col_hi, int ncol_loc,
                                                    const int *__restrict__ gpos, int *__restrict__ lcol,
                                                    double *__restrict__ lval)
{
   for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (long)gridDim.x * 256)
   {
      const int c = gcol[k];
      lcol[k]     = (c >= col_lo && c < col_hi) ? (int)(c - col_lo) : ncol_loc + gpos[c];
      lval[k]     = gval[k];
   }
}
__global__ __launch_bounds__(256) void k_compact_ghost(int ncols, const int *__restrict__ flags, const int *__restrict__ gpos,
                                                       long long *__restrict__ out)
{
   const int c = blockIdx.x * 256 + threadIdx.x;
   if (c < ncols && flags[c]) out[gpos[c]] = c;
}

void localize(const DCsr &G, long long row_lo, long long row_hi, long long col_lo, long long col_hi, DCsr &L,
              std::vector<long long> &ghost_gids)
{
   const int n = (int)(row_hi - row_lo);
   L.nrows     = n;
   L.rowptr.alloc((size_t)n + 1);
   k_slice_rowptr<<<ceil_div(n + 1, 256), 256, 0, STREAM>>>(n, G.rowptr.data(), row_lo, L.rowptr.data());
   int k0 = 0, k1 = 0;
   HDA_HIP(hipMemcpyAsync(&k0, G.rowptr.data() + row_lo, 4, hipMemcpyDeviceToHost, STREAM));
   HDA_HIP(hipMemcpyAsync(&k1, G.rowptr.data() + row_hi, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   const long nnz = k1 - k0;
   L.nnz          = (int)nnz;
   L.col.alloc((size_t)std::max<long>(nnz, 1));
   L.val.alloc((size_t)std::max<long>(nnz, 1));
   DArray<int> flags((size_t)G.ncols + 1), gpos((size_t)G.ncols + 1);
   flags.zero();
   const int g = (int)std::min<long>(std::max<long>((nnz + 255) / 256, 1), 1 << 16);
   if (nnz) k_flag_ghost<<<g, 256, 0, STREAM>>>(nnz, G.col.data() + k0, col_lo, col_hi, flags.data());
   exclusive_scan(G.ncols, flags.data(), gpos.data(), nullptr);
   int nghost = 0;
   HDA_HIP(hipMemcpyAsync(&nghost, gpos.data() + G.ncols, 4, hipMemcpyDeviceToHost, STREAM));
   Context::get().sync();
   const int ncol_loc = (int)(col_hi - col_lo);
   L.ncols            = ncol_loc + nghost;
   if (nnz)
      k_remap_cols<<<g, 256, 0, STREAM>>>(nnz, G.col.data() + k0, G.val.data() + k0, col_lo, col_hi, ncol_loc, gpos.data(),
                                          L.col.data(), L.val.data());
   ghost_gids.assign((size_t)nghost, 0);
   if (nghost)
   {
      DArray<long long> gg((size_t)nghost);
      k_compact_ghost<<<ceil_div(G.ncols, 256), 256, 0, STREAM>>>(G.ncols, flags.data(), gpos.data(), gg.data());
      gg.download(ghost_gids.data(), (size_t)nghost);
   }
   // owned columns keep their relative order and ghosts sort after them in ascending global
   // order, but a ghost below col_lo now sorts after the owned block: restore sorted rows
   sort_rows(L);
   Context::get().sync();
}

// Host half of the plan (no device call: also driven on CPU by the gloo tests through hda_halo_plan_host): who owns every
// ghost, how many entries every peer wants from this rank, and which owned rows they are (grouped by ascending destination).
void halo_plan_host(int nloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids, std::vector<int> &send_counts,
                    std::vector<int> &recv_counts, std::vector<int> &send_idx)
{
   Comm &cm = Comm::world();
   send_counts.assign((size_t)cm.size, 0);
   recv_counts.assign((size_t)cm.size, 0);
   send_idx.clear();
   if (cm.size == 1)
   { // a matrix with off-rank columns on a one-rank communicator would read a ghost tail nobody ever writes (e.g. a
     // driver started under mpiexec -n > 1 whose ranks never joined the library's communicator)
      HDA_REQUIRE(ghost_gids.empty(), "matrix has off-rank columns but no communicator was joined (call HYPREDRV_AMD_CommInit / "
                                      "HYPREDRV_AMD_CommInitCallbacks on every rank before assembling row blocks)");
      return;
   }
   // who owns each ghost (ghost_gids ascending => grouped by ascending owner)
   for (long long g : ghost_gids)
   {
      int owner = (int)(std::upper_bound(part.begin(), part.end(), g) - part.begin()) - 1;
      HDA_REQUIRE(owner >= 0 && owner < cm.size && owner != cm.rank, "ghost column without a remote owner");
      recv_counts[(size_t)owner]++;
   }
   // tell every owner how many (then which) entries are wanted
   std::vector<long>      eight((size_t)cm.size, 8);
   std::vector<long long> want((size_t)cm.size), asked((size_t)cm.size);
   for (int p = 0; p < cm.size; p++) want[(size_t)p] = recv_counts[(size_t)p];
   cm.alltoallv_host(want.data(), eight.data(), asked.data(), eight.data());
   std::vector<long> sb((size_t)cm.size), rb((size_t)cm.size);
   long              tot = 0;
   for (int p = 0; p < cm.size; p++)
   {
      send_counts[(size_t)p] = (int)asked[(size_t)p];
      sb[(size_t)p]          = 8L * recv_counts[(size_t)p]; // my request lists go out
      rb[(size_t)p]          = 8L * send_counts[(size_t)p]; // peers' request lists come in
      tot += send_counts[(size_t)p];
   }
   std::vector<long long> req((size_t)std::max<long>(tot, 1));
   cm.alltoallv_host(ghost_gids.empty() ? (const void *)req.data() : (const void *)ghost_gids.data(), sb.data(), req.data(),
                     rb.data());
   send_idx.resize((size_t)tot);
   const long long lo = part[(size_t)cm.rank];
   for (long q = 0; q < tot; q++)
   {
      const long long l = req[(size_t)q] - lo;
      HDA_REQUIRE(l >= 0 && l < nloc, "peer requested a row this rank does not own");
      send_idx[(size_t)q] = (int)l;
   }
}

HaloPlan make_halo_plan(int nloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids)
{
   HaloPlan h;
   h.nloc   = nloc;
   h.nghost = (int)ghost_gids.size();
   std::vector<int> idx;
   halo_plan_host(nloc, part, ghost_gids, h.send_counts, h.recv_counts, idx);
   if (Comm::world().size == 1) return h;
   const long tot = (long)idx.size();
   if (idx.empty()) idx.push_back(0);
   h.send_total = (int)tot;
   h.send_idx.upload(idx.data(), idx.size());
   h.send_buf.alloc((size_t)std::max<long>(tot, 1));
   return h;
}

__global__ __launch_bounds__(256) void k_pack(int n, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ buf)
{
   const int q = blockIdx.x * 256 + threadIdx.x;
   if (q < n) buf[q] = x[idx[q]];
}

void halo_exchange(const HaloPlan &h, double *x_ext)
{
   Comm &cm = Comm::world();
   if (cm.size == 1 || h.send_counts.empty()) return; // no plan: operator is not partitioned (replicated tail)
   if (h.send_total) k_pack<<<ceil_div(h.send_total, 256), 256, 0, STREAM>>>(h.send_total, h.send_idx.data(), x_ext, (double *)h.send_buf.data());
   cm.exchange_dev(h.send_buf.data(), h.send_counts.data(), x_ext + h.nloc, h.recv_counts.data());
}

bool halo_active(const HaloPlan &h) { return Comm::world().size > 1 && !h.send_counts.empty(); }

static void halo_events(const HaloPlan &h)
{
   if (h.ev_packed) return;
   HDA_HIP(hipEventCreateWithFlags(&h.ev_packed, hipEventDisableTiming));
   HDA_HIP(hipEventCreateWithFlags(&h.ev_landed, hipEventDisableTiming));
}
void halo_pack(const HaloPlan &h, const double *x)
{
   halo_events(h);
   if (h.send_total) k_pack<<<ceil_div(h.send_total, 256), 256, 0, STREAM>>>(h.send_total, h.send_idx.data(), x, (double *)h.send_buf.data());
   HDA_HIP(hipEventRecord(h.ev_packed, STREAM));
}
void halo_transfer(const HaloPlan &h, double *x)
{
   Context &c = Context::get();
   HDA_HIP(hipStreamWaitEvent(c.comm_stream, h.ev_packed, 0));
   Comm::world().exchange_dev(h.send_buf.data(), h.send_counts.data(), x + h.nloc, h.recv_counts.data(), c.comm_stream);
   HDA_HIP(hipEventRecord(h.ev_landed, c.comm_stream));
}
void halo_wait(const HaloPlan &h) { HDA_HIP(hipStreamWaitEvent(STREAM, h.ev_landed, 0)); }

void gather_global(const DCsr &Aloc, const std::vector<long long> &part, const std::vector<long long> &ghost_gids, DCsr &G)
{
   Comm &cm = Comm::world();
   HDA_REQUIRE(part.back() < (1LL << 31), "replicated AMG setup needs global rows < 2^31");
   const int        nloc = Aloc.nrows;
   std::vector<int> rp   = Aloc.rowptr.to_host();
   std::vector<int> cj((size_t)std::max(Aloc.nnz, 1));
   std::vector<double> v((size_t)std::max(Aloc.nnz, 1));
   if (Aloc.nnz)
   {
      Aloc.col.download(cj.data(), (size_t)Aloc.nnz);
      Aloc.val.download(v.data(), (size_t)Aloc.nnz);
   }
   const long long lo       = part[(size_t)cm.rank];
   const int       ncol_loc = nloc; // square operator: owned columns == owned rows
   for (int k = 0; k < Aloc.nnz; k++)
      cj[(size_t)k] = (cj[(size_t)k] < ncol_loc) ? (int)(lo + cj[(size_t)k]) : (int)ghost_gids[(size_t)(cj[(size_t)k] - ncol_loc)];
   std::vector<int> lens((size_t)std::max(nloc, 1));
   for (int i = 0; i < nloc; i++) lens[(size_t)i] = rp[(size_t)i + 1] - rp[(size_t)i];
   std::vector<char> all_len, all_col, all_val;
   std::vector<long> c1, c2, c3;
   cm.allgatherv_bytes(lens.data(), 4L * nloc, all_len, c1);
   cm.allgatherv_bytes(cj.data(), 4L * Aloc.nnz, all_col, c2);
   cm.allgatherv_bytes(v.data(), 8L * Aloc.nnz, all_val, c3);
   const int        N = (int)part.back();
   long             nnz = 0;
   for (long b : c2) nnz += b / 4;
   std::vector<int> grp((size_t)N + 1, 0);
   const int       *L = (const int *)all_len.data();
   for (int i = 0; i < N; i++) grp[(size_t)i + 1] = grp[(size_t)i] + L[i];
   HDA_REQUIRE(grp[(size_t)N] == nnz, "gather_global: inconsistent lengths");
   G.nrows = G.ncols = N;
   G.nnz             = (int)nnz;
   G.rowptr.upload(grp.data(), (size_t)N + 1);
   G.col.upload((const int *)all_col.data(), (size_t)std::max<long>(nnz, 1));
   G.val.upload((const double *)all_val.data(), (size_t)std::max<long>(nnz, 1));
   sort_rows(G);
   Context::get().sync();
}

} // namespace hda
