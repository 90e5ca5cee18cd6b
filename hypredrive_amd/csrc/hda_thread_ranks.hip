// hda_thread_ranks.hip -- test seam: the ranks of a row partition as THREADS of one process.
//
// Why it exists: BASELINE config 3 is a 2x2x2 rank grid (reference examples/src/C_laplacian/laplacian.c:561-582, `-P 2 2 2`),
// whose coarse levels have edge and corner neighbours (7 peers) that no 1xPxQ layout produces -- and a GPU box admits at most
// six processes on its card, so eight rank PROCESSES sharing the one GPU cannot be rehearsed there.  The library's state is
// process-global (hda_common.h); a thread that enters this seam (enter_thread_rank) gets a private copy of every piece of it
// (context + stream, allocator, communicator, API and error state) until it leaves, so eight ranks can run as eight threads of
// one process over the in-process transport of hda_testranks_comm.hip
// (ThreadComm: host-staged messages, generation barrier).  Each thread drives the PUBLIC API exactly like one rank of
// laplacian.c:331-468 does: Initialize, Create, InputArgsParse, (generator), ResetInitialGuess + LinearSolverCreate + Setup +
// Apply + Destroy, getters.  Nothing here is on the product path: this file and the transport are built into libhypredrv_amd_testranks.so, not the product library.
#include "HYPREDRV.h"
#include "hda_testranks.h"
#include "hypredrv_amd.h"

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace hda;




namespace {
struct RankOut {
   int         iters = 0, converged = 0, vcycles = 0, part_levels = 0;
   double      final_rel = 0, l2 = 0, l1 = 0, linf = 0;
   Comm::Stats comm;
   std::string err;
};

void rank_body(int rank, int nranks, const std::shared_ptr<void> &world, const int n[3], const int P[3], const char *yaml, int nsolves, double *x_out,
               const long long *row_start, RankOut &out)
{
   enter_thread_rank(); // this thread is one rank: context, allocator, communicator and API state of its own (hda_common.h)
   try
   {
      Comm::set_world(make_thread_comm(rank, world));
      auto chk = [&](uint32_t code, const char *what) {
         if (code) throw Error(std::string(what) + ": " + HYPREDRV_AMD_LastErrorMessage());
      };
      chk(HYPREDRV_Initialize(), "HYPREDRV_Initialize");
      HYPREDRV_t h = nullptr;
      chk(HYPREDRV_Create(MPI_COMM_WORLD, &h), "HYPREDRV_Create");
      chk(HYPREDRV_SetLibraryMode(h), "HYPREDRV_SetLibraryMode");
      std::string y(yaml);
      char       *argv[1] = {y.data()};
      chk(HYPREDRV_InputArgsParse(1, argv, h), "HYPREDRV_InputArgsParse");
      const double c[3] = {1.0, 1.0, 1.0};
      chk(HYPREDRV_AMD_LinearSystemSetLaplacian7pt(h, n, P, c), "HYPREDRV_AMD_LinearSystemSetLaplacian7pt");
      for (int s = 0; s < std::max(nsolves, 1); s++)
      {
         chk(HYPREDRV_LinearSystemResetInitialGuess(h), "HYPREDRV_LinearSystemResetInitialGuess");
         chk(HYPREDRV_LinearSolverCreate(h), "HYPREDRV_LinearSolverCreate");
         chk(HYPREDRV_LinearSolverSetup(h), "HYPREDRV_LinearSolverSetup");
         Comm::world().stats = Comm::Stats(); // rank-to-rank operations of the solve alone
         chk(HYPREDRV_LinearSolverApply(h), "HYPREDRV_LinearSolverApply");
         out.comm        = Comm::world().stats;
         out.vcycles     = hda_last_precond_calls();
         out.part_levels = hda_amd_partitioned_levels(h);
         chk(HYPREDRV_LinearSolverGetNumIter(h, &out.iters), "HYPREDRV_LinearSolverGetNumIter");
         chk(HYPREDRV_LinearSolverGetConverged(h, &out.converged), "HYPREDRV_LinearSolverGetConverged");
         chk(HYPREDRV_LinearSolverGetFinalRelativeResidualNorm(h, &out.final_rel), "HYPREDRV_LinearSolverGetFinalRelativeResidualNorm");
         chk(HYPREDRV_LinearSolverDestroy(h), "HYPREDRV_LinearSolverDestroy");
      }
      chk(HYPREDRV_LinearSystemGetSolutionNorm(h, "L2", &out.l2), "HYPREDRV_LinearSystemGetSolutionNorm");
      chk(HYPREDRV_LinearSystemGetSolutionNorm(h, "L1", &out.l1), "HYPREDRV_LinearSystemGetSolutionNorm");
      chk(HYPREDRV_LinearSystemGetSolutionNorm(h, "Linf", &out.linf), "HYPREDRV_LinearSystemGetSolutionNorm");
      if (x_out)
      {
         HYPRE_Complex *xv  = nullptr;
         HYPRE_BigInt   len = 0;
         chk(HYPREDRV_LinearSystemGetSolutionValues(h, &xv), "HYPREDRV_LinearSystemGetSolutionValues");
         chk(HYPREDRV_LinearSystemGetSolutionLength(h, &len), "HYPREDRV_LinearSystemGetSolutionLength");
         if (len != row_start[rank + 1] - row_start[rank]) throw Error("thread ranks: local solution length differs from the generator's block");
         memcpy(x_out + row_start[rank], xv, sizeof(double) * (size_t)len);
      }
      chk(HYPREDRV_Destroy(&h), "HYPREDRV_Destroy");
      chk(HYPREDRV_Finalize(), "HYPREDRV_Finalize");
   }
   catch (const std::exception &e)
   {
      out.err = std::string("rank ") + std::to_string(rank) + ": " + e.what();
      thread_world_fail(world); // release the ranks waiting for this one
   }
   leave_thread_rank();
   (void)nranks;
}
} // namespace

// AMG-Krylov solve of the generator's 7-pt Laplacian (global grid n, rank grid P, P0*P1*P2 = nranks) on `nranks` thread ranks.
// out16: iters, converged, final_rel, ||x||_2, ||x||_1, ||x||_inf, all-reduces / exchanges / overlapped exchanges / doubles
// all-reduced / doubles exchanged of rank 0's last solve, V-cycles, partitioned levels, max |iters_r - iters_0|, ranks, 0.
// x_global (may be null): the solution in the generator's block numbering (rank blocks in rank order), length n0*n1*n2.
extern "C" int hda_thread_ranks_lap7(int nranks, const int n[3], const int P[3], const char *yaml, int nsolves, double out16[16], double *x_global,
                                     char *errbuf, int errlen)
{
   if (nranks < 1 || P[0] * P[1] * P[2] != nranks) return 1;
   // block sizes of the generator (laplacian.c:561-582): rows of rank r = product of its three extents
   std::vector<long long> row_start((size_t)nranks + 1, 0);
   for (int r = 0; r < nranks; r++)
   {
      const int pc[3] = {r / (P[1] * P[2]), (r / P[2]) % P[1], r % P[2]};
      long long rows  = 1;
      for (int d = 0; d < 3; d++)
      {
         const int size = n[d] / P[d], rest = n[d] - size * P[d];
         rows *= size + (pc[d] < rest ? 1 : 0);
      }
      row_start[(size_t)r + 1] = row_start[(size_t)r] + rows;
   }
   std::shared_ptr<void>    world = make_thread_world(nranks);
   std::vector<RankOut>     outs((size_t)nranks);
   std::vector<std::thread> th;
   for (int r = 0; r < nranks; r++) th.emplace_back(rank_body, r, nranks, std::cref(world), n, P, yaml, nsolves, x_global, row_start.data(), std::ref(outs[(size_t)r]));
   for (auto &t : th) t.join();
   std::string err;
   for (auto &o : outs)
      if (!o.err.empty()) err += o.err + "\n";
   if (!err.empty())
   {
      if (errbuf && errlen > 0) { strncpy(errbuf, err.c_str(), (size_t)errlen - 1); errbuf[errlen - 1] = 0; }
      return 2;
   }
   const RankOut &o = outs[0];
   int            spread = 0;
   for (auto &q : outs) spread = std::max(spread, std::abs(q.iters - o.iters));
   const double v[16] = {(double)o.iters, (double)o.converged, o.final_rel, o.l2, o.l1, o.linf, (double)o.comm.allreduce, (double)o.comm.exchange,
                         (double)o.comm.overlapped, (double)o.comm.allreduce_doubles, (double)o.comm.exchange_doubles, (double)o.vcycles,
                         (double)o.part_levels, (double)spread, (double)nranks, 0.0};
   memcpy(out16, v, sizeof(v));
   return 0;
}

// ---- the same thing with the caller's own threads: a host thread joins an in-process world as one rank and then drives the public API
// itself (tests: Python threads, each handing over its row block of an arbitrary CSR matrix).  world: hda_thread_world_create's handle.
extern "C" void *hda_thread_world_create(int nranks)
{
   if (nranks < 1) return nullptr;
   return new std::shared_ptr<void>(make_thread_world(nranks));
}
extern "C" int hda_thread_world_join(void *world, int rank)
{
   if (!world) return 1;
   try
   {
      enter_thread_rank();
      Comm::set_world(make_thread_comm(rank, *(std::shared_ptr<void> *)world));
   }
   catch (const std::exception &)
   {
      return 2;
   }
   return 0;
}
// failed != 0: this rank gives up -- the ranks waiting for it in a collective are released with an error instead of hanging
extern "C" int hda_thread_world_leave(void *world, int failed)
{
   if (world && failed) thread_world_fail(*(std::shared_ptr<void> *)world);
   try
   {
      leave_thread_rank(); // (blocks this rank still holds move to the process's allocator: whoever releases them later finds them)
   }
   catch (const std::exception &)
   {
      return 2;
   }
   return 0;
}
extern "C" void hda_thread_world_destroy(void *world) { delete (std::shared_ptr<void> *)world; }

// ---- self tests of the seam itself (tests/test_gpu_hypredrv.py)
// what = 0: two thread ranks enter DIFFERENT collectives (an all-reduce against an all-to-all).  Both must come back with an error that
//           names the disagreement -- before round 4's end this read a stale pointer of the other rank and crashed the process.
// what = 2: the same collective with send and receive counts that disagree: an error naming the two ranks and the byte counts.
// what = 1: the device allocator when the device is full of OTHER rank threads' cached blocks: a second rank allocates `cache_gb` and
//           releases it (the block stays cached in that rank's pool), then this rank asks for more than the driver has left -- the
//           request must be served by giving the other rank's cache back, not fail.
// Returns 0 when the behaviour is as described, 1 otherwise; the ranks' messages in errbuf.
extern "C" int hda_testranks_selftest(int what, double cache_gb, char *errbuf, int errlen)
{
   std::string report;
   int         rc = 1;
   if (what == 0 || what == 2)
   {
      std::shared_ptr<void>    world = make_thread_world(2);
      std::string              msg[2];
      std::vector<std::thread> th;
      for (int r = 0; r < 2; r++)
         th.emplace_back([&, r] {
            enter_thread_rank();
            try
            {
               Comm::set_world(make_thread_comm(r, world));
               long long v[2] = {1, 2};
               long      eight[2] = {8, 8};
               long long out[2]   = {0, 0};
               long      sixteen[2] = {16, 16};
               long long out4[4]    = {0, 0, 0, 0};
               if (what == 2) // same collective, but rank 1 expects 16 bytes from everybody where 8 are sent
               {
                  if (r == 0) Comm::world().alltoallv_host(v, eight, out, eight);
                  else Comm::world().alltoallv_host(v, eight, out4, sixteen);
               }
               else if (r == 0) Comm::world().allreduce_host(v, 2, 0);
               else Comm::world().alltoallv_host(v, eight, out, eight);
               msg[r] = "no error";
            }
            catch (const std::exception &e)
            {
               msg[r] = e.what();
               thread_world_fail(world);
            }
            leave_thread_rank();
         });
      for (auto &t : th) t.join();
      report = "rank 0: " + msg[0] + "\nrank 1: " + msg[1];
      const char *needle = what == 2 ? "which expects 16" : "different collectives";
      const bool  named  = msg[0].find(needle) != std::string::npos || msg[1].find(needle) != std::string::npos;
      rc = (named && msg[0] != "no error" && msg[1] != "no error") ? 0 : 1;
   }
   else if (what == 1)
   {
      std::mutex              mu;
      std::condition_variable cv;
      int                     stage = 0; // 1: the other rank's block is cached; 2: this rank is done
      std::string             other;
      std::thread             t([&] {
         enter_thread_rank();
         try
         {
            {
               DArray<char> big((size_t)(cache_gb * (double)(1ull << 30)));
               HDA_HIP(hipMemsetAsync(big.data(), 0, 4096, Context::get().stream));
               Context::get().sync();
            } // released: cached in this rank's pool
            other = "cached " + std::to_string(hda_memory_cached() / (double)(1ull << 30)) + " GB";
         }
         catch (const std::exception &e)
         {
            other = std::string("error: ") + e.what();
         }
         {
            std::unique_lock<std::mutex> lk(mu);
            stage = 1;
            cv.notify_all();
            cv.wait(lk, [&] { return stage == 2; });
         }
         leave_thread_rank();
      });
      {
         std::unique_lock<std::mutex> lk(mu);
         cv.wait(lk, [&] { return stage == 1; });
      }
      std::string mine;
      try
      {
         (void)hda_memory_trim(); // nothing cached on this thread: what is short can only come from the other rank's pool
         size_t freeb = 0, totalb = 0;
         HDA_HIP(hipMemGetInfo(&freeb, &totalb));
         const size_t want = freeb + (size_t)(0.5 * cache_gb * (double)(1ull << 30)); // more than the driver has, less than free + the other rank's cache
         {
            DArray<char> big(want);
            HDA_HIP(hipMemsetAsync(big.data(), 0, 4096, Context::get().stream));
            Context::get().sync();
         }
         (void)hda_memory_trim();
         mine = "served " + std::to_string((double)want / (double)(1ull << 30)) + " GB with " + std::to_string((double)freeb / (double)(1ull << 30)) + " GB free";
         rc   = other.rfind("cached", 0) == 0 ? 0 : 1;
      }
      catch (const std::exception &e)
      {
         mine = std::string("error: ") + e.what();
      }
      {
         std::lock_guard<std::mutex> lk(mu);
         stage = 2;
         cv.notify_all();
      }
      t.join();
      report = "other rank: " + other + "\nthis rank: " + mine;
   }
   if (errbuf && errlen > 0) { strncpy(errbuf, report.c_str(), (size_t)errlen - 1); errbuf[errlen - 1] = 0; }
   return rc;
}
