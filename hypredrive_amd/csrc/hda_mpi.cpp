// hda_mpi.cpp -- joining the ranks of an MPI program through the communicator it hands to HYPREDRV_Create /
// HYPRE_IJMatrixCreate (reference: src/HYPREDRV.c:1014-1041 takes rank and size from that communicator; its multi-rank
// checks are plain `mpiexec -n N` launches of unmodified programs: tests/test_setmatrix_from_csr_mpi.c:145-190,
// examples/src/C_laplacian/CMakeLists.txt:76, cmake/HYPREDRV_Testing.cmake:938).
//
// The library has no link-time MPI dependency: the MPI entry points are looked up in the running process
// (dlsym(RTLD_DEFAULT)), i.e. they exist exactly when the application itself is an MPI program.  MPI_Comm is not
// ABI-portable, so the handle is interpreted only after MPI_Get_library_version has named an implementation of the
// MPICH ABI (MPICH, Intel MPI, MVAPICH, Cray MPICH: int handles, the constants below -- checked against <mpi.h> by
// tests/test_mpi_join.py where a header exists).  Any other MPI (Open MPI: pointer handles) is refused by name with the
// way out: build hypredrive_amd/csrc/hda_mpi_shim.c with the application's own mpicc and call HYPREDRV_AMD_CommInitMPI.
//
// What joining does: duplicate the communicator (the library's traffic never meets the application's), gather host
// name and PCI bus id of every rank's GPU, bind rank -> GPU by the rank's position on its host, then
//   * one physical GPU per rank  -> RCCL (the unique id travels by MPI_Bcast), the product transport;
//   * ranks sharing a GPU        -> the host-staged transport over MPI_Iallreduce / MPI_Ialltoallv (RCCL refuses two ranks
//                                   on one device; this is the single-GPU box of the test pool and oversubscribed nodes).
// No process is ever re-exec'd; a failure is an exception -> error bits on this rank, and HYPREDRV_SafeCallHandleError ends
// all ranks with MPI_Abort like the reference.  HDA_COMM_TIMEOUT_S (default 0 = wait forever, as MPI does) bounds every
// staged collective: a rank whose peers never arrive reports rank, operation and stage and aborts the job.
#include "hda_comm.h"
#include "hda_mpi_join.h"

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>

namespace hda {

namespace {
// ---- MPICH ABI (mpi.h of MPICH 3.x / 4.x and derivatives): handles are ints
typedef int MComm;
typedef int MType;
typedef int MOp;
typedef int MReq;
constexpr MType kMpiByte = 0x4c00010d, kMpiDouble = 0x4c00080b, kMpiLongLong = 0x4c000809;
constexpr MOp   kMpiMax = 0x58000001, kMpiSum = 0x58000003;
constexpr MComm kMpiCommNull = 0x04000000;
#define HDA_MPI_IN_PLACE ((void *)-1)
#define HDA_MPI_STATUS_IGNORE ((void *)1)

struct MpiApi {
   bool resolved = false, usable = false;
   std::string why; // why it is not usable
   int (*Initialized)(int *)                                                                   = nullptr;
   int (*Finalized)(int *)                                                                     = nullptr;
   int (*Get_library_version)(char *, int *)                                                   = nullptr;
   int (*Comm_rank)(MComm, int *)                                                              = nullptr;
   int (*Comm_size)(MComm, int *)                                                              = nullptr;
   int (*Comm_dup)(MComm, MComm *)                                                             = nullptr;
   int (*Comm_free)(MComm *)                                                                   = nullptr;
   int (*Bcast)(void *, int, MType, int, MComm)                                                = nullptr;
   int (*Allgather)(const void *, int, MType, void *, int, MType, MComm)                       = nullptr;
   int (*Iallreduce)(const void *, void *, int, MType, MOp, MComm, MReq *)                     = nullptr;
   int (*Ialltoallv)(const void *, const int *, const int *, MType, void *, const int *, const int *, MType, MComm, MReq *) = nullptr;
   int (*Test)(MReq *, int *, void *)                                                          = nullptr;
   int (*Wait)(MReq *, void *)                                                                 = nullptr;
   int (*Abort)(MComm, int)                                                                    = nullptr;
};

MpiApi &api()
{
   static MpiApi a;
   if (a.resolved) return a;
   a.resolved = true;
   *(void **)(&a.Initialized) = dlsym(RTLD_DEFAULT, "MPI_Initialized");
   if (!a.Initialized) { a.why = "the process has no MPI (MPI_Initialized not found)"; return a; }
#define HDA_MSYM(f)                                                             \
   *(void **)(&a.f) = dlsym(RTLD_DEFAULT, "MPI_" #f);                           \
   if (!a.f) { a.why = "MPI symbol missing in the process: MPI_" #f; return a; }
   HDA_MSYM(Finalized) HDA_MSYM(Get_library_version) HDA_MSYM(Comm_rank) HDA_MSYM(Comm_size) HDA_MSYM(Comm_dup) HDA_MSYM(Comm_free)
   HDA_MSYM(Bcast) HDA_MSYM(Allgather) HDA_MSYM(Iallreduce) HDA_MSYM(Ialltoallv) HDA_MSYM(Test) HDA_MSYM(Wait) HDA_MSYM(Abort)
#undef HDA_MSYM
   std::vector<char> ver(8192 + 64, 0); // MPI_MAX_LIBRARY_VERSION_STRING of the MPICH family
   int               len = 0;
   a.Get_library_version(ver.data(), &len);
   const std::string v(ver.data());
   const char *family[] = {"MPICH", "Intel(R) MPI", "MVAPICH", "CRAY MPICH"};
   for (const char *f : family)
      if (v.find(f) != std::string::npos) a.usable = true;
   if (!a.usable)
      a.why = "the process's MPI (\"" + v.substr(0, v.find('\n')) + "\") is not of the MPICH ABI family, so its MPI_Comm cannot be read by a library "
              "built without its header: compile hypredrive_amd/csrc/hda_mpi_shim.c with the application's mpicc and call "
              "HYPREDRV_AMD_CommInitMPI(comm) before HYPREDRV_Create (INTEGRATION.md section 1)";
   return a;
}

bool mpi_running()
{
   MpiApi &a = api();
   if (!a.Initialized) return false;
   int ini = 0, fin = 0;
   a.Initialized(&ini);
   if (a.Finalized) a.Finalized(&fin);
   return ini && !fin;
}

// ---- state of a joined process
MComm g_dup        = kMpiCommNull; // the library's duplicate
bool  g_joined     = false;
int   g_rank       = 0;


// wait for a nonblocking collective; with HDA_COMM_TIMEOUT_S a rank whose peers never arrive says so and ends the job
void wait_request(MReq *rq, const char *what)
{
   MpiApi &a = api();
   const double lim = wait_limit_s();
   if (lim <= 0.0)
   {
      if (a.Wait(rq, HDA_MPI_STATUS_IGNORE) != 0) throw Error(std::string("MPI transport: MPI_Wait failed in ") + what);
      return;
   }
   const auto t0 = std::chrono::steady_clock::now();
   for (long spin = 0;; spin++)
   {
      int done = 0;
      if (a.Test(rq, &done, HDA_MPI_STATUS_IGNORE) != 0) throw Error(std::string("MPI transport: MPI_Test failed in ") + what);
      if (done) return;
      if (spin > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > lim)
      {
         fprintf(stderr, "[hypredrv_amd] rank %d: %s did not complete within %.1f s (HDA_COMM_TIMEOUT_S) during %s: a peer rank has failed or "
                         "left the collective sequence; aborting the job\n", g_rank, what, lim, current_stage());
         fflush(nullptr);
         usleep(300000); // (hydra drops what a rank wrote just before its MPI_Abort)
         a.Abort(g_dup, 86);
         _exit(86);
      }
   }
}

int cb_allreduce(void *buf, long count, int dtype, int op)
{
   MpiApi &a = api();
   if (count > 0x7fffffffL) return 2;
   MReq rq = 0;
   if (a.Iallreduce(HDA_MPI_IN_PLACE, buf, (int)count, dtype == 0 ? kMpiDouble : kMpiLongLong, op == 0 ? kMpiSum : kMpiMax, g_dup, &rq) != 0) return 1;
   wait_request(&rq, dtype == 0 ? "all-reduce (doubles)" : "all-reduce (integers)");
   return 0;
}

int cb_alltoallv(const void *send, const long *sb, void *recv, const long *rb)
{
   MpiApi &a = api();
   int np = 0;
   a.Comm_size(g_dup, &np);
   // counts in bytes when every block and offset fits an int, else in 8-byte words (every large message of the library is whole words)
   long st = 0, rt = 0;
   bool words = false;
   for (int p = 0; p < np; p++) { st += sb[p]; rt += rb[p]; }
   if (st > 0x7fffffffL || rt > 0x7fffffffL) words = true;
   std::vector<int> sc((size_t)np), sd((size_t)np), rc((size_t)np), rd((size_t)np);
   long so = 0, ro = 0;
   for (int p = 0; p < np; p++)
   {
      if (words && ((sb[p] | rb[p]) & 7)) return 3;
      const long u = words ? 8 : 1;
      if (sb[p] / u > 0x7fffffffL || rb[p] / u > 0x7fffffffL || so / u > 0x7fffffffL || ro / u > 0x7fffffffL) return 4;
      sc[(size_t)p] = (int)(sb[p] / u); sd[(size_t)p] = (int)(so / u);
      rc[(size_t)p] = (int)(rb[p] / u); rd[(size_t)p] = (int)(ro / u);
      so += sb[p]; ro += rb[p];
   }
   MReq rq = 0;
   const MType t = words ? kMpiDouble : kMpiByte;
   if (a.Ialltoallv(send, sc.data(), sd.data(), t, recv, rc.data(), rd.data(), t, g_dup, &rq) != 0) return 1;
   wait_request(&rq, "all-to-all");
   return 0;
}

std::string pci_bus_id(int dev)
{
   char b[64] = {0};
   if (hipDeviceGetPCIBusId(b, (int)sizeof(b), dev) != hipSuccess) { (void)hipGetLastError(); snprintf(b, sizeof(b), "dev%d", dev); }
   return b;
}
} // namespace

bool mpi_joined() { return g_joined; }

bool mpi_comm_size(int comm, int *rank, int *size)
{
   if (!mpi_running() || !api().usable) return false;
   return api().Comm_rank(comm, rank) == 0 && api().Comm_size(comm, size) == 0;
}

// Called by every entry point that receives a communicator (HYPREDRV_Create, HYPRE_IJMatrixCreate, HYPRE_IJVectorCreate, the readers).
void mpi_autojoin(int comm)
{
   if (g_joined || Comm::explicitly_joined() || in_thread_rank()) return;
   if (getenv("HDA_MPI_JOIN") && !strcmp(getenv("HDA_MPI_JOIN"), "0")) return;
   if (!mpi_running()) return;
   MpiApi &a = api();
   if (!a.usable)
   { // an MPI program on several ranks whose handle cannot be read: say so once instead of running N unconnected solves
      static bool told = false;
      if (!told) fprintf(stderr, "[hypredrv_amd] not joining the MPI ranks: %s\n", a.why.c_str());
      told = true;
      return;
   }
   int rank = 0, size = 1;
   if (a.Comm_rank(comm, &rank) != 0 || a.Comm_size(comm, &size) != 0) throw Error("MPI_Comm_rank/size failed on the communicator given to the library");
   if (size <= 1) return; // MPI_COMM_SELF or a one-rank job: nothing to join
   if (a.Comm_dup(comm, &g_dup) != 0) throw Error("MPI_Comm_dup failed");
   g_rank = rank;

   // who shares a host, who shares a GPU
   struct Card { char host[64]; char bus[32]; };
   std::vector<Card> all((size_t)size);
   Card mine;
   memset(&mine, 0, sizeof(mine));
   gethostname(mine.host, sizeof(mine.host) - 1);
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
   // (no device: the ranks are still joined -- over MPI -- so that the first device call fails on every rank alike, and so
   // that the host half of the partition code can be driven on a CPU-only machine: tests/test_mpi_join.py)
   // first pass: host names -> position of this rank on its host -> device
   a.Allgather(&mine, (int)sizeof(Card), kMpiByte, all.data(), (int)sizeof(Card), kMpiByte, g_dup);
   int local = 0;
   for (int p = 0; p < rank; p++)
      if (!strncmp(all[(size_t)p].host, mine.host, sizeof(mine.host))) local++;
   const int dev = ndev > 0 ? local % ndev : -1;
   if (dev >= 0 && hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); throw Error("hipSetDevice failed on an MPI rank"); }
   snprintf(mine.bus, sizeof(mine.bus), "%s", dev >= 0 ? pci_bus_id(dev).c_str() : "none");
   a.Allgather(&mine, (int)sizeof(Card), kMpiByte, all.data(), (int)sizeof(Card), kMpiByte, g_dup);
   bool shared = false;
   for (int p = 0; p < size && !shared; p++)
      for (int q = p + 1; q < size; q++)
         if (!strncmp(all[(size_t)p].host, all[(size_t)q].host, 64) && !strncmp(all[(size_t)p].bus, all[(size_t)q].bus, 32)) { shared = true; break; }
   const char *force = getenv("HDA_MPI_TRANSPORT"); // "staged" keeps the data path on MPI even with one GPU per rank
   if (force && !strcmp(force, "staged")) shared = true;

   if (!shared)
   {
      char uid[128];
      memset(uid, 0, sizeof(uid));
      if (rank == 0) rccl_get_unique_id(uid);
      a.Bcast(uid, 128, kMpiByte, 0, g_dup);
      Comm::set_world(make_rccl_comm(rank, size, uid));
   }
   else
      Comm::set_world(make_callback_comm(rank, size, cb_allreduce, cb_alltoallv));
   g_joined = true;
   if (verbose() || getenv("HDA_MPI_VERBOSE"))
      fprintf(stderr, "[hypredrv_amd] rank %d of %d joined through the MPI communicator: device %d (%s), transport %s\n", rank, size, dev, mine.bus,
              Comm::world().name());
}

// HYPREDRV_Finalize / HYPRE_Finalize: give the duplicate back while MPI is still alive (the caller finalizes MPI afterwards)
void mpi_leave()
{
   if (!g_joined) return;
   Comm::set_world(make_self_comm());
   if (mpi_running() && g_dup != kMpiCommNull) api().Comm_free(&g_dup);
   g_dup    = kMpiCommNull;
   g_joined = false;
}

// the reference ends a failed HYPREDRV_SAFE_CALL with MPI_Abort(comm, code): the peers of a failed rank must not wait for it
bool mpi_abort(int comm, int code)
{
   if (!mpi_running() || !api().usable) return false;
   fflush(nullptr);
   usleep(300000); // (hydra drops what a rank wrote just before its MPI_Abort)
   api().Abort(g_joined ? g_dup : comm, code);
   return true;
}

} // namespace hda
