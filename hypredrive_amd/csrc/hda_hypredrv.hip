// hda_hypredrv.hip -- HYPREDRV_* public API (include/HYPREDRV.h) for the AMG-Krylov path.
//
// Mirrors the call structure of the reference so that the hot path crosses the same seams:
//   HYPREDRV_LinearSolverSetup  (src/HYPREDRV.c:3001-3119)  -> HYPRE_ParCSR{PCG,GMRES}Setup
//       -> PreconSetupDispatch ("prec" timer, src/internal/solver.c:268-311) -> HYPRE_BoomerAMGSetup
//   HYPREDRV_LinearSolverApply  (src/HYPREDRV.c:3126-3338, src/internal/solver.c:627-693)
//       -> r0 (untimed) -> "solve" timer { HYPRE_ParCSR{PCG,GMRES}Solve -> PreconSolveDispatch
//          -> HYPRE_BoomerAMGSolve } -> true relative residual (untimed)
// Error handling: sticky process-global bit field, reset at the start of each lifecycle call
// (reference include/internal/error.h:16-48, src/HYPREDRV.c:2797,2901,2965,3005,3129).
#include "../../include/HYPREDRV.h"

#include "hda_hypre.h"
#include "hda_yaml.h"
#include "hda_mpi_join.h"

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <climits>
#include <cmath>
#include <csignal>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <sstream>

using namespace hda;

// ------------------------------------------------------------------- errors

// (process-global, as in the reference: the contract of HYPREDRV.h:66-70 is one thread AT A TIME, any thread may make the next call;
// a thread rank of the test seam has its own copy, hda_common.h RankState)
#define g_err (RankState<uint32_t, 4>::get())
#define g_errmsg (RankState<std::string, 4>::get())
#define g_initialized (RankState<bool, 4>::get())

static uint32_t err_set(uint32_t e, const std::string &msg = "")
{
   g_err |= e;
   if (!msg.empty())
   {
      if (!g_errmsg.empty()) g_errmsg += "\n";
      g_errmsg += msg;
   }
   return g_err;
}
static void err_reset()
{
   g_err = 0;
   g_errmsg.clear();
   HYPRE_ClearAllErrors();
}
// The code-level half of the reference's error state (include/internal/error.h:56-62, src/internal/error.c:920-984): not part of the
// public header, but the reference's own unit tests (tests/test_setmatrix_from_csr.c:255-459, tests/test_helpers.h) call it between
// their negative cases, so a drop-in boundary exports it.  Same sticky bits as every HYPREDRV_* call returns.
extern "C" void     hypredrv_ErrorCodeSet(uint32_t code) { err_set(code); }
extern "C" uint32_t hypredrv_ErrorCodeGet(void) { return g_err; }
extern "C" bool     hypredrv_ErrorCodeActive(void) { return g_err != 0; }
extern "C" void     hypredrv_ErrorCodeReset(uint32_t bits) { g_err &= ~bits; }
extern "C" void     hypredrv_ErrorCodeResetAll(void) { g_err = 0; }
extern "C" void     hypredrv_ErrorStateReset(void) { err_reset(); }

// hypredrv_HypreConsumeErrors (reference src/internal/utils.c:33-79): HYPRE_ERROR_CONV and
// HYPRE_ERROR_ARG are soft, anything else becomes ERROR_HYPRE_INTERNAL
static void consume_hypre_errors()
{
   const int h = HYPRE_GetError();
   if (h & ~(HYPRE_ERROR_CONV | HYPRE_ERROR_ARG)) err_set(ERR_HYPRE_INTERNAL, hypre_last_error());
   HYPRE_ClearAllErrors();
}

extern "C" const char *HYPREDRV_AMD_LastErrorMessage(void) { return g_errmsg.c_str(); }

extern "C" void HYPREDRV_ErrorCodeDescribe(uint32_t code)
{
   static const struct { uint32_t bit; const char *text; } table[] = {
      {ERR_YAML_INVALID_INDENT, "YAML: invalid indentation"}, {ERR_YAML_INVALID_BASE_INDENT, "YAML: invalid base indentation"},
      {ERR_YAML_INCONSISTENT_INDENT, "YAML: inconsistent indentation"}, {ERR_YAML_INVALID_DIVISOR, "YAML: missing ':' divisor"},
      {ERR_YAML_TREE_NULL, "YAML: empty tree"}, {ERR_YAML_TREE_INVALID, "YAML: invalid tree"},
      {ERR_YAML_MIXED_INDENT, "YAML: tabs mixed with spaces"}, {ERR_YAML_INVALID_INDENT_JUMP, "YAML: indentation jump"},
      {ERR_INVALID_KEY, "invalid key"}, {ERR_INVALID_VAL, "invalid value"}, {ERR_UNEXPECTED_VAL, "unexpected value"},
      {ERR_MAYBE_INVALID_VAL, "possibly invalid value"}, {ERR_MISSING_KEY, "missing key"}, {ERR_EXTRA_KEY, "extra key"},
      {ERR_MISSING_SOLVER, "missing solver"}, {ERR_MISSING_PRECON, "missing preconditioner"}, {ERR_MISSING_DOFMAP, "missing dofmap"},
      {ERR_INVALID_SOLVER, "invalid solver"}, {ERR_INVALID_PRECON, "invalid preconditioner"}, {ERR_FILE_NOT_FOUND, "file not found"},
      {ERR_FILE_UNEXPECTED_ENTRY, "unexpected entry in file"}, {ERR_UNKNOWN_HYPREDRV_OBJ, "unknown HYPREDRV object"},
      {ERR_HYPREDRV_NOT_INITIALIZED, "HYPREDRV is not initialized"}, {ERR_UNKNOWN_TIMING, "unknown timing annotation"},
      {ERR_HYPRE_INTERNAL, "error inside the solver backend"}, {ERR_MISSING_LIB, "feature not available in this build"},
      {ERR_ALLOCATION, "allocation failure"}, {ERR_OUT_OF_BOUNDS, "index out of bounds"}, {ERR_UNKNOWN, "unknown error"}};
   if (!code) return;
   fprintf(stderr, "HYPREDRIVE Failure!!!\n");
   for (auto &t : table)
      if (code & t.bit) fprintf(stderr, "  --> %s\n", t.text);
   if (!g_errmsg.empty()) fprintf(stderr, "%s\n", g_errmsg.c_str());
}
extern "C" void HYPREDRV_ErrorCodeClear(void) { err_reset(); }
extern "C" uint32_t HYPREDRV_ErrorInvalidValue(const char *message) { return err_set(ERR_INVALID_VAL, message ? message : ""); }
extern "C" void HYPREDRV_SafeCallHandleError(uint32_t code, MPI_Comm comm, const char *file, int line, const char *func)
{
   if (!code) return;
   fprintf(stderr, "At %s:%d in %s():\n", file, line, func);
   HYPREDRV_ErrorCodeDescribe(code);
   const char *dbg = getenv("HYPREDRV_DEBUG");
   if (dbg && !strcmp(dbg, "1")) raise(SIGTRAP);
   fflush(nullptr);
   const int status = (int)(code & 0x7F) ? (int)(code & 0x7F) : 1;
   mpi_abort((int)comm, status); // the reference calls MPI_Abort (include/HYPREDRV_utils.h:50-80): the peers of a failed rank end with it
   _exit(status);
}

// -------------------------------------------------------------------- stats

namespace {
using clk = std::chrono::steady_clock;
struct StatEntry {
   double      build = 0.0, prec = 0.0, solve = 0.0, r0 = 0.0, rr = 0.0;
   int         iters = 0;
   bool        has_solve = false;
   std::string path; // "timestep.newton.system" when level annotations are active (reference stats.c:232-307)
};
// hierarchical annotations (reference src/internal/stats.c:953-1122, include/HYPREDRV.h:2021-2079)
constexpr int kStatsMaxLevels = 10;
struct LevelEntry {
   int id, solve_start, solve_end; // entry-index range [start, end) covered by the region
};
struct Stats {
   std::vector<StatEntry>                 entries{StatEntry()};
   std::map<std::string, clk::time_point> open;
   double                                 pending_build = 0.0;
   int                                    ls_counter = -1;
   bool                                   system_open = false;
   int                                    use_millisec = 0;
   // level annotations
   std::string             level_name[kStatsMaxLevels];
   bool                    level_open[kStatsMaxLevels] = {};
   int                     level_current_id[kStatsMaxLevels] = {};
   int                     level_solve_start[kStatsMaxLevels] = {};
   std::vector<LevelEntry> level_entries[kStatsMaxLevels];
   int                     systems_solved = 0; // flat 1-based counter of linear systems (leaf of the path)
   bool                    new_system = true;  // a matrix was set since the last solve
   int  solved_entries() const
   {
      int c = 0;
      for (const StatEntry &e : entries) c += e.has_solve ? 1 : 0;
      return c;
   }
   StatEntry &cur() { return entries.back(); }
   void       next_entry_if_used()
   {
      if (cur().has_solve) entries.emplace_back();
   }
};
} // namespace

struct PreconCookie { // the void* the Krylov solver hands back (reference include/internal/precon.h:84-95)
   struct hypredrv_struct *self;
};

struct hypredrv_struct {
   MPI_Comm       comm = MPI_COMM_WORLD;
   int            mypid = 0, nprocs = 1;
   bool           lib_mode = false;
   std::string    name;
   InputArgs      args;
   YNode          tree;
   HYPRE_IJMatrix mat_A = nullptr, mat_M = nullptr;
   bool           owns_M = false; // the preconditioning matrix was read from linear_system.precmat_filename
   HYPRE_IJVector vec_b = nullptr, vec_x = nullptr, vec_x0 = nullptr, vec_xref = nullptr;
   bool           owns_A = false, owns_b = false, owns_x = false, owns_x0 = false;
   HYPRE_Solver   solver = nullptr, precon = nullptr;
   bool           precon_is_setup = false;
   PreconCookie   cookie{nullptr};
   Stats          stats;
   std::vector<int> dofmap; // function / field label of every locally owned unknown
   std::vector<HYPRE_Solver> precon_aux; // component solvers owned together with an MGR preconditioner (coarse, F-relaxation)
   std::vector<HYPRE_IJVector> state; // borrowed time-level vectors (host resident), logical index i = state[(state_first + i) % n]
   int            state_first = 0;
   int            current_system_index = -1;
   // exact null-space modes (HYPREDRV_LinearSystemSetNullSpace): orthonormalised, component-major, device resident,
   // with the row range they were built for
   DArray<double> ns_modes;
   int            num_ns = 0, ns_nloc = 0;
   long long      ns_lower = 0, ns_upper = -1;
   int            last_iters = 0, last_converged = 0;
   double         last_rel = 0.0, last_setup_s = 0.0, last_solve_s = 0.0;
   // Scaling_context (reference include/internal/scaling.h:39-51): what of the caller's system is scaled right now
   struct ScalingCtx {
      bool           enabled = false, matrices_are_scaled = false, rhs_is_scaled = false, x_is_scaled = false;
      int            type = 0;
      double         scalar_factor = 1.0;       // rhs_l2
      DArray<double> scaling, inverse_scaling;  // dofmap types: one weight per owned row, and its reciprocal
      bool           is_applied() const { return matrices_are_scaled || rhs_is_scaled || x_is_scaled; }
   } scal;
};

#define CHECK_INIT_OBJ(h)                                          \
   if (!g_initialized) return err_set(ERR_HYPREDRV_NOT_INITIALIZED); \
   if (!(h)) return err_set(ERR_UNKNOWN_HYPREDRV_OBJ)

#define API_TRY try {
#define API_CATCH                                                        \
   }                                                                     \
   catch (const std::exception &e) { err_set(ERR_HYPRE_INTERNAL, e.what()); } \
   return g_err;

// hypredrv_DistributedErrorStateSync (reference src/HYPREDRV.c:3101, :3279): at the end of Setup / Apply every rank
// learns the error bits of all ranks, so that a failure on one rank is a failure of the collective call everywhere
static void dist_error_sync()
{
   Comm &cm = Comm::world();
   if (cm.size <= 1) return;
   try
   {
      std::vector<long long> all;
      cm.allgather_ll((long long)g_err, all);
      uint32_t merged = 0;
      for (long long v : all) merged |= (uint32_t)v;
      if (merged & ~g_err)
      {
         g_err |= merged;
         if (g_errmsg.empty()) g_errmsg = "an error was raised on another rank";
      }
   }
   catch (const std::exception &e)
   {
      err_set(ERR_HYPRE_INTERNAL, e.what());
   }
}
#define API_CATCH_SYNC                                                   \
   }                                                                     \
   catch (const std::exception &e) { err_set(ERR_HYPRE_INTERNAL, e.what()); } \
   dist_error_sync();                                                    \
   return g_err;

static void stats_begin(Stats &s, const std::string &name) { s.open[name] = clk::now(); }
static double stats_end(Stats &s, const std::string &name)
{
   auto it = s.open.find(name);
   if (it == s.open.end()) return 0.0;
   double dt = std::chrono::duration<double>(clk::now() - it->second).count();
   s.open.erase(it);
   return dt;
}

static uint32_t annotate(hypredrv_struct *h, const char *name, bool begin)
{
   if (!name) return err_set(ERR_UNKNOWN_TIMING);
   std::string n(name);
   Stats      &s = h->stats;
   if (n.rfind("Run", 0) == 0 || n == "initialize" || n == "finalize") return g_err; // free-form markers (stats.c:324-327)
   if (n == "system" || n == "matrix" || n == "rhs" || n == "dofmap")
   {
      if (begin)
      {
         stats_begin(s, n);
         if (n == "system" || n == "matrix") { s.system_open = true; s.new_system = true; }
      }
      else s.pending_build += stats_end(s, n);
      return g_err;
   }
   if (n == "prec" || n == "solve" || n == "reset_x0")
   {
      if (begin && n == "solve")
      { // first solve of a new system advances the flat counter; active levels prefix the entry's path
         if (s.new_system) { s.systems_solved++; s.new_system = false; }
         std::string p;
         for (int l = 0; l < kStatsMaxLevels; l++)
            if (s.level_open[l] && s.level_current_id[l] > 0) p += (p.empty() ? "" : ".") + std::to_string(s.level_current_id[l]);
         if (!p.empty()) p += "." + std::to_string(s.systems_solved);
         s.cur().path = p;
      }
      if (begin) stats_begin(s, n);
      else
      {
         double dt = stats_end(s, n);
         if (n == "prec") s.cur().prec += dt;
         if (n == "solve") s.cur().solve += dt;
      }
      return g_err;
   }
   return err_set(ERR_UNKNOWN_TIMING, "unknown annotation '" + n + "'");
}

// ---------------------------------------------------------------- lifecycle

extern "C" uint32_t HYPREDRV_Initialize(void)
{
   if (!g_initialized)
   {
      HYPRE_Initialize();
      g_initialized = true;
   }
   return HYPREDRV_SUCCESS;
}
extern "C" uint32_t HYPREDRV_Finalize(void)
{
   if (g_initialized)
   {
      mpi_leave(); // ranks joined through an MPI communicator hand back the duplicate while MPI is alive
      HYPRE_Finalize();
      g_initialized = false;
   }
   return HYPREDRV_SUCCESS;
}

extern "C" uint32_t HYPREDRV_AMD_CommGetUniqueId(void *uid)
{
   err_reset();
   API_TRY
   rccl_get_unique_id(uid);
   API_CATCH
}
extern "C" uint32_t HYPREDRV_AMD_CommInit(int rank, int world, int device, const void *uid)
{
   err_reset();
   API_TRY
   if (device > 0)
   { // a launcher that shows every rank exactly one device (HIP_VISIBLE_DEVICES per rank): LOCAL_RANK is not an index then
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) == hipSuccess && ndev == 1) device = 0;
   }
   if (device >= 0) HDA_HIP(hipSetDevice(device));
   // HDA_FORCE_RCCL: build a 1-rank RCCL communicator too (transport self-test on a single GPU)
   if (world > 1 || getenv("HDA_FORCE_RCCL")) Comm::set_world(make_rccl_comm(rank, world, uid));
   Comm::set_explicitly_joined(true);
   API_CATCH
}
extern "C" uint32_t HYPREDRV_AMD_CommInitCallbacks(int rank, int world, int device, HYPREDRV_AMD_AllreduceFn ar, HYPREDRV_AMD_AlltoallvFn a2a)
{
   err_reset();
   API_TRY
   if (device >= 0 && hipSetDevice(device) != hipSuccess) (void)hipGetLastError();
   if (world > 1) Comm::set_world(make_callback_comm(rank, world, ar, a2a));
   Comm::set_explicitly_joined(true);
   API_CATCH
}
extern "C" uint32_t HYPREDRV_AMD_CommFinalize(void)
{
   Comm::set_world(make_self_comm());
   Comm::set_explicitly_joined(false);
   return HYPREDRV_SUCCESS;
}

extern "C" uint32_t HYPREDRV_Create(MPI_Comm comm, HYPREDRV_t *out)
{
   if (!g_initialized) return err_set(ERR_HYPREDRV_NOT_INITIALIZED);
   if (!out) return err_set(ERR_UNKNOWN_HYPREDRV_OBJ);
   // an MPI program hands its communicator over here (reference: rank and size come from it, src/HYPREDRV.c:1014-1041): unless the
   // launcher has joined the ranks itself (HYPREDRV_AMD_CommInit*), they are joined now -- RCCL with one GPU per rank, the
   // host-staged transport over MPI for ranks that share one (hda_mpi.cpp)
   try
   {
      mpi_autojoin((int)comm);
      int cr = 0, cs = 1;
      if (mpi_comm_size((int)comm, &cr, &cs) && cs != Comm::world().size && (mpi_joined() || Comm::explicitly_joined()))
         return err_set(ERR_UNKNOWN, "HYPREDRV_Create: the communicator has " + std::to_string(cs) + " ranks but the library is joined on " +
                                        std::to_string(Comm::world().size) + " (one communicator per process; HYPREDRV_Finalize leaves it)");
   }
   catch (const std::exception &e)
   {
      return err_set(ERR_UNKNOWN, std::string("HYPREDRV_Create: joining the MPI ranks failed: ") + e.what());
   }
   auto *h      = new hypredrv_struct();
   h->comm      = comm;
   h->mypid     = Comm::world().rank;
   h->nprocs    = Comm::world().size;
   h->cookie    = PreconCookie{h};
   h->args.precon_variants.push_back(PreconArgs());
   *out = h;
   return HYPREDRV_SUCCESS;
}

static void destroy_system(hypredrv_struct *h)
{
   if (h->owns_M && h->mat_M && h->mat_M != h->mat_A) HYPRE_IJMatrixDestroy(h->mat_M);
   h->owns_M = false;
   if (h->owns_A && h->mat_A) HYPRE_IJMatrixDestroy(h->mat_A);
   if (h->owns_b && h->vec_b) HYPRE_IJVectorDestroy(h->vec_b);
   if (h->owns_x && h->vec_x) HYPRE_IJVectorDestroy(h->vec_x);
   if (h->owns_x0 && h->vec_x0) HYPRE_IJVectorDestroy(h->vec_x0);
   h->mat_A = h->mat_M = nullptr;
   h->vec_b = h->vec_x = h->vec_x0 = nullptr;
   h->owns_A = h->owns_b = h->owns_x = h->owns_x0 = false;
}

extern "C" uint32_t HYPREDRV_Destroy(HYPREDRV_t *hp)
{
   if (!hp || !*hp) return err_set(ERR_UNKNOWN_HYPREDRV_OBJ);
   hypredrv_struct *h = *hp;
   if (h->solver) HYPRE_ParCSRPCGDestroy(h->solver); // every Krylov handle is the same struct
   if (h->precon) HYPRE_BoomerAMGDestroy(h->precon);
   for (HYPRE_Solver a : h->precon_aux) HYPRE_BoomerAMGDestroy(a);
   h->precon_aux.clear();
   destroy_system(h);
   delete h;
   *hp = nullptr;
   return HYPREDRV_SUCCESS;
}

extern "C" uint32_t HYPREDRV_PrintLibInfo(MPI_Comm, int print_datetime)
{
   if (Comm::world().rank) return HYPREDRV_SUCCESS;
   if (print_datetime)
   {
      time_t t = time(nullptr);
      char   buf[64];
      strftime(buf, sizeof(buf), "%Y-%m-%d %H:%M:%S", localtime(&t));
      printf("Date and time: %s\n", buf);
   }
   printf("\nUsing HYPREDRV_DEVELOP_STRING: %s\n\n", HYPREDRV_DEVELOP_STRING);
   printf("Running on %d MPI rank%s\n", Comm::world().size, Comm::world().size > 1 ? "s" : "");
   return HYPREDRV_SUCCESS;
}
extern "C" uint32_t HYPREDRV_PrintSystemInfo(MPI_Comm)
{
   if (Comm::world().rank) return HYPREDRV_SUCCESS;
   int n = 0;
   if (hipGetDeviceCount(&n) == hipSuccess && n > 0)
   {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, 0) == hipSuccess)
         printf("GPU: %s (%s), %d CUs, %.1f GiB HBM; transport: %s\n", p.name, p.gcnArchName, p.multiProcessorCount,
                (double)p.totalGlobalMem / (1 << 30), Comm::world().name());
   }
   else printf("GPU: none visible (the MI355X solve path cannot run)\n");
   return HYPREDRV_SUCCESS;
}
extern "C" uint32_t HYPREDRV_PrintExitInfo(MPI_Comm, const char *argv0)
{
   if (Comm::world().rank) return HYPREDRV_SUCCESS;
   time_t t = time(nullptr);
   char   buf[64];
   strftime(buf, sizeof(buf), "%Y-%m-%d %H:%M:%S", localtime(&t));
   printf("Date and time: %s\n%s done!\n", buf, argv0 ? argv0 : "hypredrive");
   return HYPREDRV_SUCCESS;
}

static bool file_exists(const std::string &p)
{
   struct stat st;
   return stat(p.c_str(), &st) == 0;
}

extern "C" uint32_t HYPREDRV_InputArgsParse(int argc, char **argv, HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   if (argc < 1 || !argv || !argv[0]) return err_set(ERR_MISSING_KEY, "no YAML input given");
   std::string text, first(argv[0]), base_dir;
   // argv[0] is a file name or the YAML text itself (reference src/internal/args.c:1478-1487)
   if (first.find('\n') == std::string::npos && file_exists(first))
   {
      const size_t slash = first.find_last_of('/');
      if (slash != std::string::npos) base_dir = first.substr(0, slash);
      std::ifstream     f(first);
      std::stringstream ss;
      ss << f.rdbuf();
      text = ss.str();
   }
   else if (first.find(':') != std::string::npos) text = first;
   else return err_set(ERR_FILE_NOT_FOUND, "cannot open YAML input '" + first + "'");
   h->tree = YNode();
   std::string msg;
   uint32_t    e = yaml_parse(text, h->tree, msg);
   if (e) return err_set(e, msg);
   e = yaml_expand_includes(h->tree, base_dir, msg); // "include: file.yml" (solver / preconditioner blocks kept in their own files)
   if (e) return err_set(e, msg);
   // "-a|--args --path:to:key value ..." overrides (src/internal/main.c:23-31); a bare
   // "--path:key value" pair is accepted too (examples/src/C_laplacian/laplacian.c:362-365)
   for (int i = 1; i < argc; i++)
   {
      std::string a(argv[i]);
      if (a == "-a" || a == "--args") continue;
      if (a.rfind("--", 0) == 0 && a.find(':') != std::string::npos && i + 1 < argc)
      {
         yaml_override(h->tree, a, argv[i + 1]);
         i++;
      }
   }
   InputArgs fresh;
   e = args_from_yaml(h->tree, h->lib_mode, fresh, msg);
   if (e) return err_set(e, msg);
   h->args               = fresh;
   h->stats.use_millisec = h->args.general.use_millisec;
   if (!h->args.general.name.empty()) h->name = h->args.general.name;
   if (h->args.general.print_config_params && !h->mypid)
   {
      printf("------------------------------------------------------------------------------------\n");
      yaml_print(h->tree, stdout);
      printf("------------------------------------------------------------------------------------\n");
   }
   API_CATCH
}

extern "C" uint32_t HYPREDRV_SetLibraryMode(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   h->lib_mode                         = true;
   h->args.general.print_config_params = 0;
   return HYPREDRV_SUCCESS;
}
extern "C" uint32_t HYPREDRV_ObjectSetName(HYPREDRV_t h, const char *name)
{
   CHECK_INIT_OBJ(h);
   h->name = name ? name : "";
   return HYPREDRV_SUCCESS;
}
extern "C" uint32_t HYPREDRV_InputArgsGetWarmup(HYPREDRV_t h, int *v) { CHECK_INIT_OBJ(h); *v = h->args.general.warmup; return HYPREDRV_SUCCESS; }
extern "C" uint32_t HYPREDRV_InputArgsGetNumRepetitions(HYPREDRV_t h, int *v) { CHECK_INIT_OBJ(h); *v = h->args.general.num_repetitions; return HYPREDRV_SUCCESS; }
extern "C" uint32_t HYPREDRV_InputArgsGetNumLinearSystems(HYPREDRV_t h, int *v) { CHECK_INIT_OBJ(h); *v = h->args.ls.num_systems; return HYPREDRV_SUCCESS; }
extern "C" uint32_t HYPREDRV_InputArgsGetNumPreconVariants(HYPREDRV_t h, int *v) { CHECK_INIT_OBJ(h); *v = (int)h->args.precon_variants.size(); return HYPREDRV_SUCCESS; }
extern "C" uint32_t HYPREDRV_InputArgsSetPreconVariant(HYPREDRV_t h, int idx)
{
   CHECK_INIT_OBJ(h);
   if (idx < 0 || idx >= (int)h->args.precon_variants.size()) return err_set(ERR_OUT_OF_BOUNDS, "preconditioner variant index out of range");
   h->args.active_variant = idx;
   return HYPREDRV_SUCCESS;
}

static std::map<std::string, std::string> &user_presets(bool solver)
{
   return solver ? RankState<std::map<std::string, std::string>, 6>::get() : RankState<std::map<std::string, std::string>, 7>::get();
}
extern "C" uint32_t HYPREDRV_PreconPresetRegister(const char *name, const char *yaml, const char *)
{
   if (!name || !yaml) return err_set(ERR_INVALID_VAL, "preset needs a name and a YAML text");
   user_presets(false)[name] = yaml;
   return HYPREDRV_SUCCESS;
}
extern "C" uint32_t HYPREDRV_SolverPresetRegister(const char *name, const char *yaml, const char *)
{
   if (!name || !yaml) return err_set(ERR_INVALID_VAL, "preset needs a name and a YAML text");
   user_presets(true)[name] = yaml;
   return HYPREDRV_SUCCESS;
}
extern "C" uint32_t HYPREDRV_InputArgsSetPreconPreset(HYPREDRV_t h, const char *preset)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   if (!preset) return err_set(ERR_INVALID_VAL, "null preset");
   std::string text = std::string("preset: ") + preset, msg;
   auto        it   = user_presets(false).find(preset);
   PreconArgs  p;
   uint32_t    e;
   if (it != user_presets(false).end()) e = precon_from_text(it->second, p, msg);
   else
   {
      YNode root;
      e = yaml_parse("preconditioner:\n  " + text + "\n", root, msg);
      InputArgs tmp;
      if (!e) e = args_from_yaml(root, true, tmp, msg);
      if (!e) p = tmp.precon_variants[0];
   }
   if (e) return err_set(e, msg);
   h->args.precon_variants.assign(1, p);
   h->args.active_variant = 0;
   h->args.has_precon     = true;
   return g_err;
}
extern "C" uint32_t HYPREDRV_InputArgsSetSolverPreset(HYPREDRV_t h, const char *preset)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   if (!preset) return err_set(ERR_INVALID_VAL, "null preset");
   std::string msg;
   auto        it = user_presets(true).find(preset);
   uint32_t    e  = solver_from_text(it != user_presets(true).end() ? it->second : std::string(preset), h->args.solver, msg);
   if (e) return err_set(e, msg);
   return g_err;
}

// ------------------------------------------------------------- linear system

// LinearSystemDataFilenameResolve (reference src/internal/linsys.c:832-866): with a dirname the files of system k live in
// "<dirname>_<suffix>/<filename>", suffix = init_suffix + k zero-padded to digits_suffix; a plain filename is used as it
// is; a basename names "<basename>_<suffix>"
static std::string ls_path(const hypredrv_struct *h, const std::string &fn, const std::string &base = std::string())
{
   const LSArgs &l = h->args.ls;
   char          suf[32];
   const int id  = std::max(h->current_system_index, 0);
   const int num = (id >= 1 && (size_t)(id - 1) < l.set_suffix.size()) ? l.set_suffix[(size_t)(id - 1)] : std::max(l.init_suffix, 0) + id;
   snprintf(suf, sizeof suf, "_%0*d", std::max(l.digits_suffix, 1), num);
   if (!l.dirname.empty()) return l.dirname + suf + "/" + fn;
   if (!fn.empty()) return fn;
   if (!base.empty()) return base + suf;
   return fn;
}

extern "C" uint32_t HYPREDRV_LinearSystemReadMatrix(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (h->args.ls.matrix_filename.empty() && h->args.ls.matrix_basename.empty())
      return err_set(ERR_MISSING_KEY, "linear_system.matrix_filename is not set");
   if (h->args.ls.type != 1 && h->args.ls.type != 3)
      return err_set(ERR_MISSING_LIB, "linear_system.type must be 'ij' (hypre ASCII or hypredrive multipart binary files) or 'mtx' (Matrix Market)");
   annotate(h, "matrix", true);
   if (h->owns_A && h->mat_A) HYPRE_IJMatrixDestroy(h->mat_A);
   h->mat_A = nullptr;
   const std::string path = ls_path(h, h->args.ls.matrix_filename, h->args.ls.matrix_basename);
   // reference src/internal/linsys.c:946-1000: binary parts win over ASCII files of the same prefix
   HYPRE_Int rc;
   if (h->args.ls.type == 3) rc = HYPRE_IJMatrixReadMM(path.c_str(), h->comm, HYPRE_PARCSR, &h->mat_A);
   else if (const int np = hda_count_binary_parts(path.c_str()); np > 0) rc = hda_IJMatrixReadMultipartBinary(path.c_str(), h->comm, np, &h->mat_A);
   else rc = HYPRE_IJMatrixRead(path.c_str(), h->comm, HYPRE_PARCSR, &h->mat_A);
   if (rc)
   {
      annotate(h, "matrix", false);
      const std::string why = hypre_last_error();
      HYPRE_ClearAllErrors();
      h->mat_A = nullptr;
      const bool missing = why.find("cannot open") != std::string::npos;
      const bool badfile = why.find("Could not read") != std::string::npos || why.find("Invalid") != std::string::npos ||
                           why.find("Detected") != std::string::npos || why.find("exceeds") != std::string::npos;
      return err_set(missing ? ERR_FILE_NOT_FOUND : badfile ? ERR_FILE_UNEXPECTED_ENTRY : ERR_HYPRE_INTERNAL, why);
   }
   h->owns_A = true;
   h->mat_M  = h->mat_A;
   h->stats.new_system = true;
   annotate(h, "matrix", false);
   API_CATCH
}

// ASCII "<prefix>.<rank>" or multipart binary "<prefix>.<part>.bin" (reference linsys.c:884)
static HYPRE_Int read_vector_file(hypredrv_struct *h, const std::string &path, HYPRE_IJVector *v)
{
   if (const int np = hda_count_binary_parts(path.c_str()); np > 0) return hda_IJVectorReadMultipartBinary(path.c_str(), h->comm, np, v);
   return HYPRE_IJVectorRead(path.c_str(), h->comm, HYPRE_PARCSR, v);
}

static HYPRE_IJVector new_vector_like(hypredrv_struct *h, double value)
{
   HYPRE_IJVector v = nullptr;
   HYPRE_IJVectorCreate(h->comm, h->mat_A->ilower, h->mat_A->iupper, &v);
   HYPRE_IJVectorSetObjectType(v, HYPRE_PARCSR);
   v->initialized = true; // device resident from the start: HYPRE_IJVectorInitialize would zero a host staging copy first
   HYPRE_ParVectorSetConstantValues(v, value);
   return v;
}

extern "C" uint32_t HYPREDRV_LinearSystemSetMatrix(HYPREDRV_t h, HYPRE_Matrix A)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (h->owns_A && h->mat_A && h->mat_A != A) HYPRE_IJMatrixDestroy(h->mat_A);
   h->mat_A  = A;
   h->mat_M  = A;
   h->stats.new_system = true;
   h->owns_A = !h->lib_mode; // driver mode takes ownership (reference src/HYPREDRV.c:2013)
   if (A && !A->assembled) HYPRE_IJMatrixAssemble(A);
   consume_hypre_errors();
   API_CATCH
}
// reference src/internal/linsys.c:2620-2660: a handle, else the file named by linear_system.precmat_filename, else A itself
extern "C" uint32_t HYPREDRV_LinearSystemSetPrecMatrix(HYPREDRV_t h, HYPRE_Matrix M)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (h->owns_M && h->mat_M && h->mat_M != h->mat_A && h->mat_M != M) HYPRE_IJMatrixDestroy(h->mat_M);
   h->owns_M = false;
   if (M) h->mat_M = M;
   else if ((!h->args.ls.precmat_filename.empty() && h->args.ls.precmat_filename != h->args.ls.matrix_filename) ||
            (h->args.ls.precmat_filename.empty() && !h->args.ls.precmat_basename.empty()))
   {
      const std::string path = ls_path(h, h->args.ls.precmat_filename, h->args.ls.precmat_basename);
      HYPRE_IJMatrix    P    = nullptr;
      HYPRE_Int         rc;
      if (h->args.ls.type == 3) rc = HYPRE_IJMatrixReadMM(path.c_str(), h->comm, HYPRE_PARCSR, &P);
      else if (const int np = hda_count_binary_parts(path.c_str()); np > 0) rc = hda_IJMatrixReadMultipartBinary(path.c_str(), h->comm, np, &P);
      else rc = HYPRE_IJMatrixRead(path.c_str(), h->comm, HYPRE_PARCSR, &P);
      if (rc)
      {
         const std::string why = hypre_last_error();
         HYPRE_ClearAllErrors();
         return err_set(why.find("cannot open") != std::string::npos ? ERR_FILE_NOT_FOUND : ERR_FILE_UNEXPECTED_ENTRY, why);
      }
      h->mat_M  = P;
      h->owns_M = true;
   }
   else h->mat_M = h->mat_A;
   API_CATCH
}

extern "C" uint32_t HYPREDRV_LinearSystemSetRHS(HYPREDRV_t h, HYPRE_Vector vec)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (h->owns_b && h->vec_b && h->vec_b != vec) HYPRE_IJVectorDestroy(h->vec_b);
   h->owns_b = false;
   if (vec)
   {
      h->vec_b  = vec;
      h->owns_b = !h->lib_mode;
      if (!vec->assembled) HYPRE_IJVectorAssemble(vec);
   }
   else
   { // built from the YAML input (reference src/internal/linsys.c:1779-1840)
      if (!h->mat_A) return err_set(ERR_UNKNOWN, "SetRHS needs the matrix first");
      annotate(h, "rhs", true);
      const LSArgs &l = h->args.ls;
      if (l.rhs_mode == 2 && (!l.rhs_filename.empty() || !l.rhs_basename.empty()))
      {
         const std::string path = ls_path(h, l.rhs_filename, l.rhs_basename);
         if (read_vector_file(h, path, &h->vec_b))
         {
            annotate(h, "rhs", false);
            const std::string why = hypre_last_error();
            HYPRE_ClearAllErrors();
            return err_set(why.find("cannot open") != std::string::npos ? ERR_FILE_NOT_FOUND : ERR_HYPRE_INTERNAL, why);
         }
      }
      else if (l.rhs_mode == 0) h->vec_b = new_vector_like(h, 0.0);
      else if (l.rhs_mode == 1 || l.rhs_mode == 2) h->vec_b = new_vector_like(h, 1.0);
      else
      {
         annotate(h, "rhs", false);
         return err_set(ERR_MISSING_LIB, "rhs_mode random/randsol is not supported by this build");
      }
      h->owns_b = true;
      annotate(h, "rhs", false);
   }
   consume_hypre_errors();
   API_CATCH
}

extern "C" uint32_t HYPREDRV_LinearSystemSetInitialGuess(HYPREDRV_t h, HYPRE_Vector vec)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (!h->mat_A) return err_set(ERR_UNKNOWN, "SetInitialGuess needs the matrix first");
   if (h->owns_x0 && h->vec_x0 && h->vec_x0 != vec) HYPRE_IJVectorDestroy(h->vec_x0);
   h->owns_x0 = false;
   // (the working solution is recreated only at the END: init_guess_mode "previous" reads the last solve's values from it while x0
   // is built -- reference src/internal/linsys.c:1999-2003, 2046-2068; asserted by its tests/test_init_guess.c:170-199)
   if (vec)
   {
      h->vec_x0 = vec;
      if (!vec->assembled) HYPRE_IJVectorAssemble(vec);
   }
   else
   {
      const LSArgs &l = h->args.ls;
      if (l.init_guess_mode == 2 && !l.x0_filename.empty())
      {
         const std::string path = ls_path(h, l.x0_filename);
         if (read_vector_file(h, path, &h->vec_x0))
         {
            const std::string why = hypre_last_error();
            HYPRE_ClearAllErrors();
            return err_set(why.find("cannot open") != std::string::npos ? ERR_FILE_NOT_FOUND : ERR_HYPRE_INTERNAL, why);
         }
      }
      else if (l.init_guess_mode == 1) h->vec_x0 = new_vector_like(h, 1.0);
      else if (l.init_guess_mode == 0 || l.init_guess_mode == 2) h->vec_x0 = new_vector_like(h, 0.0);
      else if (l.init_guess_mode == 4)
      { // previous: the last solve's solution when its row range is this system's, zeros otherwise
         h->vec_x0 = new_vector_like(h, 0.0);
         if (h->vec_x && h->vec_x->assembled && h->vec_x->jlower == h->vec_x0->jlower && h->vec_x->jupper == h->vec_x0->jupper)
            HYPRE_ParVectorCopy(h->vec_x, h->vec_x0);
      }
      else return err_set(ERR_MISSING_LIB, "init_guess_mode random is not supported by this build");
      h->owns_x0 = true;
   }
   if (h->owns_x && h->vec_x) HYPRE_IJVectorDestroy(h->vec_x);
   h->vec_x  = new_vector_like(h, 0.0);
   h->owns_x = true;
   HYPRE_ParVectorCopy(h->vec_x0, h->vec_x);
   consume_hypre_errors();
   API_CATCH
}

extern "C" uint32_t HYPREDRV_LinearSystemResetInitialGuess(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (!h->vec_x || !h->vec_x0) return err_set(ERR_UNKNOWN, "ResetInitialGuess needs SetInitialGuess first");
   annotate(h, "reset_x0", true);
   h->stats.next_entry_if_used(); // entry indices advance on reset_x0 (reference stats.c:350-371)
   HYPRE_ParVectorCopy(h->vec_x0, h->vec_x);
   annotate(h, "reset_x0", false);
   API_CATCH
}

extern "C" uint32_t HYPREDRV_LinearSystemSetSolution(HYPREDRV_t h, HYPRE_Vector vec)
{
   CHECK_INIT_OBJ(h);
   if (!vec) return g_err;
   if (h->owns_x && h->vec_x && h->vec_x != vec) HYPRE_IJVectorDestroy(h->vec_x);
   h->vec_x  = vec;
   h->owns_x = false;
   return g_err;
}
extern "C" uint32_t HYPREDRV_LinearSystemSetReferenceSolution(HYPREDRV_t h, HYPRE_Vector vec)
{
   CHECK_INIT_OBJ(h);
   h->vec_xref = vec;
   return g_err;
}

extern "C" uint32_t HYPREDRV_LinearSystemBuild(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   h->current_system_index++;
   h->scal.matrices_are_scaled = h->scal.rhs_is_scaled = h->scal.x_is_scaled = false; // src/HYPREDRV.c:1949-1956
   if (HYPREDRV_LinearSystemReadMatrix(h)) return g_err;
   if (HYPREDRV_LinearSystemSetRHS(h, nullptr)) return g_err;
   if (HYPREDRV_LinearSystemSetInitialGuess(h, nullptr)) return g_err;
   HYPREDRV_LinearSystemSetPrecMatrix(h, nullptr);
   if (HYPREDRV_LinearSystemReadDofmap(h)) return g_err; // src/HYPREDRV.c LinearSystemBuild: the dofmap file, when one is named
   h->stats.ls_counter++;
   if (!h->mypid)
   {
      printf("====================================================================================\n");
      printf("Solving linear system #%d with %lld rows and %lld nonzeros...\n", h->stats.ls_counter, h->mat_A->global_rows,
             h->mat_A->global_nnz);
   }
   return g_err;
}

// CSR ingestion (reference src/internal/linsys.c:1190-1405): buffers are copied, the
// resulting objects are always owned (include/HYPREDRV.h:830-834 there)
// HYPREDRV_LinearSystemSetMatrixFromCSR / SetRHSFromArray (reference src/HYPREDRV.c:2139-2190, 2197-2254 over
// src/internal/linsys.c:1190-1405, 1412-1491): same order of events and the same refusals, each with ERROR_INVALID_VAL -- the old
// matrix is released first (a failed build leaves NO matrix behind), row_end < row_start, a negative indptr[0], counts outside
// HYPRE_Int, a decreasing indptr, missing column / value arrays for a non-empty block; the right-hand side needs a matrix and its
// exact row range.  The reference's own tests/test_setmatrix_from_csr.c runs against these (oracle/Makefile ref_tests).
// allow_empty: an EMPTY row block (row_end == row_start - 1), which hypre's IJ layer represents and the reference's entry refuses --
// only reachable through HYPREDRV_AMD_LinearSystemSetEmptyBlock (row partitions with a rank that owns nothing).
static uint32_t set_matrix_from_csr(HYPREDRV_t h, HYPRE_BigInt row_start, HYPRE_BigInt row_end, const HYPRE_BigInt *indptr,
                                    const HYPRE_BigInt *cols, const HYPRE_Real *data, bool allow_empty)
{
   if (h->owns_A && h->mat_A) HYPRE_IJMatrixDestroy(h->mat_A);
   if (h->owns_M && h->mat_M && h->mat_M != h->mat_A) HYPRE_IJMatrixDestroy(h->mat_M);
   h->mat_A = h->mat_M = nullptr;
   h->owns_A = h->owns_M = false;
   if (!indptr) return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: mat_ptr and indptr must be non-NULL");
   if (row_end < row_start && !(allow_empty && row_end == row_start - 1))
      return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: row_end (" + std::to_string((long long)row_end) + ") < row_start (" +
                                          std::to_string((long long)row_start) + ")");
   const long long nrows_big = (long long)row_end - (long long)row_start + 1;
   if (nrows_big > 2147483647LL) return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: local row count is out of HYPRE_Int range");
   const int n = (int)nrows_big;
   if (indptr[0] < 0) return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: indptr[0] must be nonnegative");
   const long long nnz_big = (long long)indptr[n] - (long long)indptr[0];
   if (nnz_big < 0) return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: indptr[nrows] < indptr[0]");
   if (nnz_big > 2147483647LL) return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: local nonzero count exceeds HYPRE_Int range");
   if (nnz_big > 0 && (!cols || !data)) return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: col_indices/data must be non-NULL when nnz > 0");
   for (int i = 0; i < n; i++)
      if (indptr[i + 1] < indptr[i])
         return err_set(ERR_INVALID_VAL, "BuildMatrixFromCSR: indptr is not monotonically non-decreasing at row " + std::to_string(i));
   annotate(h, "matrix", true);
   HYPRE_IJMatrix A = nullptr;
   HYPRE_IJMatrixCreate(h->comm, row_start, row_end, row_start, row_end, &A);
   HYPRE_IJMatrixSetObjectType(A, HYPRE_PARCSR);
   HYPRE_IJMatrixInitialize(A);
   const HYPRE_BigInt base = indptr[0];
   const size_t       nnz  = (size_t)nnz_big;
   // the caller's arrays go to the device as they are (columns mapped, rows sorted and checked there); only a block with a column
   // named twice in a row takes the staged path below, which defines what that means (the later value wins)
   const bool direct = !(getenv("HDA_CSR_DIRECT") && atoi(getenv("HDA_CSR_DIRECT")) == 0);
   int ndev = 0;
   if (direct && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) // (no device: the staged path reports it through HYPRE_IJMatrixAssemble)
   {
      bool done = false;
      try
      {
         done = A->assemble_csr((const long long *)indptr, (const long long *)cols, data);
      }
      catch (...)
      {
         HYPRE_IJMatrixDestroy(A);
         annotate(h, "matrix", false);
         throw;
      }
      if (done)
      {
         h->mat_A = h->mat_M = A;
         h->owns_A           = true;
         annotate(h, "matrix", false);
         return g_err;
      }
   }
   A->t_row.resize(nnz); A->t_col.resize(nnz); A->t_val.resize(nnz); A->t_add.assign(nnz, 0);
   for (int i = 0; i < n; i++)
      for (HYPRE_BigInt k = indptr[i]; k < indptr[i + 1]; k++)
      {
         A->t_row[(size_t)(k - base)] = i;
         A->t_col[(size_t)(k - base)] = cols[k];
         A->t_val[(size_t)(k - base)] = data[k];
      }
   if (HYPRE_IJMatrixAssemble(A))
   {
      HYPRE_IJMatrixDestroy(A);
      annotate(h, "matrix", false);
      consume_hypre_errors();
      return g_err;
   }
   h->mat_A = h->mat_M = A;
   h->owns_A           = true;
   annotate(h, "matrix", false);
   return g_err;
}

static uint32_t set_rhs_from_array(HYPREDRV_t h, HYPRE_BigInt row_start, HYPRE_BigInt row_end, const HYPRE_Real *values, bool allow_empty)
{
   if (!h->mat_A) return err_set(ERR_INVALID_VAL, "HYPREDRV_LinearSystemSetRHSFromArray: matrix must be set before RHS");
   if (row_start != h->mat_A->ilower || row_end != h->mat_A->iupper)
      return err_set(ERR_INVALID_VAL, "HYPREDRV_LinearSystemSetRHSFromArray: RHS row range [" + std::to_string((long long)row_start) + ", " +
                                          std::to_string((long long)row_end) + "] does not match matrix row range [" +
                                          std::to_string((long long)h->mat_A->ilower) + ", " + std::to_string((long long)h->mat_A->iupper) + "]");
   if (h->owns_b && h->vec_b) HYPRE_IJVectorDestroy(h->vec_b);
   h->vec_b  = nullptr;
   h->owns_b = false;
   if (row_end < row_start && !(allow_empty && row_end == row_start - 1)) return err_set(ERR_INVALID_VAL, "BuildRHSFromArray: row_end < row_start");
   const int n = (int)(row_end - row_start + 1);
   if (!values && !(allow_empty && n == 0)) return err_set(ERR_INVALID_VAL, "BuildRHSFromArray: values must be non-NULL");
   annotate(h, "rhs", true);
   HYPRE_IJVector b = nullptr;
   HYPRE_IJVectorCreate(h->comm, row_start, row_end, &b);
   int ndev = 0;
   if (n > 0 && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0)
   { // the caller's array goes to the device as it is (no host staging copy)
      b->initialized = true;
      b->d.alloc((size_t)n);
      b->capacity = b->d.size();
      b->d.upload(values, (size_t)n);
      b->assembled = true;
   }
   else
   {
      HYPRE_IJVectorInitialize(b);
      if (n) memcpy(b->stage.data(), values, sizeof(double) * (size_t)n);
      HYPRE_IJVectorAssemble(b);
   }
   h->vec_b  = b;
   h->owns_b = true;
   annotate(h, "rhs", false);
   consume_hypre_errors();
   return g_err;
}

extern "C" uint32_t HYPREDRV_LinearSystemSetMatrixFromCSR(HYPREDRV_t h, HYPRE_BigInt row_start, HYPRE_BigInt row_end,
                                                          const HYPRE_BigInt *indptr, const HYPRE_BigInt *cols,
                                                          const HYPRE_Real *data)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   return set_matrix_from_csr(h, row_start, row_end, indptr, cols, data, false);
   API_CATCH
}

extern "C" uint32_t HYPREDRV_LinearSystemSetRHSFromArray(HYPREDRV_t h, HYPRE_BigInt row_start, HYPRE_BigInt row_end,
                                                         const HYPRE_Real *values)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   return set_rhs_from_array(h, row_start, row_end, values, false);
   API_CATCH
}

// A rank that owns NO rows of a row-partitioned system: empty matrix block and empty right-hand side at row_start (hypre's IJ layer
// represents such a rank as [row_start, row_start - 1]; the reference's CSR entry refuses the range, src/internal/linsys.c:1220-1226).
extern "C" uint32_t HYPREDRV_AMD_LinearSystemSetEmptyBlock(HYPREDRV_t h, HYPRE_BigInt row_start)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   const HYPRE_BigInt ip[1] = {0};
   if (set_matrix_from_csr(h, row_start, row_start - 1, ip, nullptr, nullptr, true)) return g_err;
   return set_rhs_from_array(h, row_start, row_start - 1, nullptr, true);
   API_CATCH
}

extern "C" uint32_t HYPREDRV_AMD_LinearSystemSetLaplacian7pt(HYPREDRV_t h, const int n[3], const int P[3], const double c[3])
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   Comm &cm = Comm::world();
   HDA_REQUIRE(P[0] * P[1] * P[2] == cm.size, "processor grid does not match the number of ranks");
   // MPI_Cart_create row-major rank -> block coordinates (laplacian.c:545-548)
   int pc[3] = {cm.rank / (P[1] * P[2]), (cm.rank / P[2]) % P[1], cm.rank % P[2]};
   long long st[3], ln[3];
   for (int d = 0; d < 3; d++)
   {
      int size = n[d] / P[d], rest = n[d] - size * P[d];
      auto start = [&](int j) { return (long long)size * j + (j < rest ? j : rest); };
      st[d] = start(pc[d]);
      ln[d] = start(pc[d] + 1) - st[d];
   }
   const long long nloc   = ln[0] * ln[1] * ln[2];
   const long long ilower = st[0] * n[1] * n[2] + st[1] * n[2] * ln[0] + st[2] * ln[0] * ln[1];
   annotate(h, "system", true);
   destroy_system(h);
   HYPRE_IJMatrix A = nullptr;
   HYPRE_IJVector b = nullptr;
   HYPRE_IJMatrixCreate(h->comm, ilower, ilower + nloc - 1, ilower, ilower + nloc - 1, &A);
   HYPRE_IJVectorCreate(h->comm, ilower, ilower + nloc - 1, &b);
   // count the entries of this block analytically: 7 per row minus missing neighbours on the global boundary
   DArray<int>       rp((size_t)nloc + 1);
   long long         nnz = 0;
   {
      auto faces = [&](int d, bool hi) { return hi ? (st[d] + ln[d] == n[d]) : (st[d] == 0); };
      nnz        = 7 * nloc;
      const long long area[3] = {ln[1] * ln[2], ln[0] * ln[2], ln[0] * ln[1]};
      for (int d = 0; d < 3; d++) nnz -= ((faces(d, false) ? 1 : 0) + (faces(d, true) ? 1 : 0)) * area[d];
   }
   HDA_REQUIRE(nnz < (1LL << 31), "local nnz must fit int32");
   DArray<long long> gc((size_t)nnz);
   DArray<double>    gv((size_t)nnz);
   b->d.alloc((size_t)nloc);
   b->capacity = (size_t)nloc;
   lap7_generate(n, P, pc, c, rp.data(), gc.data(), gv.data(), b->d.data(), (int)nloc);
   b->initialized = b->assembled = true;
   // ghost columns: the faces of neighbouring blocks; collect from the generated columns
   if (cm.size > 1)
   {
      std::vector<long long> hc = gc.to_host();
      std::vector<long long> gh;
      for (long long v : hc)
         if (v < ilower || v >= ilower + nloc) gh.push_back(v);
      std::sort(gh.begin(), gh.end());
      gh.erase(std::unique(gh.begin(), gh.end()), gh.end());
      A->ghost_gids = gh;
   }
   A->initialized = true;
   A->adopt_device((int)nloc, (int)nnz, rp, gc, gv);
   h->mat_A = h->mat_M = A;
   h->vec_b            = b;
   h->owns_A = h->owns_b = true;
   annotate(h, "system", false);
   if (HYPREDRV_LinearSystemSetInitialGuess(h, nullptr)) return g_err;
   API_CATCH
}

extern "C" uint32_t HYPREDRV_LinearSystemGetSolutionValues(HYPREDRV_t h, HYPRE_Complex **data)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (!h->vec_x || !data) return err_set(ERR_UNKNOWN, "no solution vector");
   HYPRE_IJVector x = h->vec_x;
   x->host_mirror.resize((size_t)std::max(x->nloc, 1));
   download_sync(x->host_mirror.data(), x->data(), sizeof(double) * (size_t)x->nloc);
   *data = x->host_mirror.data();
   API_CATCH
}
extern "C" uint32_t HYPREDRV_LinearSystemGetRHSValues(HYPREDRV_t h, HYPRE_Complex **data)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (!h->vec_b || !data) return err_set(ERR_UNKNOWN, "no right-hand side");
   HYPRE_IJVector b = h->vec_b;
   b->host_mirror.resize((size_t)std::max(b->nloc, 1));
   download_sync(b->host_mirror.data(), b->data(), sizeof(double) * (size_t)b->nloc);
   *data = b->host_mirror.data();
   API_CATCH
}
extern "C" uint32_t HYPREDRV_LinearSystemGetSolutionLength(HYPREDRV_t h, HYPRE_BigInt *length)
{
   CHECK_INIT_OBJ(h);
   if (!h->vec_x || !length) return err_set(ERR_UNKNOWN, "no solution vector");
   *length = h->vec_x->nloc;
   return g_err;
}

// L1 / L2 / Linf of the solution (reference src/internal/linsys.c:2815-2924; unknown type -> -1)
extern "C" uint32_t HYPREDRV_LinearSystemGetSolutionNorm(HYPREDRV_t h, const char *norm_type, double *norm)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (!h->vec_x || !norm || !norm_type) return err_set(ERR_UNKNOWN, "no solution vector");
   std::string t(norm_type);
   for (auto &ch : t) ch = (char)tolower((unsigned char)ch);
   if (t == "l2")
   {
      double p = 0.0;
      HYPRE_ParVectorInnerProd(h->vec_x, h->vec_x, &p);
      *norm = std::sqrt(p);
   }
   else if (t == "l1" || t == "linf")
   {
      HYPRE_Complex *d = nullptr;
      HYPREDRV_LinearSystemGetSolutionValues(h, &d);
      long long v[1];
      double    acc = 0.0;
      for (int i = 0; i < h->vec_x->nloc; i++) acc = (t == "l1") ? acc + std::fabs(d[i]) : std::max(acc, std::fabs(d[i]));
      // cross-rank: sum (L1) or max (Linf) through the integer-collective on the bit pattern is
      // not meaningful for doubles, so reduce through a one-element device all-reduce for L1
      if (Comm::world().size > 1)
      {
         if (t == "l1")
         {
            DArray<double> s(1);
            s.upload(&acc, 1);
            Comm::world().allreduce_sum_dev(s.data(), 1);
            s.download(&acc, 1);
         }
         else
         { // non-negative doubles order like their bit patterns
            memcpy(v, &acc, 8);
            Comm::world().allreduce_host(v, 1, 1);
            memcpy(&acc, v, 8);
         }
      }
      *norm = acc;
   }
   else *norm = -1.0;
   consume_hypre_errors();
   API_CATCH
}
extern "C" uint32_t HYPREDRV_LinearSystemGetSolution(HYPREDRV_t h, HYPRE_Vector *vec) { CHECK_INIT_OBJ(h); *vec = h->vec_x; return g_err; }
extern "C" uint32_t HYPREDRV_LinearSystemGetRHS(HYPREDRV_t h, HYPRE_Vector *vec) { CHECK_INIT_OBJ(h); *vec = h->vec_b; return g_err; }
extern "C" uint32_t HYPREDRV_LinearSystemGetMatrix(HYPREDRV_t h, HYPRE_Matrix *mat) { CHECK_INIT_OBJ(h); *mat = h->mat_A; return g_err; }

extern "C" uint32_t HYPREDRV_LinearSystemPrint(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (h->mat_A) HYPRE_IJMatrixPrint(h->mat_A, "IJ.out.A");
   if (h->vec_b) HYPRE_IJVectorPrint(h->vec_b, "IJ.out.b");
   consume_hypre_errors();
   API_CATCH
}

// ------------------------------------------------- outside the AMG-Krylov hot path

#define UNSUPPORTED(sig, what)                                                          \
   extern "C" uint32_t sig                                                              \
   {                                                                                    \
      return err_set(HYPREDRV_ERROR_UNSUPPORTED_AMD, what " is outside the MI355X AMG-Krylov path of this build"); \
   }
UNSUPPORTED(HYPREDRV_LinearSystemSetDiscreteGradient(HYPREDRV_t, HYPRE_Matrix), "AMS/ADS discrete gradient")
UNSUPPORTED(HYPREDRV_LinearSystemSetDiscreteCurl(HYPREDRV_t, HYPRE_Matrix), "ADS discrete curl")
UNSUPPORTED(HYPREDRV_LinearSystemSetCoordinates(HYPREDRV_t, HYPRE_Vector, HYPRE_Vector, HYPRE_Vector), "AMS/ADS coordinates")
// degree-of-freedom maps (reference src/HYPREDRV.c:2680-2724): here they feed BoomerAMG's dof_func when
// systems AMG is selected (coarsening.num_functions > 1, reference src/internal/amg.c:792-862)
extern "C" uint32_t HYPREDRV_LinearSystemSetDofmap(HYPREDRV_t h, int size, const int *dofmap)
{
   CHECK_INIT_OBJ(h);
   if (size < 0 || (size > 0 && !dofmap)) return err_set(ERR_INVALID_VAL, "SetDofmap: bad size or NULL map");
   h->dofmap.assign(dofmap, dofmap + size);
   return g_err;
}
extern "C" uint32_t HYPREDRV_LinearSystemSetInterleavedDofmap(HYPREDRV_t h, int num_local_blocks, int num_dof_types)
{
   CHECK_INIT_OBJ(h);
   if (num_local_blocks < 0 || num_dof_types <= 0) return err_set(ERR_INVALID_VAL, "SetInterleavedDofmap: bad block or type count");
   h->dofmap.resize((size_t)num_local_blocks * num_dof_types);
   for (int i = 0; i < num_local_blocks; i++)
      for (int j = 0; j < num_dof_types; j++) h->dofmap[(size_t)i * num_dof_types + j] = j;
   return g_err;
}
extern "C" uint32_t HYPREDRV_LinearSystemSetContiguousDofmap(HYPREDRV_t h, int num_local_blocks, int num_dof_types)
{
   CHECK_INIT_OBJ(h);
   if (num_local_blocks < 0 || num_dof_types <= 0) return err_set(ERR_INVALID_VAL, "SetContiguousDofmap: bad block or type count");
   h->dofmap.resize((size_t)num_local_blocks * num_dof_types);
   for (int i = 0; i < num_dof_types; i++)
      for (int j = 0; j < num_local_blocks; j++) h->dofmap[(size_t)i * num_local_blocks + j] = i;
   return g_err;
}
extern "C" uint32_t HYPREDRV_LinearSystemPrintDofmap(HYPREDRV_t h, const char *filename)
{
   CHECK_INIT_OBJ(h);
   if (!filename) return err_set(ERR_INVALID_VAL, "PrintDofmap: NULL file name");
   char suffix[16];
   snprintf(suffix, sizeof(suffix), ".%05d", h->mypid);
   FILE *f = fopen((std::string(filename) + suffix).c_str(), "w");
   if (!f) return err_set(ERR_FILE_NOT_FOUND, std::string("cannot write ") + filename);
   fprintf(f, "%zu\n", h->dofmap.size());
   for (int v : h->dofmap) fprintf(f, "%d\n", v);
   fclose(f);
   return g_err;
}
// rigid-body modes only matter to nodal coarsening / interpolation vectors (reference amg.c:600-660),
// which this build does not have: the call is accepted so drivers written for it run, the
// vectors are not used
extern "C" uint32_t HYPREDRV_LinearSystemSetNearNullSpace(HYPREDRV_t h, int, int, const HYPRE_Complex *) { CHECK_INIT_OBJ(h); return g_err; }
// Exact null-space modes (reference src/HYPREDRV.c:2281-2320 -> src/internal/linsys.c:533-625): modified Gram-Schmidt on the
// component blocks, linear dependence refused at 1e-12 relative, num_components == 0 clears.  Here the modes live in HBM and
// the inner products are the library's deterministic device reductions (all-reduced over the ranks of a row partition).
extern "C" uint32_t HYPREDRV_LinearSystemSetNullSpace(HYPREDRV_t h, int num_entries, int num_components, const HYPRE_Complex *values)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (num_components == 0)
   {
      h->ns_modes.release();
      h->num_ns = 0;
      return g_err;
   }
   if (!h->mat_A) return err_set(ERR_INVALID_VAL, "The matrix must be set before calling HYPREDRV_LinearSystemSetNullSpace");
   if (num_components < 1 || num_entries < 0 || (!values && num_entries > 0))
      return err_set(ERR_INVALID_VAL, "Invalid null space input: need num_components >= 1, num_entries >= 0, and non-NULL values when num_entries > 0");
   if (num_entries != h->mat_A->nloc)
      return err_set(ERR_INVALID_VAL, "Null space modes need one entry per locally owned row (" + std::to_string(h->mat_A->nloc) + "), got " + std::to_string(num_entries));
   h->num_ns = 0;
   const int      n = num_entries;
   DArray<double> z((size_t)std::max((size_t)n * (size_t)num_components, (size_t)1));
   if (n) z.upload(values, (size_t)n * (size_t)num_components);
   auto ip = [&](const double *a, const double *b) {
      dot(n, a, b, 0);
      finalize(0, S_TMP);
      return read_scalar(S_TMP);
   };
   for (int k = 0; k < num_components; k++)
   {
      double      *zk = z.data() + (size_t)k * (size_t)n;
      const double norm_orig = std::sqrt(ip(zk, zk));
      for (int j = 0; j < k; j++)
      {
         const double *zj = z.data() + (size_t)j * (size_t)n;
         axpy(n, -ip(zk, zj), zj, zk);
      }
      const double nrm = std::sqrt(ip(zk, zk));
      if (nrm <= 1.0e-12 * norm_orig) // HYPREDRV_NULLSPACE_DEP_TOL (linsys.c:32)
      {
         char buf[160];
         snprintf(buf, sizeof(buf), "Null space modes must be linearly independent (mode %d has relative norm %e after orthogonalization)", k,
                  norm_orig > 0.0 ? nrm / norm_orig : 0.0);
         return err_set(ERR_INVALID_VAL, buf);
      }
      scale(n, 1.0 / nrm, zk);
   }
   h->ns_modes = std::move(z);
   h->num_ns   = num_components;
   h->ns_nloc  = n;
   h->ns_lower = h->mat_A->ilower;
   h->ns_upper = h->mat_A->iupper;
   API_CATCH
}
// hypredrv_LinearSystemProjectOutNullSpace (src/internal/linsys.c:637-752): x -= sum_k <x, z_k> z_k with all inner products taken
// of the solution as the solver left it; a system of another size or distribution is refused on every rank together
static void project_out_null_space(hypredrv_struct *h)
{
   if (h->num_ns < 1 || !h->vec_x) return;
   long long mismatch[1] = {(h->vec_x->jlower != h->ns_lower || h->vec_x->jupper != h->ns_upper || h->vec_x->nloc != h->ns_nloc) ? 1 : 0};
   Comm::world().allreduce_host(mismatch, 1, 1);
   if (mismatch[0])
   {
      err_set(ERR_INVALID_VAL, "Null space modes are incompatible with the current linear system; call HYPREDRV_LinearSystemSetNullSpace() again "
                               "(or clear the modes with num_components = 0) after changing the system size or distribution");
      return;
   }
   HDA_REQUIRE(h->num_ns <= Context::kNumSlots && S_GMRES + h->num_ns <= Context::kNumScalars, "too many null space modes for the reduction scratch");
   h->vec_x->ensure_device();
   const int n = h->ns_nloc;
   for (int k = 0; k < h->num_ns; k++) dot(n, h->vec_x->data(), h->ns_modes.data() + (size_t)k * (size_t)n, k);
   finalize_n(0, h->num_ns, S_GMRES);
   for (int k = 0; k < h->num_ns; k++) axpy_dev(n, S_GMRES + k, -1.0, h->ns_modes.data() + (size_t)k * (size_t)n, h->vec_x->data());
}
// State vectors (reference src/HYPREDRV.c state-vector block, include/HYPREDRV.h:1521-1695): the
// time levels of a nonlinear / transient driver.  The drivers create them HOST-initialised and
// read and write them through raw pointers between solves, so they live in host memory here: the
// storage is the IJVector's host stage, the logical -> physical mapping rotates, and the Newton
// correction is the device solution copied back once per call.
extern "C" uint32_t HYPREDRV_StateVectorSet(HYPREDRV_t h, int nstates, HYPRE_IJVector *vecs)
{
   CHECK_INIT_OBJ(h);
   if (nstates <= 0 || !vecs) return err_set(ERR_INVALID_VAL, "StateVectorSet: need at least one vector");
   h->state.clear();
   for (int s = 0; s < nstates; s++)
   {
      HYPRE_IJVector v = vecs[s];
      if (!v) return err_set(ERR_INVALID_VAL, "StateVectorSet: NULL vector in the list");
      if (v->stage.size() != (size_t)v->nloc) v->stage.assign((size_t)v->nloc, 0.0);
      h->state.push_back(v);
   }
   h->state_first = 0;
   return g_err;
}
static HYPRE_IJVector state_at(hypredrv_struct *h, int index)
{
   const int n = (int)h->state.size();
   if (n == 0 || index < 0 || index >= n) return nullptr;
   return h->state[(size_t)((h->state_first + index) % n)];
}
extern "C" uint32_t HYPREDRV_StateVectorGetValues(HYPREDRV_t h, int index, HYPRE_Complex **data_ptr)
{
   CHECK_INIT_OBJ(h);
   HYPRE_IJVector v = state_at(h, index);
   if (!v || !data_ptr) return err_set(ERR_INVALID_VAL, "StateVectorGetValues: index outside the states set with StateVectorSet");
   *data_ptr = v->stage.data();
   return g_err;
}
extern "C" uint32_t HYPREDRV_StateVectorCopy(HYPREDRV_t h, int index_in, int index_out)
{
   CHECK_INIT_OBJ(h);
   HYPRE_IJVector a = state_at(h, index_in), b = state_at(h, index_out);
   if (!a || !b || a->nloc != b->nloc) return err_set(ERR_INVALID_VAL, "StateVectorCopy: bad indices or incompatible sizes");
   if (a != b) std::copy(a->stage.begin(), a->stage.end(), b->stage.begin());
   return g_err;
}
extern "C" uint32_t HYPREDRV_StateVectorUpdateAll(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   if (h->state.empty()) return err_set(ERR_INVALID_VAL, "StateVectorUpdateAll: no state vectors set");
   h->state_first = (h->state_first + 1) % (int)h->state.size(); // logical 0 <- what was logical 1, ...
   return g_err;
}
extern "C" uint32_t HYPREDRV_StateVectorApplyCorrection(HYPREDRV_t h, int state_idx)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   HYPRE_IJVector v = state_at(h, state_idx);
   if (!v) return err_set(ERR_INVALID_VAL, "StateVectorApplyCorrection: index outside the states set with StateVectorSet");
   if (!h->vec_x || h->vec_x->nloc != v->nloc) return err_set(ERR_UNKNOWN, "StateVectorApplyCorrection: no solution vector of matching size");
   std::vector<double> dx((size_t)std::max(v->nloc, 1));
   download_sync(dx.data(), h->vec_x->data(), sizeof(double) * (size_t)v->nloc);
   for (int i = 0; i < v->nloc; i++) v->stage[(size_t)i] += dx[(size_t)i];
   API_CATCH
}
static std::string level_region(const char *name, int id) { return id >= 0 ? std::string(name) + "-" + std::to_string(id) : std::string(name); }
extern "C" uint32_t HYPREDRV_AnnotateLevelBegin(HYPREDRV_t h, int level, const char *name, int id)
{
   CHECK_INIT_OBJ(h);
   if (!name) return err_set(ERR_UNKNOWN_TIMING);
   if (level < 0 || level >= kStatsMaxLevels) return err_set(ERR_INVALID_VAL, "Annotation level " + std::to_string(level) + " out of range [0, 10)");
   Stats &s = h->stats;
   if (s.level_open[level]) return err_set(ERR_INVALID_VAL, "Level " + std::to_string(level) + " already has active annotation '" + s.level_name[level] + "'");
   s.level_open[level]        = true;
   s.level_name[level]        = level_region(name, id);
   s.level_current_id[level]++;
   s.level_solve_start[level] = s.solved_entries();
   for (int c = level + 1; c < kStatsMaxLevels; c++) s.level_current_id[c] = 0; // child ids are local to the parent
   return g_err;
}
extern "C" uint32_t HYPREDRV_AnnotateLevelEnd(HYPREDRV_t h, int level, const char *name, int id)
{
   CHECK_INIT_OBJ(h);
   if (!name) return err_set(ERR_UNKNOWN_TIMING);
   if (level < 0 || level >= kStatsMaxLevels) return err_set(ERR_INVALID_VAL, "Annotation level " + std::to_string(level) + " out of range [0, 10)");
   Stats &s = h->stats;
   if (!s.level_open[level]) return g_err; // an end without a begin is a no-op (reference stats.c:1081-1086)
   const std::string region = level_region(name, id);
   if (region != s.level_name[level])
      return err_set(ERR_INVALID_VAL, "Level " + std::to_string(level) + " annotation mismatch: expected '" + s.level_name[level] + "', got '" + region + "'");
   s.level_entries[level].push_back({s.level_current_id[level], s.level_solve_start[level], s.solved_entries()});
   s.level_open[level] = false;
   s.level_name[level].clear();
   return g_err;
}
static void level_totals(const Stats &s, const LevelEntry &e, int *nsolves, int *iters, double *setup, double *solve)
{
   int    q = 0, n = 0, it = 0;
   double ps = 0.0, ss = 0.0;
   for (const StatEntry &x : s.entries)
   {
      if (!x.has_solve) continue;
      if (q >= e.solve_start && q < e.solve_end) { n++; it += x.iters; ps += x.prec; ss += x.solve; }
      q++;
   }
   if (nsolves) *nsolves = n;
   if (iters) *iters = it;
   if (setup) *setup = ps;
   if (solve) *solve = ss;
}
extern "C" uint32_t HYPREDRV_StatsLevelGetCount(HYPREDRV_t h, int level, int *count)
{
   CHECK_INIT_OBJ(h);
   if (level < 0 || level >= kStatsMaxLevels) return err_set(ERR_INVALID_VAL, "level out of range");
   if (count) *count = (int)h->stats.level_entries[level].size();
   return g_err;
}
extern "C" uint32_t HYPREDRV_StatsLevelGetEntry(HYPREDRV_t h, int level, int index, int *entry_id, int *num_solves, int *linear_iters,
                                                double *setup_time, double *solve_time)
{
   CHECK_INIT_OBJ(h);
   if (level < 0 || level >= kStatsMaxLevels || index < 0 || index >= (int)h->stats.level_entries[level].size())
      return err_set(ERR_UNKNOWN, "StatsLevelGetEntry: invalid level " + std::to_string(level) + " or index " + std::to_string(index));
   const LevelEntry &e = h->stats.level_entries[level][(size_t)index];
   if (entry_id) *entry_id = e.id;
   level_totals(h->stats, e, num_solves, linear_iters, setup_time, solve_time);
   return g_err;
}
// text of the reference's summary: src/internal/stats.c:1750-1768
extern "C" uint32_t HYPREDRV_StatsLevelPrint(HYPREDRV_t h, int level)
{
   CHECK_INIT_OBJ(h);
   if (level < 0 || level >= kStatsMaxLevels || h->mypid) return g_err;
   const Stats &s     = h->stats;
   const int    count = (int)s.level_entries[level].size();
   if (!count) return g_err;
   long long tsolves = 0, tlin = 0;
   double    tsetup = 0.0, tsolve = 0.0;
   for (const LevelEntry &e : s.level_entries[level])
   {
      int    n = 0, it = 0;
      double ps = 0.0, ss = 0.0;
      level_totals(s, e, &n, &it, &ps, &ss);
      tsolves += n; tlin += it; tsetup += ps; tsolve += ss;
   }
   const double ai = tsolves ? (double)tlin / (double)tsolves : 0.0, as = tsolves ? tsetup / (double)tsolves : 0.0,
                av = tsolves ? tsolve / (double)tsolves : 0.0;
   printf("\n");
   printf("Aggregate Summary:\n");
   printf("--------------------------------------------------------------\n");
   printf("Total number of Non-linear iterations: %lld\n", tsolves);
   printf("Total number of linear iterations:     %lld\n", tlin);
   printf("Avg. LS iterations:                    %.2f\n", ai);
   printf("Avg. LS times: (setup, solve, total):  %.4f, %.4f, %.4f\n", as, av, as + av);
   printf("Total LS times: (setup, solve, total): %.4f, %.4f, %.4f\n", tsetup, tsolve, tsetup + tsolve);
   printf("Avg. LS iterations per timestep:       %.2f\n", (double)tlin / count);
   printf("Avg. LS times per timestep: (s, s, t): %.4f, %.4f, %.4f\n", tsetup / count, tsolve / count, (tsetup + tsolve) / count);
   printf("--------------------------------------------------------------\n");
   printf("\n");
   return g_err;
}
extern "C" uint32_t HYPREDRV_LinearSystemReadDofmap(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   // hypredrv_IntArrayParRead (reference src/internal/containers.c:443-620): parts prefix.%05d[.bin], a count then the labels
   // (ASCII), or a size_t count then int32 labels (binary); the parts are dealt to the ranks in order
   if (h->args.ls.dofmap_filename.empty() && h->args.ls.dofmap_basename.empty()) return g_err;
   std::string prefix = ls_path(h, h->args.ls.dofmap_filename, h->args.ls.dofmap_basename);
   auto exists = [](const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; };
   char buf[64];
   snprintf(buf, sizeof buf, ".%05d.bin", 0);
   const bool binary = exists(prefix + buf);
   int        gparts = 0;
   for (;; gparts++)
   {
      snprintf(buf, sizeof buf, binary ? ".%05d.bin" : ".%05d", gparts);
      if (!exists(prefix + buf)) break;
   }
   if (gparts < h->nprocs) return err_set(ERR_FILE_UNEXPECTED_ENTRY, "Invalid dofmap filename \"" + prefix + "\" or invalid number of parts!");
   int nparts = gparts / h->nprocs + (h->mypid < gparts % h->nprocs ? 1 : 0);
   int first  = h->mypid * (gparts / h->nprocs) + std::min(h->mypid, gparts % h->nprocs);
   h->dofmap.clear();
   for (int part = first; part < first + nparts; part++)
   {
      snprintf(buf, sizeof buf, binary ? ".%05d.bin" : ".%05d", part);
      FILE *f = fopen((prefix + buf).c_str(), binary ? "rb" : "r");
      if (!f) return err_set(ERR_FILE_NOT_FOUND, "cannot open " + prefix + buf);
      size_t n = 0, got = 0;
      if ((binary ? fread(&n, sizeof(size_t), 1, f) : (size_t)fscanf(f, "%zu", &n)) != 1)
      {
         fclose(f);
         return err_set(ERR_FILE_UNEXPECTED_ENTRY, "Invalid number of header entries!");
      }
      const size_t at = h->dofmap.size();
      h->dofmap.resize(at + n);
      if (binary) got = fread(h->dofmap.data() + at, sizeof(int), n, f);
      else
         while (got < n && fscanf(f, "%d", &h->dofmap[at + got]) == 1) got++;
      fclose(f);
      if (got != n) return err_set(ERR_FILE_UNEXPECTED_ENTRY, "Expected " + std::to_string(n) + ", but found " + std::to_string(got) + " coefficients!");
   }
   return g_err;
}
// seam for bench.py (hda_borrow_hypredrv): the objects the API built, for the kernel-level measurement entries
namespace hda {
bool hypredrv_peek(void *obj, const DCsr **A, const HaloPlan **halo, const double **rhs, Amg **amg)
{
   hypredrv_struct *h = (hypredrv_struct *)obj;
   if (!h || !h->mat_A) return false;
   if (A) *A = &h->mat_A->A;
   if (halo) *halo = &h->mat_A->halo;
   if (rhs) *rhs = h->vec_b ? (h->vec_b->ensure_device(), h->vec_b->data()) : nullptr;
   if (amg) *amg = (h->precon && h->precon_is_setup && h->precon->kind == HDA_SOLVER_AMG) ? h->precon->amg.get() : nullptr;
   return true;
}
} // namespace hda
// levels of the set-up BoomerAMG hierarchy that are row partitioned (0 on one rank or before Setup): bench.py's line, thread-rank tests
extern "C" int hda_amd_partitioned_levels(void *obj)
{
   hypredrv_struct *h = (hypredrv_struct *)obj;
   if (!h || !h->precon || !h->precon_is_setup || !h->precon->amg) return 0;
   return h->precon->amg->partitioned_levels();
}
extern "C" int hda_amd_hierarchy_levels(void *obj) // partitioned levels + the levels of the replicated tail
{
   hypredrv_struct *h = (hypredrv_struct *)obj;
   if (!h || !h->precon || !h->precon_is_setup || !h->precon->amg) return 0;
   return h->precon->amg->total_levels();
}
// bytes THIS rank streams per Krylov iteration (operator product + vector updates) and per
// V-cycle with the hierarchy that was set up: [0] CSR figures of SURVEY 8(d), [1] the formats
// actually read (coded operators).  bench.py sums them over the ranks.
extern "C" uint32_t HYPREDRV_AMD_SolvePhaseBytes(HYPREDRV_t h, double iteration[2], double vcycle[2])
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (!h->mat_A || !h->precon || !h->precon->amg) return err_set(ERR_UNKNOWN, "SolvePhaseBytes: set up the solver first");
   for (int f = 0; f < 2; f++)
   {
      iteration[f] = pcg_iteration_bytes(h->mat_A->A, f == 1);
      vcycle[f]    = h->precon->amg->vcycle_bytes(f == 1);
   }
   API_CATCH
}
// Arms (arm != 0) the launch-timing probe on the Jacobi sweeps of this rank's largest operator kept
// in plain CSR (level 0 is usually stencil-coded and cheaper) and returns its level and local sizes;
// arm == 0 reads the average launch duration back and disarms.  bench.py's N > 1 roofline object.
extern "C" uint32_t HYPREDRV_AMD_ProbeDominant(HYPREDRV_t h, int arm, int *level, double dims[3], double *avg_ms, int *count)
{
   CHECK_INIT_OBJ(h);
   API_TRY
   if (!h->mat_A || !h->precon || !h->precon->amg) return err_set(ERR_UNKNOWN, "ProbeDominant: set up the solver first");
   Amg &amg = *h->precon->amg;
   if (arm)
   {
      int    best = 0;
      double bn   = -1.0;
      for (int l = 0; l + 1 < amg.num_levels(); l++)
      {
         const DCsr &M = amg.level_A(l);
         spmv_prepare(M);
         const double w = M.coded == 1 ? 0.0 : (double)M.nnz;
         if (w > bn) { bn = w; best = l; }
      }
      const DCsr &M = amg.level_A(best);
      if (level) *level = best;
      if (dims) { dims[0] = M.nrows; dims[1] = M.ncols; dims[2] = M.nnz; }
      spmv_probe_set(&M, 2);
   }
   else
   {
      spmv_probe_read(avg_ms, count);
      spmv_probe_set(nullptr, -1);
   }
   API_CATCH
}
extern "C" uint32_t HYPREDRV_LinearSystemComputeEigenspectrum(HYPREDRV_t h) { CHECK_INIT_OBJ(h); return g_err; } // no-op unless built with eigspec

// ------------------------------------------------------------- THE HOT PATH

// hypredrv_PreconReuseShouldRebuild, static policy (reference src/internal/precon_reuse.c:774-827), and the
// collective form of src/HYPREDRV.c:233-256: every rank takes the maximum of the local decisions
static bool reuse_should_rebuild(hypredrv_struct *h)
{
   const ReuseArgs &a  = h->args.reuse;
   const int        id = std::max(h->stats.systems_solved, 0); // = StatsGetLinearSystemID + 1
   int              rebuild;
   if (a.enabled && !a.linear_system_ids.empty())
      rebuild = std::find(a.linear_system_ids.begin(), a.linear_system_ids.end(), id) != a.linear_system_ids.end() ? 1 : 0;
   else
   {
      const int freq = (a.enabled && a.frequency > 0) ? a.frequency : 0;
      rebuild        = (id % (freq + 1)) == 0 ? 1 : 0;
   }
   long long v[1] = {rebuild};
   Comm::world().allreduce_host(v, 1, 1); // max
   return v[0] != 0;
}

// the reference's solver_ops table (src/internal/solver.c:204-253), indexed by the kind of the Krylov handle
struct SolverOps {
   HYPRE_Int (*set_precond)(HYPRE_Solver, HYPRE_PtrToSolverFcn, HYPRE_PtrToSolverFcn, HYPRE_Solver);
   HYPRE_Int (*setup)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector);
   HYPRE_Int (*solve)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector);
   HYPRE_Int (*destroy)(HYPRE_Solver);
   HYPRE_Int (*get_num_iterations)(HYPRE_Solver, HYPRE_Int *);
   HYPRE_Int (*get_converged)(HYPRE_Solver, HYPRE_Int *);
   HYPRE_Int (*get_final_rel_res_norm)(HYPRE_Solver, HYPRE_Real *);
};
static const SolverOps &solver_ops(const HYPRE_Solver s)
{
   static const SolverOps pcg = {HYPRE_PCGSetPrecond, HYPRE_ParCSRPCGSetup, HYPRE_ParCSRPCGSolve, HYPRE_ParCSRPCGDestroy,
                                 HYPRE_PCGGetNumIterations, HYPRE_PCGGetConverged, HYPRE_PCGGetFinalRelativeResidualNorm};
   static const SolverOps gmres = {HYPRE_GMRESSetPrecond, HYPRE_ParCSRGMRESSetup, HYPRE_ParCSRGMRESSolve, HYPRE_ParCSRGMRESDestroy,
                                   HYPRE_GMRESGetNumIterations, HYPRE_GMRESGetConverged, HYPRE_GMRESGetFinalRelativeResidualNorm};
   static const SolverOps fgmres = {HYPRE_FlexGMRESSetPrecond, HYPRE_ParCSRFlexGMRESSetup, HYPRE_ParCSRFlexGMRESSolve,
                                    HYPRE_ParCSRFlexGMRESDestroy, HYPRE_FlexGMRESGetNumIterations, HYPRE_FlexGMRESGetConverged,
                                    HYPRE_FlexGMRESGetFinalRelativeResidualNorm};
   static const SolverOps bicgstab = {HYPRE_BiCGSTABSetPrecond, HYPRE_ParCSRBiCGSTABSetup, HYPRE_ParCSRBiCGSTABSolve,
                                      HYPRE_ParCSRBiCGSTABDestroy, HYPRE_BiCGSTABGetNumIterations, hypre_BiCGSTABGetConverged,
                                      HYPRE_BiCGSTABGetFinalRelativeResidualNorm};
   switch (s->kind)
   {
      case HDA_SOLVER_GMRES: return gmres;
      case HDA_SOLVER_FGMRES: return fgmres;
      case HDA_SOLVER_BICGSTAB: return bicgstab;
      default: return pcg;
   }
}

// reference src/internal/solver.c:268-311: the Krylov setup calls back here; the "prec"
// timer brackets exactly the AMG setup
static HYPRE_Int PreconSetupDispatch(HYPRE_Solver cookie, HYPRE_Matrix A, HYPRE_Vector b, HYPRE_Vector x)
{
   hypredrv_struct *h = ((PreconCookie *)(void *)cookie)->self;
   annotate(h, "prec", true);
   HYPRE_Int ierr = (h->precon->kind == HDA_SOLVER_ILU)   ? HYPRE_ILUSetup(h->precon, A, b, x)
                    : (h->precon->kind == HDA_SOLVER_MGR) ? HYPRE_MGRSetup(h->precon, A, b, x)
                                                          : HYPRE_BoomerAMGSetup(h->precon, A, b, x);
   if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
   annotate(h, "prec", false);
   h->precon_is_setup = (ierr == 0);
   return ierr;
}
// reference src/internal/solver.c:314-329
static HYPRE_Int PreconSolveDispatch(HYPRE_Solver cookie, HYPRE_Matrix A, HYPRE_Vector b, HYPRE_Vector x)
{
   hypredrv_struct *h = ((PreconCookie *)(void *)cookie)->self;
   return (h->precon->kind == HDA_SOLVER_ILU)   ? HYPRE_ILUSolve(h->precon, A, b, x)
          : (h->precon->kind == HDA_SOLVER_MGR) ? HYPRE_MGRSolve(h->precon, A, b, x)
                                                : HYPRE_BoomerAMGSolve(h->precon, A, b, x);
}

// hypredrv_ILUCreate (reference src/internal/ilu.c:63-115): same setter sequence
static void ilu_create(const IluArgs &a, HYPRE_Solver *out)
{
   HYPRE_Solver p = nullptr;
   HYPRE_ILUCreate(&p);
   HYPRE_ILUSetType(p, a.type);
   HYPRE_ILUSetLevelOfFill(p, a.fill_level);
   HYPRE_ILUSetLocalReordering(p, a.reordering);
   HYPRE_ILUSetTriSolve(p, a.tri_solve);
   HYPRE_ILUSetLowerJacobiIters(p, a.lower_jac_iters);
   HYPRE_ILUSetUpperJacobiIters(p, a.upper_jac_iters);
   HYPRE_ILUSetPrintLevel(p, a.print_level);
   HYPRE_ILUSetMaxIter(p, a.max_iter);
   HYPRE_ILUSetTol(p, a.tolerance);
   HYPRE_ILUSetMaxNnzPerRow(p, a.max_row_nnz);
   HYPRE_ILUSetDropThreshold(p, a.droptol);
   const bool schur = a.type == 10 || a.type == 11 || a.type == 20 || a.type == 21 || a.type == 40 || a.type == 41 || a.type == 50;
   if (schur) HYPRE_ILUSetSchurMaxIter(p, a.schur_max_iter);
   if (a.type == 20 || a.type == 21) HYPRE_ILUSetNSHDropThreshold(p, a.nsh_droptol);
   *out = p;
}

// hypredrv_AMGCreate (reference src/internal/amg.c:864-1035): same setter sequence
static void amg_create(const AmgArgs &a, HYPRE_Solver *out)
{
   HYPRE_Solver p = nullptr;
   HYPRE_BoomerAMGCreate(&p);
   HYPRE_BoomerAMGSetInterpType(p, a.prolongation_type);
   HYPRE_BoomerAMGSetRestriction(p, a.restriction_type);
   HYPRE_BoomerAMGSetStrongThresholdR(p, a.restrict_strong_th);
   HYPRE_BoomerAMGSetFilterThresholdR(p, a.restrict_filter_th);
   HYPRE_BoomerAMGSetCoarsenType(p, a.type);
   HYPRE_BoomerAMGSetSabs(p, a.sabs);
   HYPRE_BoomerAMGSetTol(p, a.tolerance);
   HYPRE_BoomerAMGSetStrongThreshold(p, a.strong_th);
   HYPRE_BoomerAMGSetSeqThreshold(p, a.seq_amg_th);
   HYPRE_BoomerAMGSetMaxCoarseSize(p, a.max_coarse_size);
   HYPRE_BoomerAMGSetMinCoarseSize(p, a.min_coarse_size);
   HYPRE_BoomerAMGSetTruncFactor(p, a.trunc_factor);
   HYPRE_BoomerAMGSetPMaxElmts(p, a.max_nnz_row);
   HYPRE_BoomerAMGSetPrintLevel(p, a.print_level);
   if (a.relax_type >= 0) HYPRE_BoomerAMGSetRelaxType(p, a.relax_type);
   HYPRE_BoomerAMGSetRelaxOrder(p, a.order);
   HYPRE_BoomerAMGSetRelaxWt(p, a.weight);
   HYPRE_BoomerAMGSetOuterWt(p, a.outer_weight);
   HYPRE_BoomerAMGSetMaxLevels(p, a.max_levels);
   HYPRE_BoomerAMGSetChebyOrder(p, a.cheby_order); // amg.c:886-890
   HYPRE_BoomerAMGSetChebyFraction(p, a.cheby_fraction);
   HYPRE_BoomerAMGSetChebyEigEst(p, a.cheby_eig_est);
   HYPRE_BoomerAMGSetChebyVariant(p, a.cheby_variant);
   HYPRE_BoomerAMGSetChebyScale(p, a.cheby_scale);
   HYPRE_BoomerAMGSetSmoothType(p, a.smooth_type);
   HYPRE_BoomerAMGSetSmoothNumSweeps(p, a.smooth_num_sweeps);
   HYPRE_BoomerAMGSetSmoothNumLevels(p, a.smooth_num_levels);
   HYPRE_BoomerAMGSetMaxRowSum(p, a.max_row_sum);
   HYPRE_BoomerAMGSetILUType(p, a.smooth_ilu.type);
   HYPRE_BoomerAMGSetILULocalReordering(p, a.smooth_ilu.reordering);
   HYPRE_BoomerAMGSetILUTriSolve(p, a.smooth_ilu.tri_solve);
   HYPRE_BoomerAMGSetILULowerJacobiIters(p, a.smooth_ilu.lower_jac_iters);
   HYPRE_BoomerAMGSetILUUpperJacobiIters(p, a.smooth_ilu.upper_jac_iters);
   HYPRE_BoomerAMGSetILULevel(p, a.smooth_ilu.fill_level);
   HYPRE_BoomerAMGSetILUDroptol(p, a.smooth_ilu.droptol);
   HYPRE_BoomerAMGSetILUMaxRowNnz(p, a.smooth_ilu.max_row_nnz);
   HYPRE_BoomerAMGSetILUMaxIter(p, a.smooth_num_sweeps);
   HYPRE_BoomerAMGSetNumFunctions(p, a.nodal ? 3 : a.num_functions);
   HYPRE_BoomerAMGSetFilterFunctions(p, a.filter_functions);
   HYPRE_BoomerAMGSetAggNumLevels(p, a.agg_num_levels);
   HYPRE_BoomerAMGSetAggInterpType(p, a.agg_prolongation_type);
   HYPRE_BoomerAMGSetAggTruncFactor(p, a.agg_trunc_factor);
   HYPRE_BoomerAMGSetAggP12TruncFactor(p, a.agg_P12_trunc_factor);
   HYPRE_BoomerAMGSetAggPMaxElmts(p, a.agg_max_nnz_row);
   HYPRE_BoomerAMGSetAggP12MaxElmts(p, (int)a.agg_P12_max_elements);
   HYPRE_BoomerAMGSetNumPaths(p, a.agg_num_paths);
   HYPRE_BoomerAMGSetMaxIter(p, a.max_iter);
   HYPRE_BoomerAMGSetRAP2(p, a.rap2);
   HYPRE_BoomerAMGSetModuleRAP2(p, a.mod_rap2);
   HYPRE_BoomerAMGSetKeepTranspose(p, a.keep_transpose);
   HYPRE_BoomerAMGSetCycleRelaxType(p, a.down_type, 1);
   HYPRE_BoomerAMGSetCycleNumSweeps(p, a.down_sweeps > -1 ? a.down_sweeps : a.num_sweeps, 1);
   HYPRE_BoomerAMGSetCycleRelaxType(p, a.up_type, 2);
   HYPRE_BoomerAMGSetCycleNumSweeps(p, a.up_sweeps > -1 ? a.up_sweeps : a.num_sweeps, 2);
   HYPRE_BoomerAMGSetCycleRelaxType(p, a.coarse_type, 3);
   HYPRE_BoomerAMGSetCycleNumSweeps(p, a.coarse_sweeps > -1 ? a.coarse_sweeps : a.num_sweeps, 3);
   *out = p;
}

// hypredrv_MGRCreate (reference src/internal/mgr.c:3449-3900): C points of every reduction level by dof label, the
// per-level option arrays, BoomerAMG as coarse solver -- handed over through hypre's own MGR calls
// A Krylov solver handle from its arguments (hypredrv_PCGCreate / GMRESCreate / FGMRESCreate / BiCGSTABCreate, reference
// src/internal/pcg.c, gmres.c, fgmres.c:36-48, bicgstab.c:41-55): the outer solver and the nested ones of MGR components
static bool krylov_handle_create(MPI_Comm comm, const KrylovArgs &k, HYPRE_Solver *out)
{
   if (k.method == 0)
   {
      HYPRE_ParCSRPCGCreate(comm, out);
      HYPRE_PCGSetMaxIter(*out, k.max_iter);
      HYPRE_PCGSetTwoNorm(*out, k.two_norm);
      HYPRE_PCGSetStopCrit(*out, k.stop_crit);
      HYPRE_PCGSetRelChange(*out, k.rel_change);
      HYPRE_PCGSetPrintLevel(*out, k.print_level);
      HYPRE_PCGSetRecomputeResidual(*out, k.recompute_res);
      HYPRE_PCGSetTol(*out, k.relative_tol);
      HYPRE_PCGSetAbsoluteTol(*out, k.absolute_tol);
      HYPRE_PCGSetResidualTol(*out, k.residual_tol);
      HYPRE_PCGSetConvergenceFactorTol(*out, k.conv_fac_tol);
   }
   else if (k.method == 1)
   {
      HYPRE_ParCSRGMRESCreate(comm, out);
      HYPRE_GMRESSetMinIter(*out, k.min_iter);
      HYPRE_GMRESSetMaxIter(*out, k.max_iter);
      HYPRE_GMRESSetStopCrit(*out, k.stop_crit);
      HYPRE_GMRESSetSkipRealResidualCheck(*out, k.skip_real_res_check);
      HYPRE_GMRESSetKDim(*out, k.krylov_dim);
      HYPRE_GMRESSetRelChange(*out, k.rel_change);
      HYPRE_GMRESSetLogging(*out, k.logging);
      HYPRE_GMRESSetPrintLevel(*out, k.print_level);
      HYPRE_GMRESSetTol(*out, k.relative_tol);
      HYPRE_GMRESSetAbsoluteTol(*out, k.absolute_tol);
      HYPRE_GMRESSetConvergenceFactorTol(*out, k.conv_fac_tol);
   }
   else if (k.method == 2)
   { // hypredrv_FGMRESCreate (reference src/internal/fgmres.c:36-48)
      HYPRE_ParCSRFlexGMRESCreate(comm, out);
      HYPRE_FlexGMRESSetMinIter(*out, k.min_iter);
      HYPRE_FlexGMRESSetMaxIter(*out, k.max_iter);
      HYPRE_FlexGMRESSetKDim(*out, k.krylov_dim);
      HYPRE_FlexGMRESSetLogging(*out, k.logging);
      HYPRE_FlexGMRESSetPrintLevel(*out, k.print_level);
      HYPRE_FlexGMRESSetTol(*out, k.relative_tol);
      HYPRE_FlexGMRESSetAbsoluteTol(*out, k.absolute_tol);
   }
   else if (k.method == 3)
   { // hypredrv_BiCGSTABCreate (reference src/internal/bicgstab.c:41-55)
      HYPRE_ParCSRBiCGSTABCreate(comm, out);
      HYPRE_BiCGSTABSetMinIter(*out, k.min_iter);
      HYPRE_BiCGSTABSetMaxIter(*out, k.max_iter);
      HYPRE_BiCGSTABSetStopCrit(*out, k.stop_crit);
      HYPRE_BiCGSTABSetLogging(*out, k.logging);
      HYPRE_BiCGSTABSetPrintLevel(*out, k.print_level);
      HYPRE_BiCGSTABSetTol(*out, k.relative_tol);
      HYPRE_BiCGSTABSetAbsoluteTol(*out, k.absolute_tol);
      HYPRE_BiCGSTABSetConvergenceFactorTol(*out, k.conv_fac_tol);
   }
   else
      return false;
   return true;
}

// A nested Krylov component of MGR (reference hypredrv_NestedKrylovCreate, src/internal/krylov.c:418-505): the Krylov handle and, when
// the block names one, the BoomerAMG / ILU handle installed as its preconditioner.  Both are owned by the HYPREDRV object.
static HYPRE_Solver nested_krylov_create(hypredrv_struct *h, const NestedKrylovArgs &nk)
{
   HYPRE_Solver ks = nullptr, ps = nullptr;
   krylov_handle_create(h->comm, nk.solver, &ks);
   if (nk.precon == 0) amg_create(nk.amg, &ps);
   else if (nk.precon == 2) ilu_create(nk.ilu, &ps);
   if (ps)
   {
      h->precon_aux.push_back(ps);
      HYPRE_PtrToSolverFcn solve = nk.precon == 0 ? (HYPRE_PtrToSolverFcn)HYPRE_BoomerAMGSolve : (HYPRE_PtrToSolverFcn)HYPRE_ILUSolve;
      HYPRE_PtrToSolverFcn setup = nk.precon == 0 ? (HYPRE_PtrToSolverFcn)HYPRE_BoomerAMGSetup : (HYPRE_PtrToSolverFcn)HYPRE_ILUSetup;
      switch (nk.solver.method)
      {
         case 0: HYPRE_PCGSetPrecond(ks, solve, setup, ps); break;
         case 1: HYPRE_GMRESSetPrecond(ks, solve, setup, ps); break;
         case 2: HYPRE_FlexGMRESSetPrecond(ks, solve, setup, ps); break;
         default: HYPRE_BiCGSTABSetPrecond(ks, solve, setup, ps); break;
      }
   }
   h->precon_aux.push_back(ks);
   return ks;
}

static uint32_t mgr_create(hypredrv_struct *h, const MgrArgs &a)
{
   if (h->dofmap.empty())
      return err_set(ERR_MISSING_DOFMAP, "MGR needs a dofmap (linear_system.dofmap_filename, HYPREDRV_LinearSystemSetDofmap or ...SetInterleavedDofmap)");
   if (a.level.empty()) return err_set(ERR_MISSING_KEY, "preconditioner.mgr.level: at least level 0 with its f_dofs is needed");
   if ((a.coarsest_type > 0 && a.coarsest_type != 32) || !a.coarsest_block.empty())
      return err_set(ERR_INVALID_PRECON | HYPREDRV_ERROR_UNSUPPORTED_AMD, "MGR coarsest_level: BoomerAMG and ILU are implemented on MI355X");
   const int nlev = (int)a.level.size();
   // labels present (every rank sees the same set in a well-formed dofmap; one rank here)
   std::vector<int> cur = h->dofmap;
   std::sort(cur.begin(), cur.end());
   cur.erase(std::unique(cur.begin(), cur.end()), cur.end());
   const int nlabels = cur.empty() ? 0 : cur.back() + 1;
   std::vector<std::vector<HYPRE_Int>> c_dofs((size_t)nlev);
   std::vector<HYPRE_Int>              num_c((size_t)nlev), frelax((size_t)nlev), fsweeps((size_t)nlev), interp((size_t)nlev), restr((size_t)nlev),
      coarse((size_t)nlev), gsm((size_t)nlev), git((size_t)nlev);
   std::vector<HYPRE_Int *> c_ptr((size_t)nlev);
   for (int l = 0; l < nlev; l++)
   {
      const MgrLevelArgs &L = a.level[(size_t)l];
      if (L.f_dofs.empty()) return err_set(ERR_MISSING_KEY, "preconditioner.mgr.level." + std::to_string(l) + ".f_dofs is missing");
      for (const std::string *b : {&L.f_block, &L.g_block})
         if (!b->empty())
            return err_set(ERR_INVALID_PRECON | HYPREDRV_ERROR_UNSUPPORTED_AMD,
                           "MGR level " + std::to_string(l) + ": a nested '" + *b + "' relaxation solver is not implemented on MI355X");
      for (int f : L.f_dofs)
         if (!std::binary_search(cur.begin(), cur.end(), f))
            return err_set(ERR_INVALID_VAL, "MGR level " + std::to_string(l) + ": f_dofs label " + std::to_string(f) + " is not an unknown of this level");
      for (int lab : cur)
         if (std::find(L.f_dofs.begin(), L.f_dofs.end(), lab) == L.f_dofs.end()) c_dofs[(size_t)l].push_back(lab);
      num_c[(size_t)l]   = (HYPRE_Int)c_dofs[(size_t)l].size();
      c_ptr[(size_t)l]   = c_dofs[(size_t)l].data();
      frelax[(size_t)l]  = L.f_type;
      fsweeps[(size_t)l] = L.f_type < 0 ? 0 : L.f_sweeps;
      interp[(size_t)l]  = L.prolongation_type;
      restr[(size_t)l]   = L.restriction_type;
      coarse[(size_t)l]  = L.coarse_level_type;
      gsm[(size_t)l]     = L.g_type;
      git[(size_t)l]     = L.g_type < 0 ? 0 : L.g_sweeps;
      cur.assign(c_dofs[(size_t)l].begin(), c_dofs[(size_t)l].end());
   }
   HYPRE_Solver p = nullptr;
   HYPRE_MGRCreate(&p);
   HYPRE_MGRSetCpointsByPointMarkerArray(p, nlabels, nlev, num_c.data(), c_ptr.data(), h->dofmap.data());
   HYPRE_MGRSetNonCpointsToFpoints(p, a.non_c_to_f);
   HYPRE_MGRSetPMaxElmts(p, a.pmax);
   HYPRE_MGRSetMaxIter(p, a.max_iter);
   HYPRE_MGRSetTol(p, a.tolerance);
   HYPRE_MGRSetPrintLevel(p, a.print_level);
   HYPRE_MGRSetCycleType(p, a.cycle);
   HYPRE_MGRSetFRelaxCycle(p, a.cycle_smooth_pos); // reference src/internal/mgr.c:3792-3794
   HYPRE_MGRSetGlobalSmoothCycle(p, a.cycle_smooth_pos);
   HYPRE_MGRSetTruncateCoarseGridThreshold(p, a.coarse_th);
   HYPRE_MGRSetRelaxType(p, a.relax_type);
   HYPRE_MGRSetLevelFRelaxType(p, frelax.data());
   HYPRE_MGRSetLevelNumRelaxSweeps(p, fsweeps.data());
   HYPRE_MGRSetLevelInterpType(p, interp.data());
   HYPRE_MGRSetLevelRestrictType(p, restr.data());
   HYPRE_MGRSetCoarseGridMethod(p, coarse.data());
   HYPRE_MGRSetLevelSmoothType(p, gsm.data());
   HYPRE_MGRSetLevelSmoothIters(p, git.data());
   HYPRE_MGRSetNonGalerkinMaxElmts(p, a.nonglk_max_elmts);
   for (int l = 0; l < nlev; l++)
   {
      const MgrLevelArgs &L = a.level[(size_t)l];
      if (L.f_krylov.set)
      { // f_relaxation by a nested Krylov solver (mgr.c:3938-3960): the handle goes in as the level's F-solver
         HYPRE_MGRSetFSolverAtLevel(p, nested_krylov_create(h, L.f_krylov), l);
      }
      else if (L.f_type == 2 || L.f_type == 32)
      { // f_relaxation amg / ilu: a solver handle for A_FF (reference mgr.c:2594, HYPRE_MGRSetFSolverAtLevel)
         HYPRE_Solver fs = nullptr;
         if (L.f_type == 2) amg_create(L.f_amg, &fs);
         else ilu_create(L.f_ilu, &fs);
         h->precon_aux.push_back(fs);
         HYPRE_MGRSetFSolverAtLevel(p, fs, l);
      }
      if (L.g_type == 16 && L.g_ilu_block)
      { // g_relaxation with its own ilu block: a smoother object (HYPRE_MGRSetGlobalSmootherAtLevel); the flat
        // "g_relaxation: ilu" is hypre's built-in type 16 with default ILU arguments
         HYPRE_Solver gs = nullptr;
         ilu_create(L.g_ilu, &gs);
         h->precon_aux.push_back(gs);
         HYPRE_MGRSetGlobalSmootherAtLevel(p, gs, l);
      }
   }
   HYPRE_Solver cs = nullptr;
   if (a.coarsest_krylov.set)
   { // coarsest_level by a nested Krylov solver (mgr.c:4253-4275); its handles are already on the owned list
      HYPRE_MGRSetCoarseSolver(p, HYPRE_ParCSRGMRESSolve, HYPRE_ParCSRGMRESSetup, nested_krylov_create(h, a.coarsest_krylov));
      h->precon = p;
      consume_hypre_errors();
      return g_err;
   }
   if (a.coarsest_type == 32)
   {
      ilu_create(a.coarsest_ilu, &cs);
      HYPRE_MGRSetCoarseSolver(p, HYPRE_ILUSolve, HYPRE_ILUSetup, cs);
   }
   else
   {
      amg_create(a.coarsest_amg, &cs);
      HYPRE_MGRSetCoarseSolver(p, HYPRE_BoomerAMGSolve, HYPRE_BoomerAMGSetup, cs);
   }
   h->precon_aux.push_back(cs);
   h->precon = p;
   consume_hypre_errors();
   return g_err;
}

extern "C" uint32_t HYPREDRV_PreconCreate(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   // src/HYPREDRV.c:2801-2808: an existing preconditioner is recreated only when the reuse policy says so
   if (h->precon && !reuse_should_rebuild(h)) return g_err;
   if (h->precon) { HYPRE_BoomerAMGDestroy(h->precon); h->precon = nullptr; }
   for (HYPRE_Solver a : h->precon_aux) HYPRE_BoomerAMGDestroy(a);
   h->precon_aux.clear();
   h->precon_is_setup  = false;
   const PreconArgs &p = h->args.precon();
   if (p.method == 99) return g_err; // none
   if (p.method == 2)
   {
      ilu_create(p.ilu, &h->precon);
      consume_hypre_errors();
      return g_err;
   }
   if (p.method == 1) return mgr_create(h, p.mgr);
   if (p.method != 0)
      return err_set(ERR_INVALID_PRECON | HYPREDRV_ERROR_UNSUPPORTED_AMD,
                     "preconditioner '" + p.method_name + "' is not implemented on MI355X yet (BoomerAMG, ILU and MGR only)");
   amg_create(p.amg, &h->precon);
   // hypredrv_AMGSetDofFunc (reference src/internal/amg.c:792-862): the dofmap names the function of
   // every local unknown when its labels fit [0, num_functions); otherwise hypre's interleaved default
   if (h->precon && p.amg.num_functions > 1 && !h->dofmap.empty())
   {
      bool fits = true;
      for (int v : h->dofmap) fits = fits && v >= 0 && v < p.amg.num_functions;
      long long bad[1] = {fits ? 0 : 1};
      Comm::world().allreduce_host(bad, 1, 1);
      if (bad[0] == 0)
      {
         if (h->mat_A && (int)h->dofmap.size() != h->mat_A->nloc)
            return err_set(ERR_INVALID_VAL, "Dofmap size (" + std::to_string(h->dofmap.size()) + ") does not match the number of local matrix rows (" +
                                               std::to_string(h->mat_A->nloc) + ")");
         HYPRE_BoomerAMGSetDofFunc(h->precon, h->dofmap.data());
      }
   }
   consume_hypre_errors();
   API_CATCH
}

// hypredrv_PCGCreate / hypredrv_GMRESCreate (reference src/internal/pcg.c:55-72, gmres.c:59-77)
extern "C" uint32_t HYPREDRV_LinearSolverCreate(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   // LinearSolverCreate creates the preconditioner too when the caller did not (src/HYPREDRV.c:2914-2917)
   // ... and re-evaluates the reuse policy when one exists and is set up (src/HYPREDRV.c:2905-2917)
   if ((!h->precon || h->precon_is_setup) && h->args.precon().method != 99)
   {
      if (HYPREDRV_PreconCreate(h)) return g_err;
   }
   if (h->solver) { solver_ops(h->solver).destroy(h->solver); h->solver = nullptr; }
   if (!krylov_handle_create(h->comm, h->args.solver, &h->solver)) return err_set(ERR_INVALID_SOLVER, "unknown solver method");
   consume_hypre_errors();
   API_CATCH
}

extern "C" uint32_t HYPREDRV_PreconSetup(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   if (!h->precon) return err_set(ERR_INVALID_PRECON, "PreconSetup: no preconditioner (call PreconCreate)");
   if (!h->mat_A || !h->vec_b || !h->vec_x) return err_set(ERR_UNKNOWN, "PreconSetup: linear system is incomplete");
   h->stats.next_entry_if_used();
   PreconSetupDispatch((HYPRE_Solver)(void *)&h->cookie, h->mat_M ? h->mat_M : h->mat_A, h->vec_b, h->vec_x);
   consume_hypre_errors();
   API_CATCH
}

// reference src/HYPREDRV.c:3001-3119 -> hypredrv_SolverSetupWithReuse (src/internal/solver.c:457-546)
// ------------------------------------------------------------------ scaling
// solver.scaling (reference src/internal/scaling.c): the system is transformed in place before the preconditioner is
// built, solved in the scaled variables and transformed back afterwards.  With D = diag(custom_values[dofmap]):
//   rhs_l2                     s = 1/sqrt(||b||_2):  A <- s^2 A,      b <- s b,      x <- x / s      (scaling.c:246-262, :1050-1059)
//   dofmap_custom              A <- D A D,           b <- D b,        x <- D^-1 x                    (:900-928 default branch)
//   dofmap_row_custom          A <- D A,             b <- D b,        x unchanged
//   dofmap_col_custom          A <- A D,             b unchanged,     x <- D^-1 x
//   dofmap_similarity_custom   A <- D^-1 A D,        b <- D^-1 b,     x <- D^-1 x                    (:838-848)
// The inverse transforms multiply by the reciprocals, as the reference does, so the caller's matrix comes back up to
// rounding, not bit for bit.
namespace {
__global__ void k_scl_vals(int n, double s, double *v)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) v[i] *= s;
}
// a_ij <- left_i a_ij right_j (either side may be absent); one 64-lane wave per row
__global__ void k_scl_diag(int nrows, const int *rowptr, const int *col, double *val, const double *left, const double *right)
{
   const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
   if (row >= nrows) return;
   const double l = left ? left[row] : 1.0;
   for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += 64) val[k] = l * val[k] * (right ? right[col[k]] : 1.0);
}
__global__ void k_scl_mul(int n, const double *d, double *v)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) v[i] = d[i] * v[i];
}
__global__ void k_scl_div(int n, const double *d, double *v)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) v[i] = v[i] / d[i];
}
__global__ void k_scl_inv(int n, const double *d, double *out)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) out[i] = 1.0 / d[i];
}
enum { SCL_RHS = 0, SCL_UNKNOWN = 1 };

// hypredrv_ScalingCompute (scaling.c:577-660).  Collective; every rank leaves with the same verdict.
void scaling_compute(hypredrv_struct *h)
{
   auto              &c = h->scal;
   const ScalingArgs &a = h->args.solver.scaling;
   c.enabled = a.enabled != 0;
   c.type    = a.type;
   if (!c.enabled) return;
   if (a.type == 0)
   { // ScalingComputeRHSL2 (:246-262)
      double bb = 0.0;
      HYPRE_ParVectorInnerProd(h->vec_b, h->vec_b, &bb);
      const double bn = std::sqrt(bb);
      c.scalar_factor = bn > 0.0 ? 1.0 / std::sqrt(bn) : 1.0;
      return;
   }
   // ScalingComputeDofmapCustom (:388-560)
   const int           n = h->mat_A->nloc;
   uint32_t            bad = 0;
   std::string         why;
   std::vector<double> w((size_t)n, 1.0);
   long long           max_tag = -1;
   if (h->dofmap.empty() && n > 0) { bad = ERR_MISSING_DOFMAP; why = "custom dofmap scaling requires a dofmap to be set"; }
   else if (a.custom_values.empty()) { bad = ERR_UNKNOWN; why = "custom dofmap scaling requires custom_values to be set"; }
   else if ((int)h->dofmap.size() != n) { bad = ERR_UNKNOWN; why = "dofmap size (" + std::to_string(h->dofmap.size()) + ") does not match local matrix rows (" + std::to_string(n) + ")"; }
   else
   {
      for (size_t i = 0; i < a.custom_values.size(); i++)
         if (a.custom_values[i] == 0.0) { bad = ERR_INVALID_VAL; why = "custom dofmap scaling requires nonzero custom_values (entry " + std::to_string(i) + " is zero)"; }
      for (size_t i = 0; i < h->dofmap.size(); i++)
      {
         const int t = h->dofmap[i];
         // a negative tag is a local finding, but the verdict must be collective: it rides in the same all-reduce
         if (t < 0 && !bad) { bad = ERR_UNKNOWN; why = "dofmap_custom: invalid tag " + std::to_string(t) + " at local row " + std::to_string(i); }
         max_tag = std::max<long long>(max_tag, t);
      }
   }
   long long red[2] = {max_tag, (long long)bad};
   if (Comm::world().size > 1) Comm::world().allreduce_host(red, 2, 1);
   if (!red[1] && red[0] + 1 != (long long)a.custom_values.size())
   {
      bad = ERR_UNKNOWN;
      why = "dofmap_custom: number of custom_values (" + std::to_string(a.custom_values.size()) + ") does not match number of unique dofmap tags (" + std::to_string(red[0] + 1) + ")";
      red[1] = bad;
   }
   if (!red[1])
      for (int i = 0; i < n; i++)
      {
         w[(size_t)i] = a.custom_values[(size_t)h->dofmap[(size_t)i]]; // tags are in [0, custom_values.size()) on every rank here
      }
   if (bad || red[1]) { err_set(bad ? bad : (uint32_t)red[1], why.empty() ? std::string("custom dofmap scaling was rejected on another rank") : why); return; }
   c.scaling.alloc((size_t)std::max(n, 1));
   c.inverse_scaling.alloc((size_t)std::max(n, 1));
   c.scaling.upload(w.data(), (size_t)n);
   if (n) k_scl_inv<<<ceil_div(n, 256), 256, 0, Context::get().stream>>>(n, c.scaling.data(), c.inverse_scaling.data());
}

// ScalingTransformVector (scaling.c:788-868)
bool scaling_vector(hypredrv_struct *h, HYPRE_IJVector v, int kind, bool apply)
{
   auto &c = h->scal;
   if (!c.enabled || !v) return false;
   v->ensure_device();
   const int n = v->nloc;
   if (c.type == 0)
   {
      const double s = c.scalar_factor;
      scale(n, kind == SCL_RHS ? (apply ? s : 1.0 / s) : (apply ? 1.0 / s : s), v->data());
      return true;
   }
   if ((c.type == 3 && kind != SCL_RHS) || (c.type == 4 && kind != SCL_UNKNOWN)) return false;
   if (c.type == 5 && kind == SCL_RHS) apply = !apply; // c = S^-1 b
   const bool product = (kind == SCL_RHS && apply) || (kind == SCL_UNKNOWN && !apply);
   if (n)
   {
      if (product) k_scl_mul<<<ceil_div(n, 256), 256, 0, Context::get().stream>>>(n, c.scaling.data(), v->data());
      else k_scl_div<<<ceil_div(n, 256), 256, 0, Context::get().stream>>>(n, c.scaling.data(), v->data());
   }
   return true;
}

void scaling_matrix(hypredrv_struct *h, HYPRE_IJMatrix M, bool apply)
{
   auto &c = h->scal;
   DCsr &A = M->A;
   if (c.type == 0)
   {
      const double s2 = c.scalar_factor * c.scalar_factor;
      if (A.nnz) k_scl_vals<<<ceil_div(A.nnz, 256), 256, 0, Context::get().stream>>>(A.nnz, apply ? s2 : 1.0 / s2, A.val.data());
   }
   else
   { // ScalingDofmapMatrixFactors (:900-928)
      const double *sc = c.scaling.data(), *inv = c.inverse_scaling.data(), *left = nullptr, *right = nullptr;
      switch (c.type)
      {
         case 3: left = apply ? sc : inv; break;
         case 4: right = apply ? sc : inv; break;
         case 5: left = apply ? inv : sc; right = apply ? sc : inv; break;
         default: left = right = apply ? sc : inv; break;
      }
      DArray<double> ext; // the column factor over [owned | ghost] columns
      if (right)
      {
         ext.alloc((size_t)std::max(A.ncols, 1));
         copy(M->nloc, right, ext.data());
         if (A.ncols > M->nloc) halo_exchange(M->halo, ext.data());
         right = ext.data();
      }
      if (A.nrows) k_scl_diag<<<ceil_div((long long)A.nrows * 64, 256), 256, 0, Context::get().stream>>>(A.nrows, A.rowptr.data(), A.col.data(), A.val.data(), left, right);
      Context::get().sync(); // ext is released on return
   }
   A.reset_plan(); // the streaming plan's stencil-coded shadow holds values
}

// hypredrv_ScalingApplyToSystem / UndoOnSystem (scaling.c:950-1205)
void scaling_system(hypredrv_struct *h, bool apply)
{
   auto &c = h->scal;
   if (!c.enabled || (!apply && !c.is_applied()) || (apply && c.is_applied())) return;
   HYPRE_IJMatrix M = h->mat_M ? h->mat_M : h->mat_A;
   if (apply)
   {
      scaling_matrix(h, h->mat_A, true);
      if (M != h->mat_A) scaling_matrix(h, M, true);
      c.matrices_are_scaled = true;
      if (scaling_vector(h, h->vec_b, SCL_RHS, true)) c.rhs_is_scaled = true;
      if (scaling_vector(h, h->vec_x, SCL_UNKNOWN, true)) c.x_is_scaled = true;
   }
   else
   {
      if (c.x_is_scaled && scaling_vector(h, h->vec_x, SCL_UNKNOWN, false)) c.x_is_scaled = false;
      if (c.rhs_is_scaled && scaling_vector(h, h->vec_b, SCL_RHS, false)) c.rhs_is_scaled = false;
      if (c.matrices_are_scaled)
      {
         scaling_matrix(h, h->mat_A, false);
         if (M != h->mat_A) scaling_matrix(h, M, false);
         c.matrices_are_scaled = false;
      }
   }
}
// RestoreScaledSystemState (src/HYPREDRV.c:141-158): keeps an error that is already raised
void scaling_restore(hypredrv_struct *h, bool xref_scaled)
{
   try
   {
      scaling_system(h, false);
      if (xref_scaled && h->vec_xref) scaling_vector(h, h->vec_xref, SCL_UNKNOWN, false);
   }
   catch (const std::exception &e) { err_set(ERR_HYPRE_INTERNAL, e.what()); }
}
} // namespace

extern "C" uint32_t HYPREDRV_LinearSolverSetup(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   set_stage("HYPREDRV_LinearSolverSetup");
   if (!h->solver) return err_set(ERR_INVALID_SOLVER, "LinearSolverSetup: solver is NULL (call LinearSolverCreate)");
   if (!h->mat_A || !h->vec_b || !h->vec_x) return err_set(ERR_UNKNOWN, "LinearSolverSetup: matrix, rhs or solution vector is missing");
   h->stats.next_entry_if_used();
   if (h->stats.pending_build > 0.0)
   {
      h->stats.cur().build += h->stats.pending_build;
      h->stats.pending_build = 0.0;
   }
   HYPRE_Matrix M = h->mat_M ? h->mat_M : h->mat_A;
   const auto t0  = clk::now();
   const SolverOps &ops = solver_ops(h->solver);
   if (h->precon) ops.set_precond(h->solver, PreconSolveDispatch, PreconSetupDispatch, (HYPRE_Solver)(void *)&h->cookie);
   // reuse decision (src/HYPREDRV.c:3010-3020): a preconditioner that is set up is kept unless the policy asks
   // for a rebuild on this system; the Krylov setup has no other work, so it is skipped with it
   const bool skip_precon_setup = h->precon && h->precon_is_setup && !reuse_should_rebuild(h);
   // src/HYPREDRV.c:3039-3072: the scaling is computed for every system and applied before the setup; the system
   // stays scaled until the end of LinearSolverApply
   if (h->args.solver.scaling.enabled)
   {
      scaling_compute(h);
      if (!g_err) scaling_system(h, true);
      if (g_err) { scaling_restore(h, false); dist_error_sync(); return g_err; }
   }
   if (!skip_precon_setup) ops.setup(h->solver, M, h->vec_b, h->vec_x);
   h->last_setup_s = std::chrono::duration<double>(clk::now() - t0).count();
   consume_hypre_errors();
   if (g_err && h->scal.is_applied()) scaling_restore(h, false);
   API_CATCH_SYNC
}

// ||b - A x||_2 (reference src/internal/linsys.c:2982-3067): the residual kernel on a work vector from the library's pool,
// the ghost refresh of x under it on a row block
static double residual_norm(hypredrv_struct *h)
{
   hypre_IJMatrix_struct *A = h->mat_A;
   const int              n = A->nloc;
   h->vec_x->ensure_device();
   h->vec_b->ensure_device();
   DArray<double> r((size_t)std::max(n, 1)), xe;
   double        *xin = h->vec_x->data();
   if (A->A.ncols > n || h->vec_x->capacity < (size_t)A->A.ncols)
   { // x has no room for the ghost tail: stage it
      xe.alloc((size_t)std::max(A->A.ncols, 1));
      copy(n, h->vec_x->data(), xe.data());
      xin = xe.data();
   }
   residual(A->A, xin, h->vec_b->data(), r.data(), Comm::world().size > 1 ? &A->halo : nullptr);
   dot(n, r.data(), r.data(), 0);
   finalize(0, S_TMP);
   return std::sqrt(read_scalar(S_TMP));
}

// reference src/HYPREDRV.c:3126-3338 -> hypredrv_SolverApply (src/internal/solver.c:627-693)
extern "C" uint32_t HYPREDRV_LinearSolverApply(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   set_stage("HYPREDRV_LinearSolverApply");
   if (!h->solver) return err_set(ERR_INVALID_SOLVER, "LinearSolverApply: solver is NULL");
   if (h->args.precon().method != 99 && (!h->precon || !h->precon_is_setup))
      return err_set(ERR_INVALID_PRECON, "Linear solver apply requires a successfully set up preconditioner; check the preceding setup error");
   if (!h->mat_A || !h->vec_b || !h->vec_x) return err_set(ERR_UNKNOWN, "SolverApply: matrix or vector is NULL");
   // src/HYPREDRV.c:3161-3200: a reused preconditioner skips Setup's transform, so it is applied here; r0 is
   // measured on the system as the solver sees it
   const bool scaled = h->args.solver.scaling.enabled && h->scal.enabled;
   bool       xref_scaled = false;
   if (scaled && !h->scal.is_applied()) scaling_system(h, true);
   const double r0 = residual_norm(h); // untimed (solver.c:666)
   if (scaled && h->vec_xref) xref_scaled = scaling_vector(h, h->vec_xref, SCL_UNKNOWN, true);
   annotate(h, "solve", true);
   h->stats.cur().r0 = r0;
   const SolverOps &ops  = solver_ops(h->solver);
   const HYPRE_Int  ierr = ops.solve(h->solver, h->mat_A, h->vec_b, h->vec_x);
   if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
   HYPRE_Int  iters = 0, conv = 0;
   HYPRE_Real frel  = 0.0;
   ops.get_num_iterations(h->solver, &iters);
   ops.get_converged(h->solver, &conv);
   ops.get_final_rel_res_norm(h->solver, &frel);
   h->stats.cur().iters     = ierr ? 0 : iters;
   h->stats.cur().has_solve = true;
   annotate(h, "solve", false);
   h->last_solve_s   = h->stats.cur().solve;
   h->last_iters     = iters;
   h->last_converged = conv;
   h->last_rel       = frel;
   if (scaled) scaling_restore(h, xref_scaled); // src/HYPREDRV.c:3246-3259: norms below are those of the caller's system
   if (!ierr)
   { // true relative residual, untimed (solver.c:686-690)
      double bn = 0.0;
      HYPRE_ParVectorInnerProd(h->vec_b, h->vec_b, &bn);
      bn = std::sqrt(bn);
      h->stats.cur().rr = residual_norm(h) / (bn > 0.0 ? bn : 1.0);
   }
   if (!ierr) project_out_null_space(h); // src/HYPREDRV.c:3296-3300: fix the gauge of the solution (after the solver's own residual report)
   if (h->vec_xref && !ierr)
   { // src/HYPREDRV.c:3310-3323: error against the reference solution, when one was set
      double xx = 0.0, rr = 0.0, ee = 0.0;
      h->vec_xref->ensure_device();
      HYPRE_ParVectorInnerProd(h->vec_x, h->vec_x, &xx);
      HYPRE_ParVectorInnerProd(h->vec_xref, h->vec_xref, &rr);
      HYPRE_IJVector e = new_vector_like(h, 0.0);
      HYPRE_ParVectorCopy(h->vec_xref, e);
      axpy(e->nloc, -1.0, h->vec_x->data(), e->data());
      HYPRE_ParVectorInnerProd(e, e, &ee);
      HYPRE_IJVectorDestroy(e);
      if (!h->mypid)
      {
         printf("L2 norm of error: %e\n", std::sqrt(ee));
         printf("L2 norm of solution: %e\n", std::sqrt(xx));
         printf("L2 norm of ref. solution: %e\n", std::sqrt(rr));
      }
   }
   consume_hypre_errors(); // non-convergence is not an error
   API_CATCH_SYNC
}

extern "C" uint32_t HYPREDRV_PreconApply(HYPREDRV_t h, HYPRE_Vector b, HYPRE_Vector x)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   if (!h->precon || !h->precon_is_setup) return err_set(ERR_INVALID_PRECON, "PreconApply requires a set-up preconditioner");
   if (h->precon->kind == HDA_SOLVER_ILU) HYPRE_ILUSolve(h->precon, h->mat_M ? h->mat_M : h->mat_A, b, x);
   else if (h->precon->kind == HDA_SOLVER_MGR) HYPRE_MGRSolve(h->precon, h->mat_M ? h->mat_M : h->mat_A, b, x);
   else HYPRE_BoomerAMGSolve(h->precon, h->mat_M ? h->mat_M : h->mat_A, b, x);
   consume_hypre_errors();
   API_CATCH
}
extern "C" uint32_t HYPREDRV_PreconDestroy(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   API_TRY
   // src/HYPREDRV.c:3403-3452: destroyed only when the next system will rebuild it, otherwise kept for reuse
   if (h->precon && reuse_should_rebuild(h))
   {
      HYPRE_BoomerAMGDestroy(h->precon);
      for (HYPRE_Solver a : h->precon_aux) HYPRE_BoomerAMGDestroy(a);
      h->precon_aux.clear();
      h->precon          = nullptr;
      h->precon_is_setup = false;
   }
   API_CATCH
}
extern "C" uint32_t HYPREDRV_LinearSolverDestroy(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   err_reset();
   // the preconditioner goes first, through the policy-aware PreconDestroy (src/HYPREDRV.c:3463-3496)
   if (h->precon && HYPREDRV_PreconDestroy(h)) return g_err;
   if (h->solver) solver_ops(h->solver).destroy(h->solver);
   h->solver = nullptr;
   return g_err;
}

// ------------------------------------------------------------ stats / getters

extern "C" uint32_t HYPREDRV_AnnotateBegin(HYPREDRV_t h, const char *name, int) { CHECK_INIT_OBJ(h); return annotate(h, name, true); }
extern "C" uint32_t HYPREDRV_AnnotateEnd(HYPREDRV_t h, const char *name, int) { CHECK_INIT_OBJ(h); return annotate(h, name, false); }

// table layout of the reference: src/internal/stats.c:533-536,648-657,1234-1361
extern "C" uint32_t HYPREDRV_StatsPrint(HYPREDRV_t h)
{
   CHECK_INIT_OBJ(h);
   if (!h->args.general.statistics) return g_err;
   const Stats &s     = h->stats;
   const double tf    = s.use_millisec ? 1000.0 : 1.0;
   const char  *scale = s.use_millisec ? "[ms]" : "[s]";
   const int    w[7]  = {10, 11, 11, 11, 10, 10, 6};
   auto divisor = [&]() {
      for (int i = 0; i < 7; i++)
      {
         putchar('+');
         for (int j = 0; j < w[i] + 2; j++) putchar('-');
      }
      printf("+\n");
   };
   printf("====================================================================================\n");
   if (!h->name.empty()) printf("\n\nSTATISTICS SUMMARY for %s:\n\n", h->name.c_str());
   else printf("\n\nSTATISTICS SUMMARY:\n\n");
   divisor();
   printf("| %*s | %*s | %*s | %*s | %*s | %*s | %*s |\n", w[0], "", w[1], "LS build", w[2], "setup", w[3], "solve", w[4], "initial",
          w[5], "relative", w[6], "");
   char t1[32];
   snprintf(t1, sizeof(t1), "times %s", scale);
   bool use_path = false;
   for (const StatEntry &e : s.entries) use_path |= (e.has_solve && !e.path.empty());
   printf("| %*s | %*s | %*s | %*s | %*s | %*s | %*s |\n", w[0], use_path ? "Path" : "Entry", w[1], t1, w[2], t1, w[3], t1, w[4], "res. norm", w[5],
          "res. norm", w[6], "iters");
   divisor();
   int idx = 0;
   for (const StatEntry &e : s.entries)
   {
      if (!e.has_solve) continue;
      char label[32];
      if (use_path && !e.path.empty())
      { // long paths keep their tail (reference stats.c:612-622)
         if (e.path.size() <= 10) snprintf(label, sizeof(label), "%s", e.path.c_str());
         else snprintf(label, sizeof(label), "...%s", e.path.c_str() + (e.path.size() - 7));
         idx++;
      }
      else snprintf(label, sizeof(label), "%d", idx++);
      if (e.build > 0.0)
         printf("| %*s | %*.*f | %*.*f | %*.*f | %*.*e | %*.*e | %*d |\n", w[0], label, w[1], 3, tf * e.build, w[2], 3, tf * e.prec, w[3], 3,
                tf * e.solve, w[4], 2, e.r0, w[5], 2, e.rr, w[6], e.iters);
      else
         printf("| %*s | %*s | %*.*f | %*.*f | %*.*e | %*.*e | %*d |\n", w[0], label, w[1], "", w[2], 3, tf * e.prec, w[3], 3, tf * e.solve,
                w[4], 2, e.r0, w[5], 2, e.rr, w[6], e.iters);
   }
   // general.statistics >= 2: aggregate rows inside the same table (reference src/internal/stats.c:1262-1358)
   if (idx > 1 && h->args.general.statistics > 1)
   {
      struct Agg {
         double mn = HUGE_VAL, mx = 0.0, sum = 0.0, ssq = 0.0;
         void   add(double v) { mn = std::min(mn, v); mx = std::max(mx, v); sum += v; ssq += v * v; }
         double avg(int n) const { return sum / n; }
         double sd(int n) const { const double a = avg(n), var = ssq / n - a * a; return std::sqrt(var > 0.0 ? var : 0.0); } // clamped as StatsVarianceClamp does
      } b, st, sv, r0, rr, it;
      int imin = INT_MAX, imax = 0, isum = 0;
      for (const StatEntry &e : s.entries)
      {
         if (!e.has_solve) continue;
         b.add(tf * e.build); st.add(tf * e.prec); sv.add(tf * e.solve); r0.add(e.r0); rr.add(e.rr); it.add((double)e.iters);
         imin = std::min(imin, e.iters); imax = std::max(imax, e.iters); isum += e.iters;
      }
      const int n = idx;
      divisor();
      printf("| %*s | %*.*f | %*.*f | %*.*f | %*.*e | %*.*e | %*d |\n", w[0], "Min.", w[1], 3, b.mn, w[2], 3, st.mn, w[3], 3, sv.mn, w[4], 2, r0.mn, w[5], 2, rr.mn,
             w[6], imin);
      printf("| %*s | %*.*f | %*.*f | %*.*f | %*.*e | %*.*e | %*d |\n", w[0], "Max.", w[1], 3, b.mx, w[2], 3, st.mx, w[3], 3, sv.mx, w[4], 2, r0.mx, w[5], 2, rr.mx,
             w[6], imax);
      printf("| %*s | %*.*f | %*.*f | %*.*f | %*.*e | %*.*e | %*.1f |\n", w[0], "Avg.", w[1], 3, b.avg(n), w[2], 3, st.avg(n), w[3], 3, sv.avg(n), w[4], 2,
             r0.avg(n), w[5], 2, rr.avg(n), w[6], it.avg(n));
      printf("| %*s | %*.*f | %*.*f | %*.*f | %*.*e | %*.*e | %*.1f |\n", w[0], "Std.", w[1], 3, b.sd(n), w[2], 3, st.sd(n), w[3], 3, sv.sd(n), w[4], 2,
             r0.sd(n), w[5], 2, rr.sd(n), w[6], it.sd(n));
      printf("| %*s | %*.*f | %*.*f | %*.*f | %*s | %*s | %*d |\n", w[0], "Total", w[1], 3, b.sum, w[2], 3, st.sum, w[3], 3, sv.sum, w[4], "", w[5], "", w[6], isum);
   }
   divisor();
   printf("\n");
   return g_err;
}

extern "C" uint32_t HYPREDRV_LinearSolverGetNumIter(HYPREDRV_t h, int *iters) { CHECK_INIT_OBJ(h); if (iters) *iters = h->last_iters; return g_err; }
extern "C" uint32_t HYPREDRV_LinearSolverGetConverged(HYPREDRV_t h, int *c) { CHECK_INIT_OBJ(h); if (c) *c = h->last_converged; return g_err; }
extern "C" uint32_t HYPREDRV_LinearSolverGetFinalRelativeResidualNorm(HYPREDRV_t h, double *n) { CHECK_INIT_OBJ(h); if (n) *n = h->last_rel; return g_err; }
extern "C" uint32_t HYPREDRV_LinearSolverGetSetupTime(HYPREDRV_t h, double *s) { CHECK_INIT_OBJ(h); if (s) *s = h->stats.cur().prec; return g_err; }
extern "C" uint32_t HYPREDRV_LinearSolverGetSolveTime(HYPREDRV_t h, double *s) { CHECK_INIT_OBJ(h); if (s) *s = h->last_solve_s; return g_err; }
