/*
 * hypredrive-cli (MI355X build): runs hypredrive YAML inputs (examples/ex1.yml ...) through the
 * HYPREDRV_* API exactly in the order of the reference driver's solve loop
 * (reference src/internal/main.c:176-229: Build, PreconCreate, SolverCreate, Setup, Apply,
 * Destroy per linear system x preconditioner variant x repetition), then prints the
 * statistics table.  Usage: hypredrive-cli [-q] <input.yml> [-a --path:to:key value ...]
 *
 * Ranks: one process per GPU.
 *  - `mpiexec -n N hypredrive-cli input.yml`, the reference's launch line (cmake/HYPREDRV_Testing.cmake:938 "ex2_4proc"): the binary has
 *    no link-time MPI dependency; when a process manager's variables are in the environment it loads the MPI library at run time
 *    (HDA_MPI_LIB, else libmpi.so.12 / libmpi.so), calls MPI_Init and hands MPI_COMM_WORLD to HYPREDRV_Create, where the library
 *    joins the ranks (RCCL with one GPU per rank, MPI-staged when ranks share one).  MPICH-ABI MPIs; for another MPI compile this
 *    file with its mpicc and -DHYPREDRV_AMD_USE_MPI together with hypredrive_amd/csrc/hda_mpi_shim.c.
 *  - RANK / WORLD_SIZE / LOCAL_RANK in the environment (torchrun style): rank 0 publishes the RCCL unique id through a file keyed on
 *    MASTER_PORT and the launch's id, and every rank joins the communicator before HYPREDRV_Initialize.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <dlfcn.h>

#include "HYPREDRV.h"
#include "HYPREDRV_utils.h"

static int env_int(const char *name, int dflt)
{
   const char *v = getenv(name);
   return v ? atoi(v) : dflt;
}

/* ---- started by an MPI process manager: become an MPI program at run time */
static void *mpi_lib = NULL;
static int   mpi_rank = -1;
typedef int (*mpi_fn_t)();
static mpi_fn_t mpi_sym(const char *name)
{
   mpi_fn_t f = (mpi_fn_t)dlsym(mpi_lib, name);
   if (!f) { fprintf(stderr, "hypredrive-cli: %s not found in the MPI library\n", name); exit(1); }
   return f;
}
static int start_mpi(int *argc, char ***argv)
{
#ifdef HYPREDRV_AMD_USE_MPI
   MPI_Init(argc, argv);
   MPI_Comm_rank(MPI_COMM_WORLD, &mpi_rank);
   return 1;
#else
   const char *managed[] = {"PMI_RANK", "PMI_SIZE", "PMIX_RANK", "OMPI_COMM_WORLD_RANK", "MPI_LOCALRANKID", "HYDI_CONTROL_FD"};
   int         found = 0;
   for (size_t i = 0; i < sizeof(managed) / sizeof(managed[0]); i++) found |= getenv(managed[i]) != NULL;
   if (!found || getenv("WORLD_SIZE")) return 0; /* (torchrun's variables win: that launcher's ranks join through the id file) */
   const char *names[] = {getenv("HDA_MPI_LIB"), "libmpi.so.12", "libmpi.so", "/opt/conda/lib/libmpi.so.12"};
   for (size_t i = 0; i < sizeof(names) / sizeof(names[0]) && !mpi_lib; i++)
      if (names[i]) mpi_lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
   if (!mpi_lib)
   {
      fprintf(stderr, "hypredrive-cli: started by an MPI process manager but no MPI library could be loaded (%s); set HDA_MPI_LIB\n", dlerror());
      exit(1);
   }
   char ver[8192 + 64] = {0};
   int  len = 0;
   ((int (*)(char *, int *))mpi_sym("MPI_Get_library_version"))(ver, &len);
   if (!strstr(ver, "MPICH") && !strstr(ver, "Intel(R) MPI") && !strstr(ver, "MVAPICH"))
   {
      fprintf(stderr, "hypredrive-cli: %.60s is not an MPICH-ABI MPI: compile cli/hypredrive_cli.c with its mpicc and -DHYPREDRV_AMD_USE_MPI\n", ver);
      exit(1);
   }
   ((int (*)(int *, char ***))mpi_sym("MPI_Init"))(argc, argv);
   ((int (*)(MPI_Comm, int *))mpi_sym("MPI_Comm_rank"))(MPI_COMM_WORLD, &mpi_rank);
   return 1;
#endif
}
static void stop_mpi(void)
{
#ifdef HYPREDRV_AMD_USE_MPI
   MPI_Finalize();
#else
   if (mpi_lib) ((int (*)(void))mpi_sym("MPI_Finalize"))();
#endif
}

static void join_world(void)
{
   int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
   if (world <= 1) return;
   char path[256], tmp[300], key[96];
   /* one file per LAUNCH, so that an id file a crashed earlier launch on the same port left behind is never read (that would hang every
    * rank inside ncclCommInitRank): keyed on the launcher's own id where it publishes one -- ranks started through per-rank wrapper
    * scripts have different parents --, else on the parent's pid (the ranks of a plain launch are children of one launcher process) */
   const char *ids[] = {"TORCHELASTIC_RUN_ID", "SLURM_STEP_ID", "SLURM_JOB_ID", "OMPI_MCA_ess_base_jobid", "PMI_JOBID", "HDA_LAUNCH_ID"};
   key[0] = 0;
   for (size_t i = 0; i < sizeof(ids) / sizeof(ids[0]) && !key[0]; i++)
      if (getenv(ids[i]) && strcmp(getenv(ids[i]), "none")) snprintf(key, sizeof(key), "%.80s", getenv(ids[i]));
   if (!key[0]) snprintf(key, sizeof(key), "ppid%ld", (long)getppid());
   for (char *c = key; *c; c++)
      if (*c == '/' || *c == ' ') *c = '_';
   snprintf(path, sizeof(path), "/tmp/hypredrv_amd_uid_%s_%s", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0", key);
   unsigned char uid[128];
   if (rank == 0)
   {
      HYPREDRV_SAFE_CALL(HYPREDRV_AMD_CommGetUniqueId(uid));
      snprintf(tmp, sizeof(tmp), "%s.tmp", path);
      FILE *f = fopen(tmp, "wb");
      if (!f || fwrite(uid, 1, 128, f) != 128) { fprintf(stderr, "cannot publish the RCCL id\n"); exit(1); }
      fclose(f);
      rename(tmp, path);
   }
   else
   {
      FILE *f = NULL;
      for (int tries = 0; tries < 600 && !(f = fopen(path, "rb")); tries++) usleep(100000);
      if (!f || fread(uid, 1, 128, f) != 128) { fprintf(stderr, "rank %d: no RCCL id at %s\n", rank, path); exit(1); }
      fclose(f);
   }
   HYPREDRV_SAFE_CALL(HYPREDRV_AMD_CommInit(rank, world, local, uid));
   if (rank == 0) { sleep(1); unlink(path); }
}

static void run_solve_loops(HYPREDRV_t obj)
{
   int nls = 0, nvar = 0;
   HYPREDRV_SAFE_CALL(HYPREDRV_InputArgsGetNumLinearSystems(obj, &nls));
   HYPREDRV_SAFE_CALL(HYPREDRV_InputArgsGetNumPreconVariants(obj, &nvar));
   for (int k = 0; k < nls; k++)
   {
      HYPREDRV_SAFE_CALL(HYPREDRV_LinearSystemBuild(obj));
      for (int v = 0; v < nvar; v++)
      {
         int reps = 0;
         HYPREDRV_SAFE_CALL(HYPREDRV_InputArgsSetPreconVariant(obj, v));
         HYPREDRV_SAFE_CALL(HYPREDRV_InputArgsGetNumRepetitions(obj, &reps));
         for (int i = 0; i < reps; i++)
         {
            HYPREDRV_SAFE_CALL(HYPREDRV_AnnotateBegin(obj, "Run", i));
            HYPREDRV_SAFE_CALL(HYPREDRV_LinearSystemResetInitialGuess(obj));
            HYPREDRV_SAFE_CALL(HYPREDRV_PreconCreate(obj));
            HYPREDRV_SAFE_CALL(HYPREDRV_LinearSolverCreate(obj));
            HYPREDRV_SAFE_CALL(HYPREDRV_LinearSolverSetup(obj));
            HYPREDRV_SAFE_CALL(HYPREDRV_LinearSolverApply(obj));
            HYPREDRV_SAFE_CALL(HYPREDRV_PreconDestroy(obj));
            HYPREDRV_SAFE_CALL(HYPREDRV_LinearSolverDestroy(obj));
            HYPREDRV_SAFE_CALL(HYPREDRV_AnnotateEnd(obj, "Run", i));
         }
      }
   }
}

int main(int argc, char **argv)
{
   MPI_Comm comm = MPI_COMM_WORLD;
   int      quiet = 0, first = 1;
   if (argc > 1 && !strcmp(argv[1], "-q")) { quiet = 1; first = 2; }
   if (argc <= first)
   {
      fprintf(stderr, "usage: %s [-q] <input.yml> [-a --path:to:key value ...]\n", argv[0]);
      return 1;
   }
   const int under_mpi = start_mpi(&argc, &argv);
   if (!under_mpi) join_world();
   int myid = under_mpi ? mpi_rank : env_int("RANK", 0);
   HYPREDRV_SAFE_CALL(HYPREDRV_Initialize());
   HYPREDRV_t obj = NULL;
   HYPREDRV_SAFE_CALL(HYPREDRV_Create(comm, &obj));
   if (!quiet)
   {
      HYPREDRV_SAFE_CALL(HYPREDRV_PrintLibInfo(comm, 1));
      HYPREDRV_SAFE_CALL(HYPREDRV_PrintSystemInfo(comm));
   }
   HYPREDRV_SAFE_CALL(HYPREDRV_InputArgsParse(argc - first, argv + first, obj));
   for (int i = first; i + 1 < argc; i++) /* -p <preset>: preconditioner preset (main.c FindPreconPreset) */
      if (!strcmp(argv[i], "-p") || !strcmp(argv[i], "--preset")) HYPREDRV_SAFE_CALL(HYPREDRV_InputArgsSetPreconPreset(obj, argv[i + 1]));
   run_solve_loops(obj);
   if (!myid) HYPREDRV_SAFE_CALL(HYPREDRV_StatsPrint(obj));
   HYPREDRV_SAFE_CALL(HYPREDRV_Destroy(&obj));
   if (!quiet) HYPREDRV_SAFE_CALL(HYPREDRV_PrintExitInfo(comm, argv[0]));
   HYPREDRV_SAFE_CALL(HYPREDRV_AMD_CommFinalize());
   HYPREDRV_SAFE_CALL(HYPREDRV_Finalize());
   if (under_mpi) stop_mpi();
   return 0;
}
