/*
 * hypredrive-cli (MI355X build): runs hypredrive YAML inputs (examples/ex1.yml ...) through the
 * HYPREDRV_* API exactly in the order of the reference driver's solve loop
 * (reference src/internal/main.c:176-229: Build, PreconCreate, SolverCreate, Setup, Apply,
 * Destroy per linear system x preconditioner variant x repetition), then prints the
 * statistics table.  Usage: hypredrive-cli [-q] <input.yml> [-a --path:to:key value ...]
 *
 * Ranks: one process per GPU.  When launched with RANK / WORLD_SIZE / LOCAL_RANK in the
 * environment (torchrun style) rank 0 publishes the RCCL unique id through a file next to
 * MASTER_PORT and every rank joins the communicator before HYPREDRV_Initialize.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "HYPREDRV.h"
#include "HYPREDRV_utils.h"

static int env_int(const char *name, int dflt)
{
   const char *v = getenv(name);
   return v ? atoi(v) : dflt;
}

static void join_world(void)
{
   int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
   if (world <= 1) return;
   char path[256], tmp[300];
   /* one file per LAUNCH: the ranks of a launch are children of one launcher process (torchrun's agent, mpirun's daemon, a shell), so its
    * pid tells this launch's id file from one a crashed earlier launch on the same port left behind (reading that would hang every rank
    * inside ncclCommInitRank) */
   snprintf(path, sizeof(path), "/tmp/hypredrv_amd_uid_%s_%ld", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0", (long)getppid());
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
   join_world();
   int myid = env_int("RANK", 0);
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
   return 0;
}
