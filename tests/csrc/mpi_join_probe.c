/* mpi_join_probe.c -- an MPI program that hands MPI_COMM_WORLD to HYPREDRV_Create and does nothing else to connect its ranks
 * (what the reference's multi-rank callers do: tests/test_setmatrix_from_csr_mpi.c:145-190).  Checks that the library joined them
 * (tests/test_mpi_join.py).  Modes: "host" = host collectives only (runs without a GPU), "device" = + the transport self-test,
 * "self" = every rank creates its object on MPI_COMM_SELF (N independent one-rank solves stay unjoined), "hang" = rank 1 never
 * enters the collective (the HDA_COMM_TIMEOUT_S watchdog must end the job), "shim" = join through hda_mpi_shim.c instead. */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "HYPREDRV.h"
#include "hypredrv_amd.h"
#ifdef PROBE_WITH_SHIM
uint32_t HYPREDRV_AMD_CommInitMPI(MPI_Comm comm);
#endif

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "rank %d: check failed: %s (line %d): %s\n", rank, #c, __LINE__, HYPREDRV_AMD_LastErrorMessage()); fflush(NULL); usleep(300000); MPI_Abort(MPI_COMM_WORLD, 3); } } while (0)

int main(int argc, char **argv)
{
   int rank = 0, size = 1;
   MPI_Init(&argc, &argv);
   MPI_Comm_rank(MPI_COMM_WORLD, &rank);
   MPI_Comm_size(MPI_COMM_WORLD, &size);
   const char *mode = argc > 1 ? argv[1] : "host";
   HYPREDRV_t obj = NULL;
   CHECK(HYPREDRV_Initialize() == 0);
   if (!strcmp(mode, "self"))
   {
      CHECK(HYPREDRV_Create(MPI_COMM_SELF, &obj) == 0);
      CHECK(hda_comm_size() == 1 && !strcmp(hda_comm_name(), "self"));
      CHECK(HYPREDRV_Destroy(&obj) == 0);
      CHECK(HYPREDRV_Finalize() == 0);
      MPI_Finalize();
      printf("rank %d ok (self)\n", rank);
      return 0;
   }
#ifdef PROBE_WITH_SHIM
   if (!strcmp(mode, "shim")) CHECK(HYPREDRV_AMD_CommInitMPI(MPI_COMM_WORLD) == 0);
#endif
   CHECK(HYPREDRV_Create(MPI_COMM_WORLD, &obj) == 0);
   CHECK(hda_comm_size() == size);
   if (size > 1) CHECK(!strcmp(hda_comm_name(), "host-callbacks") || !strcmp(hda_comm_name(), "rccl"));
   if (!strcmp(mode, "hang") && rank == 1)
   { /* never enters the collective below: its peers must not wait for ever */
      sleep(30);
      _exit(0);
   }
   /* the library's own partition code over the joined communicator: a 1-D chain of 5 rows per rank */
   {
      const int nloc = 5;
      long long *part = malloc(sizeof(long long) * (size + 1)), ghosts[2];
      int ng = 0, *sc = calloc(size, sizeof(int)), *rc = calloc(size, sizeof(int)), idx[8], tot = -1;
      for (int p = 0; p <= size; p++) part[p] = (long long)p * nloc;
      if (rank > 0) ghosts[ng++] = part[rank] - 1;
      if (rank < size - 1) ghosts[ng++] = part[rank + 1];
      CHECK(hda_halo_plan_host(nloc, part, ghosts, ng, sc, rc, idx, 8, &tot) == 0);
      CHECK(tot == ng);
      for (int p = 0; p < size; p++) CHECK(sc[p] == ((p == rank - 1 || p == rank + 1) ? 1 : 0) && rc[p] == sc[p]);
      /* what the left neighbour wants is my first row, the right one my last (grouped by ascending destination) */
      if (rank > 0) CHECK(idx[0] == 0);
      if (rank < size - 1) CHECK(idx[tot - 1] == nloc - 1);
      free(part); free(sc); free(rc);
   }
   if (!strcmp(mode, "device")) CHECK(hda_comm_selftest() == 0);
   CHECK(HYPREDRV_Destroy(&obj) == 0);
   CHECK(HYPREDRV_Finalize() == 0);
   if (strcmp(mode, "shim")) CHECK(hda_comm_size() == 1); /* the duplicate went back before MPI_Finalize */
   MPI_Finalize();
   printf("rank %d of %d ok (%s)\n", rank, size, mode);
   return 0;
}
