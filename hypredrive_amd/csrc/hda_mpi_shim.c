/* hda_mpi_shim.c -- application-side binding for MPI implementations whose MPI_Comm the library cannot read by itself.
 *
 * libhypredrv_amd.so joins the ranks of an MPI program on its own when the process's MPI is of the MPICH ABI family (hda_mpi.cpp).
 * For any other MPI (Open MPI: pointer handles) compile THIS file with the application's own mpicc -- it then sees the real
 * <mpi.h> -- link it into the application and call
 *
 *       HYPREDRV_AMD_CommInitMPI(comm);        // once, after MPI_Init / HYPREDRV_Initialize, before HYPREDRV_Create(comm, ...)
 *
 * It connects the ranks with the library's public launcher interface (include/HYPREDRV.h): RCCL when every rank of a host has a
 * GPU of its own (unique id from rank 0 by MPI_Bcast), otherwise the host-staged transport with MPI_Allreduce / MPI_Alltoallv
 * behind the two callbacks.  Reference: the communicator semantics of src/HYPREDRV.c:1014-1041.
 *
 *       mpicc -I<repo>/include -c hda_mpi_shim.c
 */
#include <mpi.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#define HYPREDRV_AMD_USE_MPI 1
#include "HYPREDRV.h"

static MPI_Comm shim_comm = MPI_COMM_NULL;

static int shim_allreduce(void *buf, long count, int dtype, int op)
{
   if (count > INT_MAX) return 2;
   return MPI_Allreduce(MPI_IN_PLACE, buf, (int)count, dtype == 0 ? MPI_DOUBLE : MPI_LONG_LONG, op == 0 ? MPI_SUM : MPI_MAX, shim_comm) != MPI_SUCCESS;
}

static int shim_alltoallv(const void *send, const long *sb, void *recv, const long *rb)
{
   int np = 1, rc = 0;
   MPI_Comm_size(shim_comm, &np);
   int *c = malloc(sizeof(int) * 4 * (size_t)np), *sc = c, *sd = c + np, *rcv = c + 2 * np, *rd = c + 3 * np;
   long so = 0, ro = 0;
   for (int p = 0; p < np; p++)
   {
      if (sb[p] > INT_MAX || rb[p] > INT_MAX || so > INT_MAX || ro > INT_MAX) { free(c); return 4; }
      sc[p] = (int)sb[p]; sd[p] = (int)so; rcv[p] = (int)rb[p]; rd[p] = (int)ro;
      so += sb[p]; ro += rb[p];
   }
   rc = MPI_Alltoallv(send, sc, sd, MPI_BYTE, recv, rcv, rd, MPI_BYTE, shim_comm) != MPI_SUCCESS;
   free(c);
   return rc;
}

uint32_t HYPREDRV_AMD_CommInitMPI(MPI_Comm comm)
{
   int rank = 0, size = 1, local = 0, nlocal = 1;
   MPI_Comm node;
   MPI_Comm_dup(comm, &shim_comm);
   MPI_Comm_rank(shim_comm, &rank);
   MPI_Comm_size(shim_comm, &size);
   MPI_Comm_split_type(shim_comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
   MPI_Comm_rank(node, &local);
   MPI_Comm_size(node, &nlocal);
   MPI_Comm_free(&node);
   /* HDA_SHIM_GPUS_PER_NODE: how many GPUs a node offers (default: assume one per local rank -> RCCL); with fewer GPUs than local
    * ranks the ranks share devices and the data path is staged through the host */
   const char *e = getenv("HDA_SHIM_GPUS_PER_NODE");
   const int gpus = e ? atoi(e) : nlocal;
   int shared = gpus < nlocal, any = 0;
   MPI_Allreduce(&shared, &any, 1, MPI_INT, MPI_MAX, shim_comm);
   if (any) return HYPREDRV_AMD_CommInitCallbacks(rank, size, gpus > 0 ? local % gpus : -1, shim_allreduce, shim_alltoallv);
   char uid[128];
   memset(uid, 0, sizeof(uid));
   uint32_t code = 0;
   if (rank == 0) code = HYPREDRV_AMD_CommGetUniqueId(uid);
   MPI_Bcast(&code, 1, MPI_UNSIGNED, 0, shim_comm);
   if (code) return code;
   MPI_Bcast(uid, 128, MPI_BYTE, 0, shim_comm);
   return HYPREDRV_AMD_CommInit(rank, size, local, uid);
}
