/*
 * hda_mpi.h -- the MPI_Comm type seen by HYPREDRV.h / HYPRE.h.
 *
 * The MI355X library does not link MPI: one process drives one GPU and ranks talk over
 * RCCL (see HYPREDRV_AMD_CommInit in HYPREDRV.h).  Programs that do use MPI (the reference's
 * example drivers) define HYPREDRV_AMD_USE_MPI and get the real <mpi.h>; the library only
 * ever stores the handle.  MPICH-ABI handles are ints, which is what the stub mirrors
 * (reference: include/HYPREDRV.h:11 includes <mpi.h> unconditionally).
 */
#ifndef HDA_MPI_H
#define HDA_MPI_H
#ifdef HYPREDRV_AMD_USE_MPI
#include <mpi.h>
#else
#ifndef MPI_COMM_WORLD
typedef int MPI_Comm;
#define MPI_COMM_WORLD ((MPI_Comm)0x44000000)
#define MPI_COMM_SELF ((MPI_Comm)0x44000001)
#define MPI_COMM_NULL ((MPI_Comm)0x04000000)
#define HDA_MPI_STUB 1
#endif
#endif
#endif
