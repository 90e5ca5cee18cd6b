// hda_mpi_join.h -- joining the ranks of an MPI program through the communicator handed to the library (hda_mpi.cpp).
#pragma once
namespace hda {
// Called wherever the API receives a communicator.  No-op unless the process is an initialised MPI program of the MPICH ABI
// family, the communicator has more than one rank and no HYPREDRV_AMD_CommInit* call has been made; throws hda::Error on failure.
void mpi_autojoin(int comm);
void mpi_leave();                                       // HYPREDRV_Finalize / HYPRE_Finalize
bool mpi_joined();
bool mpi_comm_size(int comm, int *rank, int *size);     // false when the handle cannot be read (no MPI / foreign ABI)
bool mpi_abort(int comm, int code);                     // MPI_Abort when the process runs under MPI; false otherwise
} // namespace hda
