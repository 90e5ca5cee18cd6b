/* The MPICH-ABI handle values libhypredrv_amd.so uses without <mpi.h> (hypredrive_amd/csrc/hda_mpi.cpp) against this MPI's header. */
#include <mpi.h>
#include <stdio.h>
_Static_assert(sizeof(MPI_Comm) == sizeof(int) && sizeof(MPI_Datatype) == sizeof(int) && sizeof(MPI_Op) == sizeof(int) && sizeof(MPI_Request) == sizeof(int), "int handles");
_Static_assert(MPI_COMM_WORLD == 0x44000000 && MPI_COMM_SELF == 0x44000001 && MPI_COMM_NULL == 0x04000000, "communicators");
_Static_assert(MPI_BYTE == 0x4c00010d && MPI_DOUBLE == 0x4c00080b && MPI_LONG_LONG == 0x4c000809, "datatypes");
_Static_assert(MPI_SUM == 0x58000003 && MPI_MAX == 0x58000001, "operations");
int main(void)
{
   char v[MPI_MAX_LIBRARY_VERSION_STRING];
   int  n = 0;
   if (MPI_IN_PLACE != (void *)-1 || MPI_STATUS_IGNORE != (MPI_Status *)1 || MPI_MAX_LIBRARY_VERSION_STRING > 8192 + 64) return 1;
   MPI_Get_library_version(v, &n);
   printf("%.60s\n", v);
   return 0;
}
