/*
 * HYPREDRV_utils.h -- HYPREDRV_SAFE_CALL / HYPREDRV_SAFE_CALL_COMM, same contract as the
 * reference's include/HYPREDRV_utils.h:50-80: evaluate the call, and on a non-zero code
 * report file/line/function through HYPREDRV_SafeCallHandleError (which aborts).
 */
#ifndef HYPREDRV_UTILS_HEADER
#define HYPREDRV_UTILS_HEADER

#include "HYPREDRV.h"

#ifndef HYPREDRV_SAFE_CALL
#define HYPREDRV_SAFE_CALL(call) \
   do { HYPREDRV_SafeCallHandleError((call), MPI_COMM_WORLD, __FILE__, __LINE__, __func__); } while (0)
#endif
#ifndef HYPREDRV_SAFE_CALL_COMM
#define HYPREDRV_SAFE_CALL_COMM(comm, call) \
   do { HYPREDRV_SafeCallHandleError((call), (comm), __FILE__, __LINE__, __func__); } while (0)
#endif

#endif
