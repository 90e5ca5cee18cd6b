"""hypredrive_amd -- MI355X-native AMG-Krylov solve path behind the hypredrive API.

Thin ctypes binding over libhypredrv_amd.so (HIP kernels + C ABI, see include/*.h).
There is no CPU fallback: every compute call fails loudly when the HIP library or a
device is missing.
"""
from . import _lib  # noqa: F401
from ._lib import (AmgParams, KrylovParams, Csr, Amg, Ilu, Mgr, load, device_count, device_name,  # noqa: F401
                   lap7, pcg, gmres, fgmres, bicgstab, solve_device, time_kernel, pcg_iteration_bytes, memory_stats, format_bytes, probe_spmv, probe_read,
                   sync, LibraryError)
