import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import hypredrive_amd as hh
from hypredrive_amd import hypredrv as hd
from oracle import oracle_ffi as orc   # (test-side generator of the host arrays only)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
A, b = orc.lap7(n, n, n)
ip = np.asarray(A.rowptr, dtype=np.int64); ix = np.asarray(A.col, dtype=np.int64); v = np.asarray(A.val, dtype=np.float64)
N = n ** 3
print("rows", N, "nnz", len(ix), flush=True)
h = hd.Hypredrv("solver: pcg\npreconditioner: amg\n")
for rep in range(3):
    hh.sync(); t0 = time.perf_counter()
    h.set_matrix_csr(0, N - 1, ip, ix, v)
    hh.sync(); t1 = time.perf_counter()
    h.set_rhs_array(0, N - 1, b)
    hh.sync(); t2 = time.perf_counter()
    print(f"SetMatrixFromCSR {1e3*(t1-t0):.1f} ms ({(len(ix)*16+len(ip)*8)/(t1-t0)/1e9:.2f} GB/s of host arrays), SetRHSFromArray {1e3*(t2-t1):.1f} ms", flush=True)
h.finish_system()
h.create_and_setup()
r = h.apply()
print("iters", r["iters"], r["converged"])
