import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hypredrive_amd as hh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = hh.lap7(n, n, n)
for kind, name in ((0, "spmv"), (1, "l1_jacobi"), (2, "residual")):
    ms, by = hh.time_kernel(kind, A, None, 50)
    print(name, round(ms * 1000, 1), "us", flush=True)
