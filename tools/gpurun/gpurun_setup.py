import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import time
import hypredrive_amd as h
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = h.lap7(n, n, n, want_rhs=False)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for rep in range(reps):
    h.sync(); t0 = time.perf_counter()
    amg = h.Amg(A)
    h.sync(); print(f"setup {rep}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    del amg
