import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import hypredrive_amd as h, time, sys, os
print(h.device_name(), flush=True)
sizes = [int(a) for a in sys.argv[1:]] or [32, 48, 64]
for n in sizes:
    A = h.lap7(n,n,n, want_rhs=False)
    for kind,name in ((0,'spmv'),(1,'jacobi'),(2,'resid')):
        ms, by = h.time_kernel(kind, A, None, 50)
        print(f"n={n} {name}: {ms:.4f} ms  {by/ms/1e6:.1f} GB/s", flush=True)
    t=time.time(); r = h.solve_timed(A); print(n, r, 'wall %.2f'%(time.time()-t), flush=True)
