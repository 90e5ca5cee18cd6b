import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import hypredrive_amd as h, sys, os
n = 256
A = h.lap7(n,n,n, want_rhs=False)
amg = h.Amg(A)
line = 'VAR=' + os.environ.get('HDA_VAR','0')
for l in range(3):
    M = amg.level_matrix(l, 0)
    for kind,name in ((0,'spmv'),(1,'jac')):
        ms, by = h.time_kernel(kind, M, None, 30)
        line += f" | L{l} {name} {ms*1e3:7.1f}us {by/ms/1e6:5.0f}"
ms, by = h.time_kernel(3, A, amg, 20); line += f" | vcycle {ms:.3f} ms"
print(line, flush=True)
