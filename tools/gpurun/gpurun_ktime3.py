import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import hypredrive_amd as h
n = 256
A = h.lap7(n, n, n, want_rhs=False)
amg = h.Amg(A)
line = 'REORDER=' + os.environ.get('HDA_REORDER', 'default')
for l in range(3):
    for which, name in ((1, 'P'), (2, 'R')):
        M = amg.level_matrix(l, which)
        ms, by = h.time_kernel(0, M, None, 30)
        line += f" | L{l} {name} {ms*1e3:6.1f}us {by/ms/1e6:5.0f}"
ms, by = h.time_kernel(3, A, amg, 20); line += f" | vcycle {ms:.3f} ms"
print(line, flush=True)
