import hypredrive_amd as h, sys, os
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = h.lap7(n,n,n, want_rhs=False)
amg = h.Amg(A)
print('mode', os.environ.get('HDA_SPMV','stream'), 'levels', amg.num_levels, flush=True)
for l in range(min(amg.num_levels, 5)):
    for which,nm in ((0,'A'),(1,'P'),(2,'R')):
        if which and l >= amg.num_levels-1: continue
        M = amg.level_matrix(l, which)
        nr, nc, nnz = M.dims
        line = f"L{l} {nm}: {nr}x{nc} nnz={nnz} avg={nnz/max(nr,1):.1f} |"
        for kind,name in ((0,'spmv'),(1,'jac'),(2,'res')):
            if which and kind: continue
            ms, by = h.time_kernel(kind, M, None, 30)
            line += f" {name} {ms*1e3:8.1f}us {by/ms/1e6:7.0f}GB/s |"
        print(line, flush=True)
ms, by = h.time_kernel(3, A, amg, 20); print(f"vcycle {ms:.3f} ms {by/ms/1e6:.0f} GB/s", flush=True)
r = h.solve_device(A, amg, nsolves=3); print(r, flush=True)
