import hypredrive_amd as h, sys, os, time
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = h.lap7(n,n,n, want_rhs=False)
h.sync(); t=time.time(); amg = h.Amg(A); h.sync(); print('setup s', time.time()-t, 'levels', amg.num_levels, amg.complexities, flush=True)
h.sync(); t=time.time(); amg2 = h.Amg(A); h.sync(); print('setup(2nd) s', time.time()-t, flush=True)
r = h.solve_device(A, amg, nsolves=2); print(r, flush=True)
print('mem', h.memory_stats())
