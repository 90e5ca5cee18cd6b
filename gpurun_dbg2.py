import hypredrive_amd as h, numpy as np, sys, time
n = int(sys.argv[1]); mode = sys.argv[2]
A = h.lap7(n,n,n, want_rhs=False)
if mode == 'amg':
    t=time.time(); amg = h.Amg(A); print('amg ok', amg.num_levels, time.time()-t, flush=True)
elif mode == 'tk_amg':
    print(h.time_kernel(0, A, None, 5), flush=True)
    t=time.time(); amg = h.Amg(A); print('amg ok', amg.num_levels, time.time()-t, flush=True)
elif mode == 'solve':
    print(h.solve_timed(A), flush=True)
