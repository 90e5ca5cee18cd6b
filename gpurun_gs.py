import hypredrive_amd as h, time
A = h.lap7(128,128,128, want_rhs=False)
for d,u in ((18,18),(13,14)):
    h.sync(); t=time.time(); amg = h.Amg(A, h.AmgParams.default(relax_down=d, relax_up=u)); h.sync(); ts=time.time()-t
    r = h.solve_device(A, amg, nsolves=2)
    print(d,u,"setup %.3f s"%ts, r["iters"], r["solve_ms"], flush=True)
