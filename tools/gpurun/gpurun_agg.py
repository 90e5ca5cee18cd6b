"""aggressive.num_levels 0 / 1 / 2 beside each other through the HYPREDRV_* API: iterations, time per solve, setup, complexities."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hypredrive_amd as hh  # noqa: E402
from hypredrive_amd import hypredrv as hd  # noqa: E402

n = int(sys.argv[1])
for agg in (0, 1, 2):
    y = "solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: %d\n" % agg
    h = hd.Hypredrv(y)
    h.set_laplacian7((n, n, n))
    ts = []
    for rep in range(2):
        hh.sync()
        t0 = time.perf_counter()
        h.create_and_setup()
        hh.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 0:
            h.destroy_solver()
    A, amg = hh._lib.borrow(h)
    g, o = amg.complexities
    rows = [amg.level_matrix(l, 0).dims[0] for l in range(amg.num_levels)]
    h.apply()
    h.apply()
    hh.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        last = h.apply()
    hh.sync()
    ms = (time.perf_counter() - t0) * 100.0
    print(json.dumps(dict(grid=n, agg=agg, iters=last["iters"], ms_per_solve=round(ms, 3), setup_ms=round(ts[1], 1), op_cx=round(o, 4),
                          grid_cx=round(g, 4), levels=amg.num_levels, rows=rows)), flush=True)
    del A, amg
    h.destroy_solver()
    h.close()
