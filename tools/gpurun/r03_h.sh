#!/bin/bash
# round 3: aggressive coarsening -- parity tests, then 256^3 / 128^3 with aggressive.num_levels 0 / 1 / 2 through the API
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03h}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "aggressive or param_struct or cabi" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
cat > $O/agg.py <<'PY'
import sys, time, json
import numpy as np
import hypredrive_amd as hh
from hypredrive_amd import hypredrv as hd
n = int(sys.argv[1])
for agg in (0, 1, 2):
    y = "solver: pcg\npreconditioner:\n  amg:\n    aggressive:\n      num_levels: %d\n" % agg
    h = hd.Hypredrv(y)
    h.set_laplacian7((n, n, n))
    ts = []
    for rep in range(2):
        hh.sync(); t0 = time.perf_counter(); h.create_and_setup(); hh.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 0: h.destroy_solver()
    A, amg = hh._lib.borrow(h)
    g, o = amg.complexities
    rows = [amg.level_matrix(l, 0).dims[0] for l in range(amg.num_levels)]
    h.apply(); h.apply()
    hh.sync(); t0 = time.perf_counter()
    for _ in range(10): last = h.apply()
    hh.sync(); ms = (time.perf_counter() - t0) * 100.0
    print(json.dumps(dict(grid=n, agg=agg, iters=last["iters"], ms_per_solve=round(ms, 3), setup_ms=round(ts[1], 1), op_cx=round(o, 4), grid_cx=round(g, 4), levels=amg.num_levels, rows=rows)), flush=True)
    del A, amg
    h.destroy_solver(); h.close()
PY
for n in 128 256; do timeout -k 10 600 python $O/agg.py $n 2> $O/agg_$n.err | tee $O/agg_$n.log || { tail -20 $O/agg_$n.err; exit 1; }; done
