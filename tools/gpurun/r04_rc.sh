#!/bin/bash
# round 4: row-class kernel variants against each other (env switches in FORMS): parity first, then level-0 kernels alone, then the headline bench
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04rc
rm -rf $O; mkdir -p $O
cd $R
FORMS=${FORMS:-"HDA_ROWCLASS_X2=0 HDA_ROWCLASS_X2=1"}
for f in $FORMS; do
  env $f timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_class or rowclass or coded or stencil or spmv" > $O/t_$f.log 2>&1 || { tail -40 $O/t_$f.log; exit 1; }
  tail -1 $O/t_$f.log
  echo "$f level 0 alone:"; env $f timeout -k 10 200 python tools/time_level0.py 256 || exit 1
done
for rep in 1 2; do
  for f in $FORMS; do
    env $f timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --no-plain-csr --no-cpu-defaults --no-side-configs --no-aggressive --no-kernel-table \
        > $O/bench_$f.$rep.json 2> $O/bench_$f.$rep.err || { tail -30 $O/bench_$f.$rep.err; exit 1; }
    python3 - $O/bench_$f.$rep.json $f <<'PYEOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], "ms_per_step", round(d["ms_per_step"], 3), "iters", d["iters"], "level0_spmv", round(d["level0_spmv"]["avg_ms"], 4), "roofline", round(d["roofline"]["avg_ms"], 4), "seam", d["seam"]["ms_per_step"])
PYEOF
  done
done
