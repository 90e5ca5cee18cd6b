#!/bin/bash
# round 3: window ratio limit (distinct columns per entry an operator may have to be windowed): 0.5 (default) / 0.6 / 0.65 / 0.8
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03q}
rm -rf $O; mkdir -p $O
cd $R
run() { tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive > $O/b_$tag.json 2> $O/b_$tag.err || { tail -30 $O/b_$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_$tag.json'))
print('$tag', round(d['ms_per_step'],4), round(d['solve_timer_ms'],4), d['iters'], 'setup', round(d['setup_ms'],1), 'P0', round(d['level0_prolongation']['avg_ms'],4), 'R0', round(d['level0_restriction']['avg_ms'],4), 'dom', round(d['roofline']['avg_ms'],4))"
}
for rep in 1 2 3; do
run r050_$rep HDA_WINDOW_RATIO=0.5
run r060_$rep HDA_WINDOW_RATIO=0.6
run r065_$rep HDA_WINDOW_RATIO=0.65
run r080_$rep HDA_WINDOW_RATIO=0.8
done
