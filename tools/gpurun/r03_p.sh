#!/bin/bash
# round 3: general-matrix path (HDA_CODED=0): does windowing the level-0 operator (0.71 distinct columns per entry) pay? same-box A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03p}
rm -rf $O; mkdir -p $O
cd $R
run() { tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive > $O/b_$tag.json 2> $O/b_$tag.err || { tail -30 $O/b_$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_$tag.json'))
print('$tag', round(d['ms_per_step'],4), d['iters'], 'k1', d['level0_spmv']['kernel'], round(d['level0_spmv']['avg_ms'],4), 'P0', round(d['level0_prolongation']['avg_ms'],4), 'R0', round(d['level0_restriction']['avg_ms'],4))"
}
for rep in 1 2; do
run plain_$rep HDA_CODED=0
run plainwin08_$rep HDA_CODED=0 HDA_WINDOW_RATIO=0.8
run win08_$rep HDA_WINDOW_RATIO=0.8
run default_$rep HDA_WIN_PF=1
done
