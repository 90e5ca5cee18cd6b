#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02am
rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2; do for w in 1 0; do
HDA_FUSE_FIRST_SWEEP=$w timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-plain-csr > $O/bench_$w$rep.json 2> $O/bench_$w$rep.err || { tail -30 $O/bench_$w$rep.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_$w$rep.json'))
print('fuse $w', {k:d[k] for k in ('ms_per_step','iters')}, 'R', d['level0_restriction']['avg_ms'], 'vcycle', d['kernels']['vcycle']['ms'], 'seam', d['seam']['ms_per_step'])"
done; done
