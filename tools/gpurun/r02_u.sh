#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02u
rm -rf $O; mkdir -p $O
cd $R
for w in 50000 0; do
HDA_REORDER=$w timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/bench_ro$w.json 2> $O/bench_ro$w.err || { tail -30 $O/bench_ro$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_ro$w.json'))
print('reorder $w', {k:d[k] for k in ('ms_per_step','iters','setup_ms')}, 'dom', d['roofline']['avg_ms'], 'P', d['level0_prolongation']['avg_ms'], 'R', d['level0_restriction']['avg_ms'])"
done
