#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02l
rm -rf $O; mkdir -p $O
cd $R
for w in 2048 1536 1024 768 512 2048; do
HDA_STREAM_GRID=$w timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/bench_g$w.json 2> $O/bench_g$w.err || { tail -30 $O/bench_g$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_g$w.json'))
print('grid $w', {k:d[k] for k in ('ms_per_step','iters')}, 'dom', d['roofline']['avg_ms'], d['roofline']['frac'], 'P', d['level0_prolongation']['avg_ms'], 'R', d['level0_restriction']['avg_ms'])"
done
timeout -k 10 300 python -m pytest tests/test_gpu_hypredrv.py -x -q -m gpu -k "statistics_level_2" 2>&1 | tail -2
