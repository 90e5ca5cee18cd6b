#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02m
rm -rf $O; mkdir -p $O
cd $R
for w in 1 2 1 2; do
HDA_RC_H=$w timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_h$w.json 2> $O/bench_h$w.err || { tail -30 $O/bench_h$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_h$w.json'))
print('H $w', {k:d[k] for k in ('ms_per_step','iters')}, 'k1', d['level0_spmv']['avg_ms'], 'kern', {k:round(v['ms'],4) for k,v in d['kernels'].items()})"
done
HDA_RC_H=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
