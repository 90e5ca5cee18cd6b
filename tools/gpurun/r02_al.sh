#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02al
rm -rf $O; mkdir -p $O
cd $R
HDA_VCYCLE_GRAPH=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for g in 1 0; do
for n in 64 96 128 160; do
HDA_VCYCLE_GRAPH=$g timeout -k 10 300 python bench.py --grid $n --steps 5 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/bench_${g}_$n.json 2> $O/bench_${g}_$n.err || { tail -30 $O/bench_${g}_$n.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_${g}_$n.json'))
print('graph $g grid $n', {k:d[k] for k in ('value','ms_per_step','iters')}, 'ms/iter', round(d['ms_per_step']/d['iters'],3))"
done; done
