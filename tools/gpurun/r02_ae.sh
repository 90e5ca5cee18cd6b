#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02ae
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for w in 1 0; do
HDA_SELL=$w timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_s$w.json 2> $O/bench_s$w.err || { tail -30 $O/bench_s$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_s$w.json'))
print('sell $w', {k:d[k] for k in ('ms_per_step','iters','setup_ms')}, 'P', d['level0_prolongation']['avg_ms'], d['level0_prolongation']['format'], 'R', d['level0_restriction']['avg_ms'], 'vcycle', d['kernels']['vcycle']['ms'])"
done
