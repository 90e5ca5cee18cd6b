#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02af
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for w in 1 2; do
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_$w.json 2> $O/bench_$w.err || { tail -30 $O/bench_$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_$w.json'))
print('run $w', {k:d[k] for k in ('ms_per_step','iters','setup_ms')}, 'P', d['level0_prolongation']['avg_ms'], 'R', d['level0_restriction']['avg_ms'], 'vcycle', d['kernels']['vcycle']['ms'])"
done
