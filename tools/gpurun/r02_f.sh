#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02f
rm -rf $O; mkdir -p $O
cd $R
for w in 1 0 1 0; do
HDA_SPMV_PIPE=$w timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_p$w.json 2> $O/bench_p$w.err || { tail -30 $O/bench_p$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_p$w.json'))
print('pipe $w', {k:d[k] for k in ('value','ms_per_step','iters')}, 'dom', d['roofline']['avg_ms'], d['roofline']['frac'], 'k1', d['level0_spmv']['avg_ms'], 'P', d['level0_prolongation']['avg_ms'], 'R', d['level0_restriction']['avg_ms'], 'vcycle', d['kernels']['vcycle']['ms'])"
done
HDA_SPMV_PIPE=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
