#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02t
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for r in 0.5 0.7; do
HDA_WINDOW_RATIO=$r timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_r$r.json 2> $O/bench_r$r.err || { tail -30 $O/bench_r$r.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_r$r.json'))
print('ratio limit $r', {k:d[k] for k in ('ms_per_step','iters','setup_ms')}, 'dom', d['roofline']['avg_ms'], d['roofline']['frac'], 'P', d['level0_prolongation']['avg_ms'], 'R', d['level0_restriction']['avg_ms'], 'vcycle', d['kernels']['vcycle']['ms'])"
done
HDA_WINDOW_RATIO=0.7 HDA_VERBOSE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-plain-csr --no-kernel-table 2>&1 | grep -i "windowed" | head -8
