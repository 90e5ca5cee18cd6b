#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02r
rm -rf $O; mkdir -p $O
cd $R
HDA_WINDOW=1 HDA_WINDOW_MIN_NNZ=0 HDA_WINDOW_RATIO=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for w in 1 0 1 0; do
HDA_WINDOW=$w timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_w$w.json 2> $O/bench_w$w.err || { tail -30 $O/bench_w$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_w$w.json'))
print('window $w', {k:d[k] for k in ('ms_per_step','iters','setup_ms')}, 'dom', d['roofline']['avg_ms'], d['roofline']['frac'], 'vcycle', d['kernels']['vcycle']['ms'])"
done
HDA_WINDOW=1 HDA_VERBOSE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-plain-csr --no-kernel-table 2>&1 | grep -i "windowed" | head -12
