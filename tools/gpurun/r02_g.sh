#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02g
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for w in 1 0; do
HDA_ROWCLASS=$w timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_rc$w.json 2> $O/bench_rc$w.err || { tail -30 $O/bench_rc$w.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_rc$w.json'))
print('rowclass $w', {k:d[k] for k in ('value','ms_per_step','iters','setup_ms')}, 'dom', d['roofline']['avg_ms'], 'k1', d['level0_spmv']['avg_ms'], d['level0_spmv']['format_frac'], 'kern', {k:round(v['ms'],4) for k,v in d['kernels'].items()})"
done
