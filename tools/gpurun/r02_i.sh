#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02i
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','iters','setup_ms')}, 'dom', d['roofline']['avg_ms'], 'k1', d['level0_spmv']['avg_ms'], 'P', d['level0_prolongation']['avg_ms'], 'R', d['level0_restriction']['avg_ms'], 'kern', {k:round(v['ms'],4) for k,v in d['kernels'].items()})"
HDA_VERBOSE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-plain-csr --no-kernel-table 2>&1 | grep -i "value-coded\|row-class" | head
