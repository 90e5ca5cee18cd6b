#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02k
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_hypredrv.py -x -q -m gpu -k "spe10 or null_space or ex8" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
for n in 128 160; do
timeout -k 10 300 python bench.py --workload aniso --grid $n --steps 3 --warmup 1 > $O/aniso$n.json 2> $O/aniso$n.err || { tail -30 $O/aniso$n.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/aniso$n.json'))
print('aniso $n', {k:d[k] for k in ('value','ms_per_step','iters','setup_ms','num_levels','operator_complexity')}, 'dom', d['roofline']['avg_ms'], d['roofline']['frac'], 'k1', d['level0_spmv']['avg_ms'], d['level0_spmv']['csr_equiv_frac'], 'res0', d['level0_residual']['avg_ms'], d['level0_residual']['csr_equiv_frac'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --workload aniso --grid 160 --steps 3 --warmup 1 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs head -14 | cut -c1-180
find $O/trace -name "*kernel_trace.csv" -delete
