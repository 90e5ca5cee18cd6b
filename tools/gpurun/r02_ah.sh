#!/bin/bash
# larger single-GPU grids with the end-of-round tree
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02ah
rm -rf $O; mkdir -p $O
cd $R
for n in 384 512; do
timeout -k 10 500 python bench.py --grid $n --steps 3 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_$n.json 2> $O/bench_$n.err || { tail -30 $O/bench_$n.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_$n.json'))
print('grid $n', {k:d[k] for k in ('value','ms_per_step','iters','setup_ms','setup_cold_ms','hbm_in_use_gb','hbm_peak_gb')}, 'dom', d['roofline']['avg_ms'], d['roofline']['frac'])"
done
