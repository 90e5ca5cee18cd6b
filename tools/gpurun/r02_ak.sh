#!/bin/bash
# small grids: where a solve stops being bandwidth-bound
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02ak
rm -rf $O; mkdir -p $O
cd $R
for n in 64 96 128 160 192; do
timeout -k 10 300 python bench.py --grid $n --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench_$n.json 2> $O/bench_$n.err || { tail -30 $O/bench_$n.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_$n.json'))
print('grid $n', {k:d[k] for k in ('value','ms_per_step','iters','setup_ms')}, 'ms/iter', round(d['ms_per_step']/d['iters'],3), 'vcycle', d['kernels']['vcycle']['ms'])"
done
