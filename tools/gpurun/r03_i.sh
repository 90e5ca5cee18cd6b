#!/bin/bash
# round 3: aggressive coarsening timing; BASELINE config 3 at full size on eight thread ranks; 64^3 trace after the small-operator dispatch
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03i}
rm -rf $O; mkdir -p $O
cd $R
for n in 128 256; do timeout -k 10 600 python tools/gpurun/gpurun_agg.py $n 2> $O/agg_$n.err | tee $O/agg_$n.log || { tail -20 $O/agg_$n.err; exit 1; }; done
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "config3_full_size" ) > $O/t_cfg3.log 2>&1 || { tail -60 $O/t_cfg3.log; exit 1; }
tail -6 $O/t_cfg3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace64 -o run -- python3 $R/bench.py --grid 64 --steps 5 --warmup 1 --no-cpu-baseline --no-kernel-table --no-plain-csr > $O/trace64.log 2>&1 || { tail -20 $O/trace64.log; exit 1; }
find $O -name "*kernel_trace.csv" -size +40M -delete
