#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02j
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/tools/gpurun/gpurun_gs.py 128 8 > $O/gs.log 2>&1 || { tail -20 $O/gs.log; exit 1; }
grep hl1GS $O/gs.log
head -12 $O/trace/*/run_kernel_stats.csv 2>/dev/null || find $O/trace -name "*kernel_stats.csv" | head -1 | xargs head -12
find $O/trace -name "*kernel_trace.csv" -delete
