#!/bin/bash
# round 5: kernel trace of series B (the reference's CPU defaults on row blocks) at 256^3 in rank-block numbering: per-kernel totals
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r05sb}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/tools/series_b.py --grid 256 --rank-grid 8 --steps 3 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
cd $R
python3 tools/trace_by_operator.py $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/kernel_by_operator.csv
find $O/trace -name "*kernel_trace.csv" -delete
head -45 $O/kernel_by_operator.csv | cut -c1-150
