#!/bin/bash
# round 3: where does a hybrid Gauss-Seidel solve spend its time?  timing + kernel trace at 128^3
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03gs}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/gpurun/gs_probe.py 128 3 > $O/gs.log 2>&1 || { tail $O/gs.log; exit 1; }
cat $O/gs.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/tools/gpurun/gs_probe.py 128 1 > $O/trace.log 2>&1 || { tail $O/trace.log; exit 1; }
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/trace/run_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,2))
PY
find $O/trace -name "*kernel_trace.csv" -size +40M -delete
