#!/bin/bash
# round 3: where the hybrid Gauss-Seidel solve goes at 128^3 (kernel trace: per-level launches against single-workgroup runs)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03r}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python tools/gpurun/gpurun_gs.py 128 | tee $O/gs.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/tools/gpurun/gpurun_gs.py 128 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/trace/run_kernel_stats.csv")))
for r in rows[:14]:
    print("%-60s calls %7s avg %9.2f us total %10.1f ms"%(r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
find $O -name "*kernel_trace.csv" -size +40M -delete
