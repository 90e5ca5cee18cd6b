#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02ai
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/tools/gpurun/gpurun_gs.py 128 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_trace.csv" -size +40M -delete
python3 - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
grep hl1GS $O/trace.log
