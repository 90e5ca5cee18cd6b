#!/bin/bash
# evidence for the widened rows: kernel statistics of an MGR solve (the reference's Darcy driver, unmodified)
# and of AMG-PCG with ILU(0) as level-0 smoother
set -o pipefail
TAG=${1:-r01j}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mgr -o run -- $R/oracle/_ref/darcy_ref -v 1 -n 96 96 48 > $O/mgr.log 2>&1 || { tail -20 $O/mgr.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ilu -o run -- python3 $R/tools/gpurun/gpurun_ilu.py 128 > $O/ilu.log 2>&1 || { tail -20 $O/ilu.log; exit 1; }
find $O -name "*kernel_trace.csv" -size +20M -delete
grep -E "Unknowns|pressure L2|^\|  *0 " $O/mgr.log; tail -4 $O/ilu.log
