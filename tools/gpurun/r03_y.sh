#!/bin/bash
# round 3: eight thread ranks on irregular CSR / config-5 stand-in / MGR
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03y}
rm -rf $O; mkdir -p $O
cd $R
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "thread_rank" ) > $O/t_thr.log 2>&1 || { tail -80 $O/t_thr.log; exit 1; }
tail -6 $O/t_thr.log
