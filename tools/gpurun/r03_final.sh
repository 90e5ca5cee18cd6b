#!/bin/bash
# round 3, end: the whole GPU suite, then the evidence run (bench line with its own counter passes, kernel trace, FETCH / WRITE passes)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
T=${1:-r03c}
mkdir -p $R/gpurun_out/${T}_suite
cd $R
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $R/gpurun_out/${T}_suite/t_all.log 2>&1 || { tail -60 $R/gpurun_out/${T}_suite/t_all.log; exit 1; }
tail -5 $R/gpurun_out/${T}_suite/t_all.log
bash tools/gpurun/r03_k.sh $T
