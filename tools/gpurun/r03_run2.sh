#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03run2}
rm -rf $O; mkdir -p $O
cd $R
HDA_VERBOSE_X=1 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "eight_ranks_2x2x2" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
