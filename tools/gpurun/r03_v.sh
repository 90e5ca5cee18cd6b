#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03v}
rm -rf $O; mkdir -p $O
cd $R
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "hmis" ) > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -6 $O/t.log
