#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03l}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "aggressive or single_reduction or eight_ranks" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
