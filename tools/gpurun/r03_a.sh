#!/bin/bash
# round 3, first GPU call: new tests first (thread ranks 2x2x2, nested PCG), then the whole suite
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03a}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "eight_ranks or nested_pcg" > $O/t_new.log 2>&1 || { tail -60 $O/t_new.log; exit 1; }
tail -3 $O/t_new.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -3 $O/t_all.log
