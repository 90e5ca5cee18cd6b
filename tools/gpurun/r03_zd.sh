#!/bin/bash
# round 3: the two fuzzers as tests of the GPU suite
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03ze}
rm -rf $O; mkdir -p $O
cd $R
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "random_partitions or random_hierarchies" ) > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -5 $O/t.log
