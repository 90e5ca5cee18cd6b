#!/bin/bash
# round 3: random operators through the setup incl. aggressive levels, device vs oracle bit for bit, under guard + poison
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03m}
rm -rf $O; mkdir -p $O
cd $R
HDA_GUARD=1 HDA_POISON=1 timeout -k 10 1000 python tests/fuzz_hierarchies.py 200 > $O/fuzz.log 2>&1; rc=$?
tail -15 $O/fuzz.log
exit $rc
