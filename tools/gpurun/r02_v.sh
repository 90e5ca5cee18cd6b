#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02v
rm -rf $O; mkdir -p $O
cd $R
HDA_VERBOSE=1 timeout -k 10 300 python tools/gpurun/gpurun_setup.py 256 > $O/setup.log 2> $O/setup.err || { tail -30 $O/setup.err; exit 1; }
cat $O/setup.log
grep -c hda $O/setup.err
