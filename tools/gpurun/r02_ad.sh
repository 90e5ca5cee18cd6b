#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02ad
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
timeout -k 10 300 python tools/gpurun/gpurun_setup.py 256 3 > $O/setup_q.log 2> $O/setup_q.err || { tail -30 $O/setup_q.err; exit 1; }
cat $O/setup_q.log
HDA_VERBOSE=1 timeout -k 10 300 python tools/gpurun/gpurun_setup.py 256 2 > $O/setup_v.log 2> $O/setup_v.err || { tail -30 $O/setup_v.err; exit 1; }
grep "setup level" $O/setup_v.err | tail -7 | head -4 | cut -c1-200
awk 'NR>100' $O/setup_v.err | grep -E "renumbering|value-coded|windowed|level 6: rap|setup level 6" | tail -14 | cut -c1-150
