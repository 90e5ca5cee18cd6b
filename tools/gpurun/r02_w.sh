#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02w
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
HDA_VERBOSE=1 timeout -k 10 300 python tools/gpurun/gpurun_setup.py 256 > $O/setup.log 2> $O/setup.err || { tail -30 $O/setup.err; exit 1; }
cat $O/setup.log
timeout -k 10 300 python tools/gpurun/gpurun_setup.py 256 > $O/setup_q.log 2> $O/setup_q.err || { tail -30 $O/setup_q.err; exit 1; }
cat $O/setup_q.log
