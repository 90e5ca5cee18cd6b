#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02ag
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gs or gauss or hybrid or mgr or relax" > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
timeout -k 10 300 python tools/gpurun/gpurun_gs.py 128 > $O/gs.log 2>&1 || { tail -20 $O/gs.log; exit 1; }
tail -1 $O/gs.log
