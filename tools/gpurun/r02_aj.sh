#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02aj
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_hypredrv.py -x -q -m gpu -k "mgr or nested or darcy" > $O/t_hd.log 2>&1 || { tail -60 $O/t_hd.log; exit 1; }
tail -2 $O/t_hd.log
