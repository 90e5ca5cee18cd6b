#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02ag
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gs or gauss or hybrid or mgr or relax or ilu" > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for p in 1 0; do
HDA_GS_PIPE=$p timeout -k 10 300 python tools/gpurun/gpurun_gs.py 128 > $O/gs_$p.log 2>&1 || { tail -20 $O/gs_$p.log; exit 1; }
echo "pipe $p: $(tail -1 $O/gs_$p.log)"
done
