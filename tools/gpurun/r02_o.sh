#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02o
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "gauss or hmis or cpudefaults or cpu_defaults or mgr or reuse or ilu" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
python3 tools/gpurun/gpurun_gs.py 128 8
python3 tools/gpurun/gpurun_gs.py 128 10 || true
