#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
for n in 32 46 58; do timeout -k 10 300 python tools/gpurun/gpurun_gs.py $n 10; done
