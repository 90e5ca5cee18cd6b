#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
for f in 0 2 4; do echo "FL variant $f"; HDA_GS_FL=$f python3 tools/gpurun/gpurun_gs.py 128 8; done
