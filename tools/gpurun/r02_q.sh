#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02q
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1 || rocprofv3 -L > $O/avail.txt 2>&1
wc -l $O/avail.txt
grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" $O/avail.txt | sort -u | wc -l
