#!/bin/bash
# round 3: the device-direct (asynchronous, RCCL-like) thread transport: its tests, then what 8 ranks time-sharing one GPU cost
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03z}
rm -rf $O; mkdir -p $O
cd $R
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "thread_rank or eight_ranks" ) > $O/t_thr.log 2>&1 || { tail -80 $O/t_thr.log; exit 1; }
tail -4 $O/t_thr.log
for cfg in "host default" "device default" "device 0"; do
  set -- $cfg
  ov=""; [ "$2" != "default" ] && ov="HDA_OVERLAP=$2"
  env HDA_THREAD_TRANSPORT=$1 $ov timeout -k 10 300 python tools/gpurun/gpurun_thread_timing.py 256 2,2,2 5 >> $O/timing.jsonl 2>> $O/timing.err || { tail -20 $O/timing.err; exit 1; }
done
timeout -k 10 300 python tools/gpurun/gpurun_thread_timing.py 256 1,1,1 5 >> $O/timing.jsonl 2>> $O/timing.err || { tail -20 $O/timing.err; exit 1; }
HDA_THREAD_TRANSPORT=device timeout -k 10 300 python tools/gpurun/gpurun_thread_timing.py 256 1,1,2 5 >> $O/timing.jsonl 2>> $O/timing.err || { tail -20 $O/timing.err; exit 1; }
cat $O/timing.jsonl
