#!/bin/bash
# round 3: BASELINE config 3 as named (512^3, 2x2x2) on eight thread ranks over the ASYNCHRONOUS device transport (overlapped products on)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zw}
rm -rf $O; mkdir -p $O
cd $R
for tr in device host; do
( time HDA_THREAD_TRANSPORT=$tr PYTHONPATH=$R timeout -k 10 500 python tests/dist_worker.py threads $O/cfg3_$tr.json 512 2,2,2 pcg 0 ) > $O/run_$tr.log 2>&1 || { tail -20 $O/run_$tr.log; exit 1; }
grep real $O/run_$tr.log; cat $O/cfg3_$tr.json; echo
done
