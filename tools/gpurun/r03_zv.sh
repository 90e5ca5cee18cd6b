#!/bin/bash
# round 3, final tree: long fuzz runs -- 300 random hierarchies (guard + poison) and 400 random partitions on thread ranks
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zv}
rm -rf $O; mkdir -p $O
cd $R
HDA_GUARD=1 HDA_POISON=1 PYTHONPATH=$R timeout -k 10 900 python tests/fuzz_hierarchies.py 300 > $O/fuzz_h.log 2>&1; rc1=$?
tail -2 $O/fuzz_h.log; grep -c MISMATCH $O/fuzz_h.log
PYTHONPATH=$R timeout -k 10 900 python tests/fuzz_ranks.py 400 20000 > $O/fuzz_r.jsonl 2> $O/fuzz_r.err; rc2=$?
grep -c '"ok": true' $O/fuzz_r.jsonl; grep '"ok": false' $O/fuzz_r.jsonl | cut -c1-600 | head -5; tail -1 $O/fuzz_r.jsonl
[ $rc1 -eq 0 ] && [ $rc2 -eq 0 ]
