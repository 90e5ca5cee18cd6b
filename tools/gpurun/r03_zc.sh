#!/bin/bash
# round 3: random partitions x random irregular matrices on 2-8 thread ranks against the oracle
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zc}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python tests/fuzz_ranks.py ${2:-60} ${3:-0} > $O/fuzz.jsonl 2> $O/fuzz.err; rc=$?
grep -c '"ok": true' $O/fuzz.jsonl; grep '"ok": false' $O/fuzz.jsonl | cut -c1-900; tail -1 $O/fuzz.jsonl; tail -5 $O/fuzz.err
exit $rc
