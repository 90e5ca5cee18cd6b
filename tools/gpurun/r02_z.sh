#!/bin/bash
# full GPU test suite, then the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02z
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1 || { tail -40 $O/t_all.log; exit 1; }
tail -2 $O/t_all.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','iters','setup_ms','setup_cold_ms','iters_match')}, d['roofline'])"
