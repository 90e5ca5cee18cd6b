#!/bin/bash
# round 3: single-reduction PCG tests, the whole GPU suite, the default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03j}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "single_reduction" > $O/t_sr.log 2>&1 || { tail -60 $O/t_sr.log; exit 1; }
tail -2 $O/t_sr.log
( time timeout -k 10 1100 python -m pytest tests -x -q -m gpu ) > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -6 $O/t_all.log
timeout -k 10 600 python bench.py > $O/bench1.json 2> $O/bench1.err || { tail -30 $O/bench1.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench1.json'))
print('N=1', {k:d.get(k) for k in ('value','ms_per_step','iters','setup_ms','solve_timer_ms','iters_match')}, 'frac', d['roofline']['frac'], 'plain', d['plain_csr']['ms_per_step'])
print('aggressive_1', d.get('aggressive_1'))
print('cpu', d.get('cpu_baseline'))"
