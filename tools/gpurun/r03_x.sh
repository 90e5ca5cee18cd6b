#!/bin/bash
# round 3: allocator cache bound -- its test, same-box A/B of the default bench (cache must not cost the warm setup), then the whole suite
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03x}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "allocator_cache or coded_operators_change" > $O/t_pool.log 2>&1 || { tail -60 $O/t_pool.log; exit 1; }
tail -2 $O/t_pool.log
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err || { tail -30 $O/bench1.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench1.json'))
print('N=1', {k:d.get(k) for k in ('value','ms_per_step','iters','setup_ms','setup_cold_ms','solve_timer_ms','hbm_in_use_gb','hbm_peak_gb')}, 'plain', d['plain_csr']['ms_per_step'])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -6 $O/t_all.log
