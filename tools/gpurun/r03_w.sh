#!/bin/bash
# round 3: the whole GPU suite on the final tree, then the default bench line (with its own counter passes for roofline.traffic)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03w}
rm -rf $O; mkdir -p $O
cd $R
( time timeout -k 10 600 python bench.py > $O/bench1.json 2> $O/bench1.err ) 2> $O/bench1.time || { tail -30 $O/bench1.err; exit 1; }
cat $O/bench1.time
python3 -c "
import json; d=json.load(open('$O/bench1.json'))
print('N=1', {k:d.get(k) for k in ('value','ms_per_step','iters','setup_ms','solve_timer_ms','iters_match')}, 'plain', d['plain_csr']['ms_per_step'])
print('roofline', d['roofline'])
print('l0 traffic', d['level0_spmv'].get('traffic'), d.get('level0_prolongation',{}).get('traffic'), d.get('level0_restriction',{}).get('traffic'))"
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -6 $O/t_all.log
