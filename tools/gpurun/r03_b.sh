#!/bin/bash
# round 3: ghost-row prolongation (3 exchanges per level), multi-rank tests, 4-rank bench rehearsal with strong_<n> + n1_reference
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03b}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "prolongation_updates or eight_ranks or row_partitioned or overlapped or fused_dot" > $O/t_dist.log 2>&1 || { tail -60 $O/t_dist.log; exit 1; }
tail -3 $O/t_dist.log
timeout -k 10 900 python bench.py --gpus 4 --grid 128 --steps 3 --warmup 1 > $O/bench4.json 2> $O/bench4.err || { tail -30 $O/bench4.err; cat $O/bench4.json; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench4.json'))
print({k:d.get(k) for k in ('value','ms_per_step','iters','allreduces_per_iter','halo_exchanges_per_iter','halo_exchanges_overlapped_per_iter','transport','ranks_seen','partitioned_levels','levels_total','speedup_weak_dofs','speedup_strong')})
print('strong', d.get('strong_128')); print('n1', d.get('n1_reference'))"
HDA_GHOST_PROLONG=0 timeout -k 10 600 python bench.py --gpus 4 --grid 128 --steps 3 --warmup 1 --no-extras > $O/bench4_nopg.json 2> $O/bench4_nopg.err || { tail -30 $O/bench4_nopg.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench4_nopg.json'))
print('no ghost prolong', {k:d.get(k) for k in ('value','ms_per_step','iters','allreduces_per_iter','halo_exchanges_per_iter','partitioned_levels')})"
