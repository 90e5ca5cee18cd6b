#!/bin/bash
# round 3: PCG fusions + 256^3 oracle parity test, bench N=1, 2-rank bench with strong_<n> + n1_reference, 4-rank rehearsal lines
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03c}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "fusions or 128_and_256 or nested_pcg or bitwise or amg_pcg" > $O/t_new.log 2>&1 || { tail -60 $O/t_new.log; exit 1; }
tail -3 $O/t_new.log
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err || { tail -30 $O/bench1.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench1.json'))
print('N=1', {k:d.get(k) for k in ('value','ms_per_step','iters','setup_ms','solve_timer_ms')}, d['roofline']['frac'], d['plain_csr']['ms_per_step'])"
timeout -k 10 900 python bench.py --gpus 2 --grid 128 --steps 3 --warmup 1 > $O/bench2.json 2> $O/bench2.err || { tail -30 $O/bench2.err; cat $O/bench2.json; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench2.json'))
print({k:d.get(k) for k in ('value','ms_per_step','iters','allreduces_per_iter','halo_exchanges_per_iter','transport','ranks_seen','partitioned_levels','levels_total','speedup_weak_dofs','speedup_strong','extras_error')})
print('strong', d.get('strong_128')); print('n1', d.get('n1_reference'))"
for pg in 1 0; do
HDA_GHOST_PROLONG=$pg timeout -k 10 600 python bench.py --gpus 4 --grid 128 --steps 3 --warmup 1 --no-extras > $O/bench4_pg$pg.json 2> $O/bench4_pg$pg.err || { tail -30 $O/bench4_pg$pg.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench4_pg$pg.json'))
print('4 ranks ghost prolong $pg', {k:d.get(k) for k in ('value','ms_per_step','iters','allreduces_per_iter','halo_exchanges_per_iter','partitioned_levels','levels_total')})"
done
