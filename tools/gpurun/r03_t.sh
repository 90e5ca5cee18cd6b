#!/bin/bash
# round 3: final-tree check of the N > 1 line (2 ranks, extras) and of the PCG fusion tests
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03t}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fusions or nested_pcg or single_reduction or amg_pcg_matches or gmres" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
timeout -k 10 900 python bench.py --gpus 2 --grid 128 --steps 3 --warmup 1 > $O/bench2.json 2> $O/bench2.err || { tail -30 $O/bench2.err; cat $O/bench2.json; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench2.json'))
print({k:d.get(k) for k in ('value','ms_per_step','iters','allreduces_per_iter','halo_exchanges_per_iter','transport','ranks_seen','partitioned_levels','levels_total','speedup_weak_dofs','speedup_strong','extras_error')})
print('strong', d.get('strong_128')); print('n1', d.get('n1_reference'))"
