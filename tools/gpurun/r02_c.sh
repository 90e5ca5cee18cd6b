#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02c
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "ex8 or parity_at_128 or fused or overlapped" > $O/t_new.log 2>&1 || { tail -40 $O/t_new.log; exit 1; }
tail -3 $O/t_new.log
for v in "1 1" "1 0" "0 1" "0 0"; do set -- $v
HDA_OVERLAP=$1 HDA_FUSE_DOTS=$2 timeout -k 10 300 python bench.py --gpus 4 --grid 128 --steps 3 --warmup 1 > $O/bench4_$1$2.json 2> $O/bench4_$1$2.err || { tail -30 $O/bench4_$1$2.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench4_$1$2.json'))
print('overlap/fuse $1$2', {k:d[k] for k in ('value','ms_per_step','solve_timer_ms','iters','allreduces_per_iter','halo_exchanges_per_iter','halo_exchanges_overlapped_per_iter')})"
done
