#!/bin/bash
# full GPU test suite + 4-rank rehearsal lines (staged transport, one GPU)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r02full}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1 || { tail -40 $O/t_all.log; exit 1; }
tail -2 $O/t_all.log
for v in "1 1" "0 0"; do set -- $v
HDA_OVERLAP=$1 HDA_FUSE_DOTS=$2 timeout -k 10 300 python bench.py --gpus 4 --grid 128 --steps 3 --warmup 1 > $O/bench4_$1$2.json 2> $O/bench4_$1$2.err || { tail -30 $O/bench4_$1$2.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench4_$1$2.json'))
print('overlap/fuse $1$2', {k:d[k] for k in ('value','ms_per_step','iters','allreduces_per_iter','halo_exchanges_per_iter','halo_exchanges_overlapped_per_iter','transport','ranks_seen')})"
done
timeout -k 10 300 python bench.py --gpus 4 --grid 128 --steps 3 --warmup 1 > $O/bench4_default.json 2> $O/bench4_default.err || { tail -30 $O/bench4_default.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench4_default.json'))
print('default', {k:d[k] for k in ('value','ms_per_step','iters','allreduces_per_iter','halo_exchanges_per_iter','halo_exchanges_overlapped_per_iter','transport','ranks_seen')})"
