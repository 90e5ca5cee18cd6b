#!/bin/bash
# the GPU suite and the benches with guard tails on every device block and poisoned allocations
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03guard}
rm -rf $O; mkdir -p $O
cd $R
export HDA_GUARD=1 HDA_POISON=1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "not config3_full and not 128_and_256" > $O/t_all.log 2>&1 || { tail -40 $O/t_all.log; exit 1; }
tail -2 $O/t_all.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-csr > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('guarded bench', d['ms_per_step'], d['iters'], d['converged'])"
timeout -k 10 300 python bench.py --gpus 3 --grid 96 --steps 2 --warmup 1 --no-extras > $O/bench3.json 2> $O/bench3.err || { tail -30 $O/bench3.err; exit 1; }
HDA_OVERLAP=1 timeout -k 10 300 python bench.py --gpus 3 --grid 96 --steps 2 --warmup 1 --no-extras > $O/bench3o.json 2> $O/bench3o.err || { tail -30 $O/bench3o.err; exit 1; }
python3 -c "
import json
for f in ('bench3','bench3o'):
    d=json.load(open('$O/'+f+'.json')); print('guarded', f, d['ms_per_step'], d['iters'], d['converged'], d['ranks_seen'], d['halo_exchanges_overlapped_per_iter'])"
timeout -k 10 300 python bench.py --workload aniso --grid 96 --steps 2 --warmup 1 > $O/aniso.json 2> $O/aniso.err || { tail -30 $O/aniso.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/aniso.json')); print('guarded aniso', d['ms_per_step'], d['iters'], d['converged'])"
