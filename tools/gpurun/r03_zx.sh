#!/bin/bash
# round 3: what a first multi-GPU run prints when RCCL cannot be joined (here: two ranks forced onto RCCL with one GPU)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zx}
rm -rf $O; mkdir -p $O
cd $R
HDA_TRANSPORT=rccl timeout -k 10 300 python bench.py --gpus 2 --grid 64 --steps 2 --warmup 1 > $O/out.txt 2> $O/err.txt; echo "exit code $?"
cut -c1-400 $O/out.txt
HDA_BENCH_NO_FALLBACK=1 HDA_TRANSPORT=rccl timeout -k 10 300 python bench.py --gpus 2 --grid 64 --steps 2 --warmup 1 > $O/out2.txt 2> $O/err2.txt; echo "no-fallback exit code $?"
cut -c1-300 $O/out2.txt
timeout -k 10 300 python bench.py --gpus 2 --grid 64 --steps 2 --warmup 1 --no-extras > $O/out3.txt 2> $O/err3.txt; echo "normal exit code $?"
python3 -c "
import json
for l in open('$O/out3.txt'):
    if l.startswith('{'): d=json.loads(l); print(d.get('transport'), d.get('rccl_error'), d.get('ms_per_step'))"
