#!/bin/bash
# round 2, first GPU call: the new dist tests, the unified bench at N = 1 and a 4-ranks-on-one-GPU rehearsal
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02a
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_hypredrv.py -x -q -m gpu -k "fused or overlapped or row_partitioned_solve or laplacian_driver_path" > $O/t_dist.log 2>&1 || { tail -40 $O/t_dist.log; exit 1; }
tail -3 $O/t_dist.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err || { tail -30 $O/bench1.err; exit 1; }
tail -c 3000 $O/bench1.json
timeout -k 10 300 python bench.py --gpus 4 --grid 128 --steps 3 --warmup 1 > $O/bench4.json 2> $O/bench4.err || { tail -30 $O/bench4.err; exit 1; }
tail -c 1500 $O/bench4.json
