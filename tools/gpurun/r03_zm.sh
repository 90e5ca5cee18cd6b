#!/bin/bash
# round 3: allocator bound at twice the peak: the warm setup must be as fast as with no bound at all (HDA_POOL_CACHE_MIN_GB=200)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zm}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "allocator_cache or coded_operators_change" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
for round in 1 2 3; do
for gb in 4 200; do
  HDA_POOL_CACHE_MIN_GB=$gb timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_${gb}_$round.json 2> $O/b_${gb}_$round.err || { tail -30 $O/b_${gb}_$round.err; exit 1; }
done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/b_*.json')):
    d=json.load(open(f)); print(os.path.basename(f), 'setup', round(d['setup_ms'],1), 'cold', round(d['setup_cold_ms'],1), 'ms/solve', round(d['ms_per_step'],3), 'hbm', round(d['hbm_in_use_gb'],2), round(d['hbm_peak_gb'],2))
PY
