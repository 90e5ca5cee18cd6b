#!/bin/bash
# round 3: fused tail of the V-cycle: bitwise test, then small-grid timings by threshold (same box, interleaved, two rounds)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zb}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fused_cycle_tail or bitwise or hierarchy_identical or amg_pcg_matches" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
for round in 1 2; do
for n in 64 128; do
for thr in 0 150 600 2500; do
  HDA_FUSE_TAIL=$thr timeout -k 10 300 python bench.py --grid $n --steps 20 --warmup 3 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_${n}_${thr}_$round.json 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
done; done; done
HDA_FUSE_TAIL=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_256_0_1.json 2>> $O/b.err || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_256_600_1.json 2>> $O/b.err || exit 1
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/b_*.json')):
    d=json.load(open(f)); print(os.path.basename(f), round(d['ms_per_step'],4), d['iters'], round(d['solve_timer_ms'],4))
PY
