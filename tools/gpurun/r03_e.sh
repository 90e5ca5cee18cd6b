#!/bin/bash
# round 3: A/B of (a) finalize fusion levels, (b) the lane-group kernel on small operators (HDA_SMALL_NNZ), interleaved on one box
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03e}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fusions or amg_pcg_matches or vcycle or relax or spmv" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
run() { # tag grid env...
  tag=$1; n=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --grid $n --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/b_${n}_$tag.json 2> $O/b_${n}_$tag.err || { tail -30 $O/b_${n}_$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_${n}_$tag.json'))
print('grid $n $tag', round(d['ms_per_step'],4), round(d['solve_timer_ms'],4), 'seam', round(d['seam']['ms_per_step'],4), d['iters'])"
}
for rep in 1 2; do
for n in 64 128 256; do
run fin1_$rep $n HDA_FUSE_FINALIZE=1
run fin0_$rep $n HDA_FUSE_FINALIZE=0
run fin2_$rep $n HDA_FUSE_FINALIZE=2
run sm100k_$rep $n HDA_SMALL_NNZ=100000
run sm1m_$rep $n HDA_SMALL_NNZ=1000000
run sm4m_$rep $n HDA_SMALL_NNZ=4000000
done; done
