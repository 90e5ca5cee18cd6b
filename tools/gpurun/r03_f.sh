#!/bin/bash
# round 3: full suite with the small-operator dispatch; A/B of the row-operand prefetch in the windowed kernel (HDA_WIN_PF)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03f}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1 || { tail -60 $O/t_all.log; exit 1; }
tail -2 $O/t_all.log
HDA_WIN_PF=1 timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "windowed or amg_pcg_matches or vcycle or parity_at or relax_jacobi or bitwise" > $O/t_pf.log 2>&1 || { tail -60 $O/t_pf.log; exit 1; }
tail -2 $O/t_pf.log
run() { tag=$1; n=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --grid $n --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/b_${n}_$tag.json 2> $O/b_${n}_$tag.err || { tail -30 $O/b_${n}_$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_${n}_$tag.json'))
print('grid $n $tag', round(d['ms_per_step'],4), round(d['solve_timer_ms'],4), 'seam', round(d['seam']['ms_per_step'],4), d['iters'], 'dom', round(d['roofline']['avg_ms'],4), round(d['roofline']['frac'],4))"
}
for rep in 1 2 3; do
for n in 256 128; do
run pf0_$rep $n HDA_WIN_PF=0
run pf1_$rep $n HDA_WIN_PF=1
done; done
