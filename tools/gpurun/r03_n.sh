#!/bin/bash
# round 3: windowed kernel with the x gather one chunk ahead (HDA_WIN_PF=3) -- tests, then same-box A/B against HDA_WIN_PF=1
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03n}
rm -rf $O; mkdir -p $O
cd $R
HDA_WIN_PF=3 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "windowed or amg_pcg_matches or vcycle or parity_at or relax_jacobi or bitwise or overlapped or eight_ranks" > $O/t_pf3.log 2>&1 || { tail -60 $O/t_pf3.log; exit 1; }
tail -2 $O/t_pf3.log
run() { tag=$1; n=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --grid $n --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive > $O/b_${n}_$tag.json 2> $O/b_${n}_$tag.err || { tail -30 $O/b_${n}_$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_${n}_$tag.json'))
print('grid $n $tag', round(d['ms_per_step'],4), round(d['solve_timer_ms'],4), 'seam', round(d['seam']['ms_per_step'],4), d['iters'], 'dom', round(d['roofline']['avg_ms'],4), round(d['roofline']['frac'],4))"
}
for rep in 1 2 3; do
for n in 256 128; do
run pf1_$rep $n HDA_WIN_PF=1
run pf3_$rep $n HDA_WIN_PF=3
done; done
