#!/bin/bash
# round 3: compacted escapes of the value-coded transfer operators -- tests, then same-box A/B (HDA_ESC_COMPACT)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03g}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "coded or parity_at or amg_pcg_matches or bitwise or full_size or windowed or vcycle" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
run() { tag=$1; n=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --grid $n --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/b_${n}_$tag.json 2> $O/b_${n}_$tag.err || { tail -30 $O/b_${n}_$tag.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/b_${n}_$tag.json'))
print('grid $n $tag', round(d['ms_per_step'],4), round(d['solve_timer_ms'],4), 'iters', d['iters'], 'P0', round(d['level0_prolongation']['avg_ms'],4), 'R0', round(d['level0_restriction']['avg_ms'],4), 'dom', round(d['roofline']['avg_ms'],4))"
}
for rep in 1 2 3; do
run esc0_$rep 256 HDA_ESC_COMPACT=0
run esc1_$rep 256 HDA_ESC_COMPACT=1
done
