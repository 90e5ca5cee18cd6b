#!/bin/bash
# round 3: run form of the windowed CSR on the general-matrix path (HDA_CODED=0): tests, then same-box A/B at 256^3 and on the anisotropic workload
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03run}
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "run_form or coded_operators or window or spe10 or row_partitioned_irregular or eight_ranks" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
for round in 1 2 3; do
for runs in 0 1; do
  HDA_CODED=0 HDA_WINDOW_RUNS=$runs timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_${runs}_$round.json 2> $O/b_${runs}_$round.err || { tail -30 $O/b_${runs}_$round.err; exit 1; }
  HDA_WINDOW_RUNS=$runs timeout -k 10 300 python bench.py --workload aniso --grid 128 --steps 10 --warmup 2 > $O/a_${runs}_$round.json 2> $O/a_${runs}_$round.err || { tail -30 $O/a_${runs}_$round.err; exit 1; }
done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/[ab]_*.json')):
    d=json.load(open(f)); k=d['level0_spmv']
    print(os.path.basename(f), 'ms/solve', round(d['ms_per_step'],3), 'iters', d['iters'], 'setup', round(d['setup_ms'],1), 'level0', k['kernel'], round(k['avg_ms'],4), 'csr frac', round(k['csr_equiv_frac'],3))
PY
