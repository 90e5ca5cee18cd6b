#!/bin/bash
# round 3: the widened rows once more on the final tree -- config 5 stand-in (aniso), Darcy MGR driver, 384^3 / 512^3 on one GPU
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03s}
rm -rf $O; mkdir -p $O
cd $R
for n in 128 160; do
timeout -k 10 300 python bench.py --workload aniso --grid $n --steps 5 --warmup 1 > $O/aniso_$n.json 2> $O/aniso_$n.err || { tail -20 $O/aniso_$n.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/aniso_$n.json'))
print('aniso $n', {k:d.get(k) for k in ('value','ms_per_step','iters','setup_ms','converged')}, 'level0 spmv frac', round(d['level0_spmv']['csr_equiv_frac'],3), 'dom frac', round(d['roofline']['frac'],3))"
done
for n in 384 512; do
timeout -k 10 600 python bench.py --grid $n --steps 3 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/lap_$n.json 2> $O/lap_$n.err || { tail -20 $O/lap_$n.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/lap_$n.json'))
print('lap7 $n', {k:d.get(k) for k in ('value','ms_per_step','iters','setup_ms','hbm_in_use_gb','hbm_peak_gb')}, 'dom frac', round(d['roofline']['frac'],3), 'agg1', {k:d['aggressive_1'][k] for k in ('ms_per_step','iters','setup_ms')})"
done
[ -x oracle/_ref/darcy_ref ] && for g in "64 64 32" "160 160 80"; do ./oracle/_ref/darcy_ref -v 1 -n $g 2>&1 | grep -i "iter\|error\|time\|setup\|solve" | head -12; done
