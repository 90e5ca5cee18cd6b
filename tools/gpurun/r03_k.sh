#!/bin/bash
# round 3 evidence: profile run (bench line, kernel trace stats, FETCH / WRITE passes) + per-level setup trace
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
bash tools/gpurun/gpurun_profile.sh ${1:-r03a} || exit 1
O=$R/gpurun_out/${1:-r03a}
HDA_VERBOSE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-table --no-plain-csr --no-aggressive --no-traffic --no-cpu-defaults --no-side-configs > $O/verbose.json 2> $O/verbose.err || { tail -20 $O/verbose.err; exit 1; }
grep "setup level\|renumbering\|windowed CSR for\|value-coded\|row-class\|stencil" $O/verbose.err | tail -60
