#!/bin/bash
# round 5: hierarchy sizes of the benchmark, a headline-only bench line, and a per-dispatch kernel trace of the same run
# (profiles/<tag>_kernel_by_operator.csv + the dispatch sequence of one PCG iteration: tools/iteration_sequence.py)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
T=${1:-r05a}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout -k 10 300 python tools/hierarchy_dims.py 256 > $O/dims.txt 2>&1 || { tail -5 $O/dims.txt; exit 1; }
cat $O/dims.txt
FLAGS="--no-cpu-baseline --no-kernel-table --no-plain-csr --no-aggressive --no-traffic --no-cpu-defaults --no-side-configs"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 $FLAGS > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-700 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 $FLAGS > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
cd $R
K=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_by_operator.py $K $O/kernel_by_operator.csv
python3 tools/iteration_sequence.py $K > $O/iteration_sequence.txt
tail -50 $O/iteration_sequence.txt
find $O/trace -name "*kernel_trace.csv" -size +40M -delete
