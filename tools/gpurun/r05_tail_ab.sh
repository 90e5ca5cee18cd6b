#!/bin/bash
# round 5: the one-launch tail of the cycle, same box: off / levels by HDA_TAIL_NNZ.  Headline ms per solve + the tail kernel's duration
# and the dispatch sequence of one iteration from a kernel trace.
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
T=${1:-r05tail}
O=$R/gpurun_out/$T
mkdir -p $O
FLAGS="--no-cpu-baseline --no-kernel-table --no-plain-csr --no-aggressive --no-traffic --no-cpu-defaults --no-side-configs"
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "tail or fusions" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for cfg in "0 60000" "1 60000" "1 100000" "1 1000000" "0 60000" "1 60000" "1 100000" "1 1000000"; do
  set -- $cfg
  HDA_TAIL=$1 HDA_TAIL_NNZ=$2 timeout -k 10 300 python bench.py --steps 10 --warmup 2 $FLAGS > $O/bench_$1_$2.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$O/bench_$1_$2.json')); print('HDA_TAIL=$1 HDA_TAIL_NNZ=$2', round(d['ms_per_step'],3), 'ms', d['iters'], 'iters')"
done
cd /tmp && export TMPDIR=/tmp
for cfg in "1 60000" "1 100000" "1 1000000"; do
  set -- $cfg
  HDA_TAIL=$1 HDA_TAIL_NNZ=$2 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$2 -o run -- python3 $R/bench.py --steps 2 --warmup 1 $FLAGS > $O/trace_$2.log 2>&1 || { tail -20 $O/trace_$2.log; exit 1; }
  K=$(find $O/trace_$2 -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/iteration_sequence.py $K > $O/iteration_sequence_$2.txt
  grep "k_cycle_tail\|^iteration" $O/iteration_sequence_$2.txt
  rm -f $K
done
