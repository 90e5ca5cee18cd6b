#!/bin/bash
# round 4 evidence: the whole GPU suite, the profile run of the headline (bench line with its own counter passes, kernel trace, FETCH /
# WRITE passes), and series B (the reference's CPU-build defaults on row blocks) with its own kernel trace
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
T=${1:-r04a}
mkdir -p $R/gpurun_out/${T}_suite
cd $R
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $R/gpurun_out/${T}_suite/t_all.log 2>&1 || { tail -60 $R/gpurun_out/${T}_suite/t_all.log; exit 1; }
tail -5 $R/gpurun_out/${T}_suite/t_all.log
bash tools/gpurun/r03_k.sh $T || exit 1
O=$R/gpurun_out/$T
for g in 64 96 128 256; do timeout -k 10 600 python tools/series_b.py --grid $g --steps 3 >> $O/series_b.jsonl 2>> $O/series_b.err || { tail -5 $O/series_b.err; exit 1; }; done
for cfg in "64 4" "128 8" "256 8"; do set -- $cfg; timeout -k 10 600 python tools/series_b.py --grid $1 --rank-grid $2 --steps 3 >> $O/series_b_rank_blocks.jsonl 2>> $O/series_b.err || { tail -5 $O/series_b.err; exit 1; }; done
cd /tmp && export TMPDIR=/tmp
HDA_VERBOSE=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_b -o run -- python3 $R/tools/series_b.py --grid 128 --steps 3 > $O/trace_b.log 2>&1 || { tail -20 $O/trace_b.log; exit 1; }
cd $R
python3 tools/trace_by_operator.py $(find $O/trace_b -name "*kernel_trace.csv" | head -1) $O/series_b_kernel_by_operator.csv
find $O/trace_b -name "*kernel_trace.csv" -size +40M -delete
cut -c1-400 $O/series_b.jsonl
