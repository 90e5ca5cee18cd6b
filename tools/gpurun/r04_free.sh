#!/bin/bash
# round 4: the barrier-free block Gauss-Seidel kernel (default; HDA_GS_FREE=0 = the ring kernel): parity of the block sweeps on every
# kernel form, then series B with both
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04free
rm -rf $O; mkdir -p $O
cd $R
for f in "HDA_GS_FREE=1 HDA_GS_FREE_CHECK=1 HDA_GS_SORTED_MIN=0" "HDA_GS_FREE=0 HDA_GS_RING=0 HDA_GS_SORTED_MIN=0" "HDA_GS_FREE=0" "HDA_GS_FREE_CHECK=1"; do
  env $f timeout -k 10 400 python -m pytest tests/test_gpu_blocks.py -x -q -m gpu > $O/t.log 2>&1 || { echo "$f"; tail -40 $O/t.log; exit 1; }
  echo "$f: $(tail -1 $O/t.log)"
done
for g in ${GRIDS:-64 96 128 256}; do
  for f in 0 1; do
    HDA_GS_FREE=$f HDA_GS_FREE_CHECK=1 HDA_VERBOSE=1 timeout -k 10 500 python tools/series_b.py --grid $g --steps 3 >> $O/series_b_$f.jsonl 2>> $O/series_b_$f.err || { tail -20 $O/series_b_$f.err; exit 1; }
  done
done
for f in 0 1; do echo "HDA_GS_FREE=$f"; python3 - $O/series_b_$f.jsonl <<'PYEOF'
import json, sys
for l in open(sys.argv[1]):
    q = json.loads(l); print("  grid", q["grid"], "V", q["V"], "iters", q["iters"], "ms", round(q["ms_per_step"], 2), "setup", round(q["setup_ms"]), "final_rel %.3e" % q["final_rel"])
PYEOF
done
cd /tmp && export TMPDIR=/tmp
HDA_GS_FREE=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/tools/series_b.py --grid 128 --steps 3 > $O/trace.log 2>&1
cd $R && python3 tools/trace_by_operator.py $O/trace/run_kernel_trace.csv $O/by_op.csv && rm -f $O/trace/run_kernel_trace.csv && grep "gs_blocks" $O/by_op.csv | head -8; grep "barrier-free" $O/series_b_1.err | sort | uniq -c | cut -c1-200
