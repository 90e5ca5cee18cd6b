#!/bin/bash
# round 4, VERDICT #7: the row-class kernel with an LDS window of x (HDA_ROWCLASS_LDS=1) against the committed one: parity first, then
# the headline bench twice per form (kernel table on: the three level-0 row-class kernels are what the experiment is about)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04rc
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_class" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
for rep in 1 2; do
  for v in 0 1; do
    HDA_ROWCLASS_LDS=$v timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-traffic --no-plain-csr --no-cpu-defaults --no-aggressive \
        > $O/bench_$v.$rep.json 2> $O/bench_$v.$rep.err || { tail -30 $O/bench_$v.$rep.err; exit 1; }
    python3 - $O/bench_$v.$rep.json $v <<'EOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print("LDS", sys.argv[2], "ms_per_step", round(d["ms_per_step"], 3), "iters", d["iters"], "level0_spmv", d["level0_spmv"]["avg_ms"], "roofline", d["roofline"]["avg_ms"])
print("    level 0 alone:", {k: round(v["ms"], 4) for k, v in d["kernels"].items()}, "seam", d["seam"]["ms_per_step"])
EOF
  done
done
