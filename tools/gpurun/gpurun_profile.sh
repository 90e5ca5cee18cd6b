#!/bin/bash
# evidence run (every round): bench line, kernel-trace stats, two PMC passes (separate, as the guide prescribes)
set -o pipefail
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $R && python bench.py --steps 5 --warmup 1 ${BENCH_EXTRA} > $O/bench_line.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-kernel-table --no-plain-csr --no-aggressive --no-traffic --no-cpu-defaults --no-side-configs > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-table --no-plain-csr --no-aggressive --no-traffic --no-cpu-defaults --no-side-configs > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-table --no-plain-csr --no-aggressive --no-traffic --no-cpu-defaults --no-side-configs > $O/pmc_write.log 2>&1 || { tail -20 $O/pmc_write.log; exit 1; }
# keep the merge-back small: the per-dispatch traces are big, the stats are what is judged
find $O/trace -name "*kernel_trace.csv" -size +40M -delete
ls -R $O | head -30
tail -c 2500 $O/bench_line.json
