#!/bin/bash
# round 5: bytes-vs-time accounting of the AMG setup at 256^3 (tools/setup_accounting.py): kernel trace + FETCH_SIZE / WRITE_SIZE passes
# over tools/setup_only.py (counters in their own runs, no tracing domains with --pmc)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
T=${1:-r05setup}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout -k 10 200 python tools/setup_only.py 256 > $O/setup_only.json 2> $O/setup_only.err || { tail -5 $O/setup_only.err; exit 1; }
cat $O/setup_only.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 $R/tools/setup_only.py 256 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/tools/setup_only.py 256 > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/tools/setup_only.py 256 > $O/pmc_write.log 2>&1 || { tail -20 $O/pmc_write.log; exit 1; }
cd $R
python3 tools/setup_accounting.py $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/pmc_fetch $O/pmc_write $O/setup_only.json > $O/setup_accounting.md || exit 1
cat $O/setup_accounting.md
