#!/bin/bash
# round 5: XCD-aware dealing of the interpolation kernel's rows, same box A/B: setup time and the kernel's duration / fetch
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r05interp}
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "hierarchy or interp" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for x in 0 1 0 1; do
  HDA_INTERP_XCD=$x timeout -k 10 200 python tools/setup_only.py 256 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HDA_INTERP_XCD=$x setup ms', [round(v,1) for v in d['setup_ms']])"
done
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  HDA_INTERP_XCD=$x timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$x -o run -- python3 $R/tools/setup_only.py 256 > $O/trace$x.log 2>&1 || { tail -20 $O/trace$x.log; exit 1; }
  grep "k_interp_wave\|k_spgemm_esc" $O/trace$x/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,400-
  HDA_INTERP_XCD=$x timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc$x -o run -- python3 $R/tools/setup_only.py 256 > $O/pmc$x.log 2>&1 || { tail -20 $O/pmc$x.log; exit 1; }
  python3 $R/tools/pmc_kernels.py $O/pmc$x k_interp_wave | cut -c1-40,300-
  rm -rf $O/trace$x/*kernel_trace.csv
done
