#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02x
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
for g in 0; do
HDA_INTERP_LANES=$g HDA_VERBOSE=1 timeout -k 10 300 python tools/gpurun/gpurun_setup.py 256 > $O/setup_$g.log 2> $O/setup_$g.err || { tail -30 $O/setup_$g.err; exit 1; }
echo "lanes $g"; cat $O/setup_$g.log
grep "setup level" $O/setup_$g.err | tail -7 | head -4 | cut -c1-200
grep "interp: build (" $O/setup_$g.err | tail -7 | head -4 | cut -c1-200
done
timeout -k 10 300 python tools/gpurun/gpurun_setup.py 256 > $O/setup_q.log 2> $O/setup_q.err || { tail -30 $O/setup_q.err; exit 1; }
cat $O/setup_q.log
