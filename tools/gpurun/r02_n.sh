#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02n
rm -rf $O; mkdir -p $O
cd $R
HYPREDRV_AMD_DEFAULTS=cpu hypredrive_amd/bin/hypredrive-cli examples/ex1.yml > $O/ex1.out 2>&1
python tools/compare_output.py $O/ex1.out tests/golden/refOutput/ex1.txt
timeout -k 10 300 python -m pytest tests/test_gpu_hypredrv.py -x -q -m gpu -k "golden_output or statistics" 2>&1 | tail -3
HYPREDRV_AMD_DEFAULTS=cpu oracle/_ref/laplacian_ref > $O/lap.out 2>&1; python tools/compare_output.py $O/lap.out tests/golden/refOutput/laplacian.txt | head -40
