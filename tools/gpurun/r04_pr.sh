#!/bin/bash
# round 4: where the level-0 transfer operators' over-fetch comes from -- FETCH_SIZE of the P / R applies at 128^3, 192^3, 256^3 (a grid
# plane = 128 / 288 / 512 KB: five planes in flight against the 4-MB L2 of an XCD)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04pr
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in ${GRIDS:-128 192 256}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/g$g/$c -o run -- python3 $R/bench.py --grid $g --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-table --no-traffic --no-plain-csr --no-aggressive --no-cpu-defaults --no-side-configs > $O/g${g}_$c.log 2>&1 || { tail -20 $O/g${g}_$c.log; exit 1; }
  done
  python3 $R/tools/pmc_kernels.py $O/g$g "k_spmv_win<0, false, true" "k_spmv_rowclass" "k_cg_dir" > $O/kernels_$g.txt 2>&1 || true
  echo "== grid $g"; head -30 $O/kernels_$g.txt
  find $O -name "*counter_collection.csv" -size +30M -delete
done
