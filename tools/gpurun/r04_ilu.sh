#!/bin/bash
# round 4: block-Jacobi ILU(0) with exact substitutions on row blocks: parity tests, then one block against the automatic blocks
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04ilu
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "ilu" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
for g in 64 128; do
  for b in 1 auto; do
    for sm in "" "--smoother"; do
      if [ $b = auto ]; then bb=""; else bb="--blocks $b"; fi
      HDA_VERBOSE=1 timeout -k 10 400 python tools/ilu_blocks.py --grid $g $bb $sm >> $O/ilu.jsonl 2>> $O/ilu.err || { tail -20 $O/ilu.err; exit 1; }
    done
  done
done
HDA_VERBOSE=1 timeout -k 10 400 python tools/ilu_blocks.py --grid 256 >> $O/ilu.jsonl 2>> $O/ilu.err || { tail -20 $O/ilu.err; exit 1; }
timeout -k 10 400 python tools/ilu_blocks.py --grid 256 --smoother >> $O/ilu.jsonl 2>> $O/ilu.err || { tail -20 $O/ilu.err; exit 1; }
timeout -k 10 400 python tools/ilu_blocks.py --grid 256 --smoother --tri-solve 0 >> $O/ilu.jsonl 2>> $O/ilu.err || { tail -20 $O/ilu.err; exit 1; }
cut -c1-330 $O/ilu.jsonl
grep -c "block Gauss-Seidel plan" $O/ilu.err
