#!/bin/bash
# round 3: with unstaged neighbour rows on registers, is a smaller staging area (more rows in flight per CU) better on level 1?
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zh}
rm -rf $O; mkdir -p $O
cd $R
for round in 1 2; do
for nbr in 0 64 128 256 1024; do
  HDA_VERBOSE=1 HDA_INTERP_NBR=$nbr timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_${nbr}_$round.json 2> $O/b_${nbr}_$round.err || { tail -30 $O/b_${nbr}_$round.err; exit 1; }
done; done
python3 - <<PY
import json,glob,os,re
for f in sorted(glob.glob('$O/b_*.json')):
    d=json.load(open(f)); e=open(f.replace('.json','.err')).read()
    it=re.findall(r'setup level (\d+):.*?interp ([\d.]+)', e)
    caps=re.findall(r'lanes/row=(\d+) caps (\d+) (\d+) (\d+)', e)
    print(os.path.basename(f), 'setup', round(d['setup_ms'],1), 'iters', d['iters'], 'interp:', [x[1] for x in it[-7:-3]], 'caps', caps[-7:-4])
PY
