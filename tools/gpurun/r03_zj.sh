#!/bin/bash
# round 3: interpolation kernel with queued candidate rows and the next neighbour's entries requested ahead: tests, then timings
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zj}
rm -rf $O; mkdir -p $O
cd $R
echo skip tests

for round in 1 2; do
for cfg in "3 64 8" "5 64 8"; do
  set -- $cfg
  HDA_VERBOSE=1 HDA_INTERP_LB=$1 HDA_INTERP_NBR=$2 HDA_INTERP_NBR_FORCE=$3 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_$1_$2_$3_$round.json 2> $O/b_$1_$2_$3_$round.err || { tail -30 $O/b_$1_$2_$3_$round.err; exit 1; }
done; done
python3 - <<PY
import json,glob,os,re
for f in sorted(glob.glob('$O/b_*.json')):
    d=json.load(open(f)); e=open(f.replace('.json','.err')).read()
    it=re.findall(r'setup level (\d+):.*?interp ([\d.]+)', e)
    print(os.path.basename(f), 'setup', round(d['setup_ms'],1), 'iters', d['iters'], 'interp:', [x[1] for x in it[-7:-3]])
PY
