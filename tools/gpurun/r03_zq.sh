#!/bin/bash
# round 3: interpolation, with LDS-only fences: next neighbour's entries requested ahead (HDA_INTERP_REG=3) against not (1)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zq}
rm -rf $O; mkdir -p $O
cd $R
HDA_INTERP_REG=3 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "interp or hierarch or long_row or parity_at or fuzz or edge" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
for round in 1 2 3; do
for reg in 1 3; do
  HDA_VERBOSE=1 HDA_INTERP_REG=$reg timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_${reg}_$round.json 2> $O/b_${reg}_$round.err || { tail -30 $O/b_${reg}_$round.err; exit 1; }
done; done
python3 - <<PY
import json,glob,os,re
for f in sorted(glob.glob('$O/b_*.json')):
    d=json.load(open(f)); e=open(f.replace('.json','.err')).read()
    it=re.findall(r'setup level (\d+):.*?interp ([\d.]+)', e)
    print(os.path.basename(f), 'setup', round(d['setup_ms'],1), 'iters', d['iters'], 'interp:', [x[1] for x in it[-7:-3]])
PY
