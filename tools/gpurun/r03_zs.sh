#!/bin/bash
# round 3: Galerkin products: larger chunks than the rows need (fewer chunks, longer sorts)?
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zs}
rm -rf $O; mkdir -p $O
cd $R
for round in 1 2; do
for cap in 2048 4096 8192; do
  HDA_VERBOSE=1 HDA_ESC_CAP=$cap timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_${cap}_$round.json 2> $O/b_${cap}_$round.err || { tail -30 $O/b_${cap}_$round.err; exit 1; }
done; done
python3 - <<PY
import json,glob,os,re
for f in sorted(glob.glob('$O/b_*.json')):
    d=json.load(open(f)); e=open(f.replace('.json','.err')).read()
    it=re.findall(r'setup level (\d+):.*?rap ([\d.]+)', e)
    print(os.path.basename(f), 'setup', round(d['setup_ms'],1), 'iters', d['iters'], 'rap:', [x[1] for x in it[-7:-3]])
PY
