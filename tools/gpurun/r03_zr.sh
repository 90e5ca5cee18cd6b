#!/bin/bash
# round 3: Galerkin products: 1024-product chunks (eight workgroups per CU) where rows are short: bit-exactness tests, then A/B
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03zr}
rm -rf $O; mkdir -p $O
cd $R
HDA_ESC_THREADS=4 timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "rap or spgemm or hierarch or parity_at_128 or fuzz" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
for round in 1 2 3; do
for sm in 4 0; do
  HDA_VERBOSE=1 HDA_ESC_THREADS=$sm timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic > $O/b_${sm}_$round.json 2> $O/b_${sm}_$round.err || { tail -30 $O/b_${sm}_$round.err; exit 1; }
done; done
python3 - <<PY
import json,glob,os,re
for f in sorted(glob.glob('$O/b_*.json')):
    d=json.load(open(f)); e=open(f.replace('.json','.err')).read()
    it=re.findall(r'setup level (\d+):.*?rap ([\d.]+)', e)
    print(os.path.basename(f), 'setup', round(d['setup_ms'],1), 'iters', d['iters'], 'rap:', [x[1] for x in it[-7:-3]])
PY
for n in 64 128; do for sm in 4 0; do
  HDA_ESC_THREADS=$sm timeout -k 10 300 python bench.py --grid $n --steps 3 --warmup 1 --no-cpu-baseline --no-plain-csr --no-kernel-table --no-aggressive --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, 'threads $sm', 'setup', round(d['setup_ms'],2), 'iters', d['iters'])"
done; done
