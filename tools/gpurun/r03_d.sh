#!/bin/bash
# round 3: same-box A/B of the PCG launch fusions at 256^3 / 128^3 / 64^3, kernel traces of the small grids (launch structure)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03d}
rm -rf $O; mkdir -p $O
cd $R
for n in 256 128 64; do
for v in "1 1" "0 1" "1 0" "0 0" "1 1"; do set -- $v
HDA_FUSE_FINALIZE=$1 HDA_FUSE_Z0=$2 timeout -k 10 300 python bench.py --grid $n --steps 10 --warmup 2 --no-cpu-baseline --no-plain-csr --no-kernel-table > $O/b_${n}_$1$2.json 2> $O/b_${n}_$1$2.err || { tail -30 $O/b_${n}_$1$2.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/b_${n}_$1$2.json'))
print('grid $n fin=$1 z0=$2', {k:round(d[k],4) if isinstance(d[k],float) else d[k] for k in ('ms_per_step','solve_timer_ms','iters','setup_ms')}, 'seam', round(d['seam']['ms_per_step'],4))"
done; done
cd /tmp && export TMPDIR=/tmp
for n in 64 128; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$n -o run -- python3 $R/bench.py --grid $n --steps 5 --warmup 1 --no-cpu-baseline --no-kernel-table --no-plain-csr > $O/trace$n.log 2>&1 || { tail -20 $O/trace$n.log; exit 1; }
done
find $O -name "*kernel_trace.csv" -size +40M -delete
ls -R $O | head -40
