#!/bin/bash
# round 3: what bounds the windowed kernel (dominant: level-1 Jacobi sweep) -- one small counter group per pass over the 256^3 bench
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r03pmc}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_BUSY_avr MemUnitStalled" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o run -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-table --no-plain-csr > $O/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $O/p$i.log; }
  echo "pass $i done: $grp"
done
python3 - <<PY
import csv, glob, statistics, collections, os
O = "$O"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = [k for k in agg if any(s in k for s in ("k_spmv_win<2", "k_spmv_win<1", "k_spmv_win<0", "k_spmv_stream<0, false, true", "k_spmv_rowclass<0, true", "k_cg_update", "k_cg_dir"))]
with open(O + "/summary.csv", "w") as o:
    o.write("kernel,counter,calls,max,median_of_top_cluster\n")
    for k in sorted(keep):
        for c, v in sorted(agg[k].items()):
            m = max(v); top = [x for x in v if x >= 0.7 * m] if m > 0 else v
            o.write(f'"{k}",{c},{len(v)},{m:.6g},{statistics.median(top):.6g}\n')
print(open(O + "/summary.csv").read())
PY
find $O -name "*counter_collection.csv" -size +20M -delete
