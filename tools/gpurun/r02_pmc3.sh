#!/bin/bash
# PMC passes over one 256^3 AMG setup: where the SpGEMM and interpolation kernels spend their cycles
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02pmc3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "TA_BUSY_avr MemUnitStalled" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o run -- python3 $R/tools/gpurun/gpurun_setup.py 256 1 > $O/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -5 $O/p$i.log; }
  echo "pass $i done: $grp"
done
python3 - <<PY
import csv, glob, collections
O = "$O"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = [k for k in agg if any(s in k for s in ("k_spgemm_esc", "k_interp_wave", "k_interp_build", "k_strength", "k_pmis_setF", "k_vhist"))]
with open(O + "/summary.csv", "w") as o:
    o.write("kernel,counter,calls,first4\n")
    for k in sorted(keep):
        for c, v in sorted(agg[k].items()):
            o.write(f'"{k}",{c},{len(v)},' + " ".join(f"{x:.4g}" for x in v[:4]) + "\n")
print(open(O + "/summary.csv").read())
PY
find $O -name "*counter_collection.csv" -size +20M -delete
