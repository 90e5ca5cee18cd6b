#!/bin/bash
# round 5 evidence: the whole GPU suite (incl. the MPI launches of the reference's unmodified multi-rank callers), the driver's own
# default bench command (the line with every extra: cpu_baseline, traffic, plain / uncoded, cpu_defaults, aggressive, gmres_amg_ilu0,
# gmres_mgr), the profile run (kernel-trace stats + FETCH / WRITE passes), and the setup accounting
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
T=${1:-r05f}
mkdir -p $R/gpurun_out/${T}_suite
cd $R
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu ) > $R/gpurun_out/${T}_suite/t_all.log 2>&1 || { tail -60 $R/gpurun_out/${T}_suite/t_all.log; exit 1; }
tail -5 $R/gpurun_out/${T}_suite/t_all.log
bash tools/gpurun/gpurun_profile.sh $T || exit 1
bash tools/gpurun/r05_setup.sh ${T}_setup > $R/gpurun_out/$T/setup.log 2>&1 || { tail -20 $R/gpurun_out/$T/setup.log; exit 1; }
cp $R/gpurun_out/${T}_setup/setup_accounting.md $R/gpurun_out/$T/setup_accounting.md
cp $R/gpurun_out/${T}_setup/setup_only.json $R/gpurun_out/$T/setup_only.json
rm -rf $R/gpurun_out/${T}_setup/trace $R/gpurun_out/${T}_setup/pmc_fetch $R/gpurun_out/${T}_setup/pmc_write
cat $R/gpurun_out/$T/setup_accounting.md | cut -c1-240
