#!/bin/bash
# rehearsal of the partitioned AMG setup: several ranks on the one visible GPU (staged transport)
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD OMP_NUM_THREADS=1 HDA_DIST_SETUP=partitioned HDA_DIST_CHECK=1
run() { # world n solver rep_rows
   echo "=== world=$1 n=$2 $3 rep_rows=$4" | tee -a gpurun_out/dist.log
   HDA_REPLICATE_ROWS=$4 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$1 --master-addr 127.0.0.1 \
      --master-port $((29700 + $1 + $2)) tests/dist_worker.py solve gpurun_out/dist_$1_$2.json $2 $3 >> gpurun_out/dist.log 2>&1 || { echo FAILED | tee -a gpurun_out/dist.log; tail -40 gpurun_out/dist.log; return 1; }
   cat gpurun_out/dist_$1_$2.json | tee -a gpurun_out/dist.log; echo
}
rm -f gpurun_out/dist.log
run 2 16 pcg 100000 && run 2 16 pcg 0 && run 4 20 pcg 0 && run 4 24 pcg 700 && run 3 12 gmres 0 && run 4 48 pcg 2000
# timing view of the phases at a larger block (no self-check)
echo "=== timing world=2 n=128" | tee -a gpurun_out/dist.log
HDA_DIST_CHECK= HDA_VERBOSE=1 HDA_REPLICATE_ROWS=100000 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 \
   --master-port 29790 tests/dist_worker.py solve gpurun_out/dist_t.json 128 pcg > gpurun_out/dist_timing.log 2>&1 || { echo FAILED; tail -30 gpurun_out/dist_timing.log; exit 1; }
grep "partitioned setup" gpurun_out/dist_timing.log | tail -20
cat gpurun_out/dist_t.json
