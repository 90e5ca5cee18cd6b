#!/bin/bash
# strong-scaling rehearsal on ONE GPU: R ranks share the card through the staged transport
set -o pipefail
R=${1:-4}; G=${2:-256}
mkdir -p gpurun_out
export PYTHONPATH=$PWD OMP_NUM_THREADS=1
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$R --master-addr 127.0.0.1 --master-port 29812 \
   bench.py --gpus $R --steps 2 --warmup 1 --grid $G --strong > gpurun_out/strong_${R}_${G}.log 2>&1 || { echo FAILED; tail -40 gpurun_out/strong_${R}_${G}.log; exit 1; }
grep -v "^\[W\|Gloo\|amdgpu.ids" gpurun_out/strong_${R}_${G}.log | tail -6
