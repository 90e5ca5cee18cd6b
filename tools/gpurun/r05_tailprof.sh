cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --no-kernel-table --no-plain-csr --no-aggressive --no-traffic --no-cpu-defaults --no-side-configs"
HDA_TAIL_PROF=40 HDA_TAIL_NNZ=100000 timeout -k 10 300 python bench.py --steps 1 --warmup 1 $FLAGS 2>&1 | grep "tail stamp" | tail -24
echo ----
HDA_TAIL_PROF=40 HDA_TAIL_NNZ=60000 timeout -k 10 300 python bench.py --steps 1 --warmup 1 $FLAGS 2>&1 | grep "tail stamp" | tail -12
