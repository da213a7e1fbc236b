#!/bin/bash
# Scaling sweep on ONE node with up to 8 MI355X (nobody has run this yet: no multi-GPU hardware was available to the build).
#   bash scripts/scale_bench.sh [steps=200] [warmup=20] [gpus="1 2 4 8"]
# Per N: weak (4096 rays per GPU), strong = C4 (the 4096-ray batch / N), strong replayed as hipGraphs with the RCCL all-reduce
# recorded inside the graph.  One JSON line per run on stdout (bench.py's contract); efficiency = value(N) / (N * value(1)) for
# weak, ms_per_step(1) / (N * ms_per_step(N)) ... is left to the reader: bench.py never reports efficiency itself.
STEPS=${1:-200}; WARM=${2:-20}; GPUS=${3:-"1 2 4 8"}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # N, extra flags...
  local n=$1; shift
  if [ "$n" = 1 ]; then python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-extra --no-cpu-baseline "$@"
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29700 + n)) bench.py --gpus $n --steps $STEPS --warmup $WARM --no-extra --no-cpu-baseline "$@"; fi
}
for n in $GPUS; do
  run $n 2>/dev/null
  run $n --scaling strong 2>/dev/null
  run $n --scaling strong --graph --graph-collective capture 2>/dev/null
done
