#!/bin/bash
# The 1 / 2 / 4 / 8-GPU lines of bench.py, weak (100 MB per GPU) and strong (100 MB in all), on a node that has the
# GPUs — one command for the day such a node is available (round 2 and 3 had one-GPU boxes only: no scaling number
# has been measured, none is claimed).  Usage: bash tools/scale.sh [max_gpus]
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO
MAXG=${1:-8}
for scaling in weak strong; do
  for n in 1 2 4 8; do
    [ $n -gt $MAXG ] && continue
    if [ $n -eq 1 ]; then
      python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --scaling $scaling
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
        bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline --scaling $scaling
    fi
  done
done
