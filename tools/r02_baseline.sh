#!/bin/bash
# Round-2 baseline on the GPU box: big goldens for classes T/X/M + bench lines per class with the round-1 kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02_base
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "full_size" > $OUT/pytest_full_size.log 2>&1
tail -3 $OUT/pytest_full_size.log
for cls in T X M; do
  for bs in 0 1; do
    timeout 300 python bench.py --cls $cls --blocksplitting $bs --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_${cls}_bs${bs}.json 2> $OUT/bench_${cls}_bs${bs}.err
    cut -c1-400 $OUT/bench_${cls}_bs${bs}.json
  done
  ZOPFLI_AMD_PROF=1 timeout 300 python bench.py --cls $cls --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_${cls}.json 2> $OUT/prof_${cls}.err
  grep -A8 "squeeze prof" $OUT/prof_${cls}.err | tail -9
done
