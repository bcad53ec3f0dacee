#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + three PMC passes of the default bench.
# Outputs under gpurun_out/; copy the summaries to profiles/ afterwards.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profile_run
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 1 --warmup 0 --no-cpu-baseline}"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$n -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_$n.log 2>&1
done
tail -1 $OUT/stats.log | cut -c1-300
ls $OUT
