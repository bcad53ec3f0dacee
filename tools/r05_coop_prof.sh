#!/bin/bash
# round 5: where wave 0 of the cooperative run-task job spends its cycles (ZOPFLI_AMD_PROF, class Z)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r05_coop_prof}
mkdir -p $OUT
cd $REPO
ZOPFLI_AMD_PROF=1 timeout -k 5 120 python bench.py --cls Z --size ${SIZE:-20000000} --steps 1 --warmup 0 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/prof_Z.json 2> $OUT/prof_Z.err
grep -a "coop prof" $OUT/prof_Z.err | tail -3
grep -a "squeeze prof" $OUT/prof_Z.err | tail -1 | cut -c1-300
cut -c1-200 $OUT/prof_Z.json
