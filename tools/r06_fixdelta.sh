#!/bin/bash
# round 6: what shift do the tasks that k_dp4_fix re-runs on long-run data need?  (ZOPFLI_AMD_SEG_DEBUG=2 prints every re-run task's check)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_fixdelta}
mkdir -p $OUT
for cls in ${CLASSES:-Z}; do
  ZOPFLI_AMD_SEG_DEBUG=2 timeout -k 5 300 python bench.py --cls $cls --size ${SIZE:-10000000} --numiterations 4 --steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1 --no-small-files 2>$OUT/$cls.err | grep "^fix b" > $OUT/$cls.fix.txt
  wc -l $OUT/$cls.fix.txt
done
