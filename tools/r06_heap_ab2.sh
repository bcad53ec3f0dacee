#!/bin/bash
# round 6: the block cache's smallest cached request (ZOPFLI_AMD_HOST_CACHE_MIN) against mallopt, T and R, bs1 only matters
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_heap_ab2}
mkdir -p $OUT
for rep in 1 2; do
for cls in T R; do
  for v in "min32k:ZOPFLI_AMD_HOST_CACHE_MIN=32768" "min4k:ZOPFLI_AMD_HOST_CACHE_MIN=4096" "min1k:ZOPFLI_AMD_HOST_CACHE_MIN=1024" "mallopt:ZOPFLI_AMD_KEEP_HEAP=1 ZOPFLI_AMD_HOST_CACHE_MB=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs timeout -k 5 200 python bench.py --cls $cls --steps 3 --warmup 1 --no-cpu-baseline --entry zopfli_compress > $OUT/${cls}_${name}_$rep.json 2> $OUT/${cls}_${name}_$rep.err
    python - $OUT/${cls}_${name}_$rep.json "$cls $name" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); b=d.get("blocksplitting1",{})
    print(f'  class {sys.argv[2]}: bs0 {d["value"]} MB/s {d["ms_per_step"]} ms | bs1 {b.get("value")} {b.get("ms_per_step")} bitexact {d["bitexact_vs_reference"]} {b.get("bitexact_vs_reference")}')
except Exception as e: print("ERR", sys.argv[2], e)
PY
  done
done
done
