#!/bin/bash
# round 6: the library's own block cache (block_cache.h) against round 5's mallopt: ZopfliCompress, 100 MB, n = 15,
# without / with block splitting, classes T and R.  KEEP_HEAP=1 = mallopt as in round 5; CACHE=0 = plain malloc / free.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_heap_ab}
mkdir -p $OUT
for rep in 1 2; do
for cls in T R; do
  for v in "cache:ZOPFLI_AMD_KEEP_HEAP=0 ZOPFLI_AMD_HOST_CACHE_MB=1024" "mallopt:ZOPFLI_AMD_KEEP_HEAP=1 ZOPFLI_AMD_HOST_CACHE_MB=0" "neither:ZOPFLI_AMD_KEEP_HEAP=0 ZOPFLI_AMD_HOST_CACHE_MB=0" "both:ZOPFLI_AMD_KEEP_HEAP=1 ZOPFLI_AMD_HOST_CACHE_MB=1024"; do
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
