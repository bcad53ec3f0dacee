#!/bin/bash
# round 4: how a 100 MB call is dealt — contexts per device (ZOPFLI_AMD_SPLIT_WAYS) x master blocks per batch
# (ZOPFLI_AMD_PARTS_PER_BATCH), with and without block splitting
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r04_ways}
mkdir -p $OUT
for bs in ${BS:-1 0}; do
for ways in ${WAYS:-3 4 6}; do
for ppb in ${PPB:-256 17 9}; do
  f=$OUT/bs${bs}_w${ways}_p${ppb}.json
  ZOPFLI_AMD_SPLIT_WAYS=$ways ZOPFLI_AMD_LANES=$ways ZOPFLI_AMD_PARTS_PER_BATCH=$ppb timeout -k 5 120 python bench.py --cls ${CLS:-T} --blocksplitting $bs --steps 3 --warmup 1 --no-cpu-baseline --no-blocksplitting1 > $f 2> $OUT/err.txt
  python - $f $bs $ways $ppb <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("bs",sys.argv[2],"ways",sys.argv[3],"ppb",sys.argv[4],":",d["value"],"MB/s",d["ms_per_step"],"ms", "bitexact", d.get("bitexact_vs_reference"))
except Exception as e: print("ERR", sys.argv[2:], e)
PY
done; done; done
