#!/bin/bash
# round 5: where a ZopfliCompress call with the reference's default block splitting spends its time, per class
# (ZOPFLI_AMD_TRACE_CALL=1: the library's own phase timers per shard)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r05_bs1}
mkdir -p $OUT
for cls in ${CLASSES:-R P T}; do
  ZOPFLI_AMD_TRACE_CALL=1 timeout -k 5 200 python bench.py --cls $cls --blocksplitting 1 --steps 2 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --entry zopfli_compress > $OUT/bench_$cls.json 2> $OUT/bench_$cls.err
  echo "== class $cls bs 1: $(grep -o '"value": [0-9.]*' $OUT/bench_$cls.json | head -1) MB/s"
  grep -a "shard\|RunPartsSharded\|DeflateWhole\|phase\|batch" $OUT/bench_$cls.err | tail -14 | cut -c1-330
  python - $OUT/bench_$cls.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("  breakdown", d["breakdown_s_per_step"])
PY
done
