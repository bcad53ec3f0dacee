#!/bin/bash
# task length and exact head length with tasks started at cut points (ZOPFLI_AMD_SEG_CUTS)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-cuts2}
mkdir -p $OUT
cd $REPO
for cfg in ${CFGS:-"1024:1024:0" "1024:2048:8192" "1024:2048:4096" "1024:1024:4096" "1024:1536:0"}; do
  IFS=: read cuts segl head <<< "$cfg"
  export ZOPFLI_AMD_SEG_CUTS=$cuts ZOPFLI_AMD_SEG_L=$segl
  if [ "$head" != "0" ]; then export ZOPFLI_AMD_SEG_HEAD=$head; else unset ZOPFLI_AMD_SEG_HEAD; fi
  for c in ${CASES:-T}; do
    timeout 600 python bench.py --cls $c --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
    python - $OUT/b.json "$cfg" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]; ch=d["roofline"]["chain"]
    print(sys.argv[2], d["config"]["workload"][:8], d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "dp", b["dp_kernel"], "chain ms/run", d["roofline"]["avg_launch_ms"], "acc", ch["accepted_frac"], "level", ch["rerun_level_frac"], "tasks", ch["tasks_per_launch"])
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1200:])
PY
  done
done
