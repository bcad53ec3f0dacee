#!/bin/bash
# round 6: class Z / M / T lines on a resident input (one context) + the chain's kernel times
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_z}
mkdir -p $OUT
for cls in ${CLASSES:-Z M T}; do
  timeout -k 5 300 python bench.py --cls $cls --steps 2 --warmup 1 --no-cpu-baseline --entry resident --no-blocksplitting1 2>$OUT/$cls.err | grep '^{"metric"' > $OUT/$cls.json
  python - $OUT/$cls.json $cls <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]; b=d["breakdown_s_per_step"]
    print(f'class {sys.argv[2]}: {d["value"]} MB/s, chain {r["avg_launch_ms"]} ms per run, bitexact {d["bitexact_vs_reference"]}, tables {b.get("tables")} squeeze {b.get("squeeze")} match {b.get("match_kernel")} chain-tasks {r.get("chain")}')
except Exception as e: print("ERR", sys.argv[2], e)
PY
done
