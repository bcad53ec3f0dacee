#!/bin/bash
# round 6: the integer step of the OTHER rows (ZOPFLI_AMD_INT_PATH bits: 2 = from the workgroup's table, 4 = the lean job's own table, 8 = class-1 windows of the lean job)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_bisect}
mkdir -p $OUT
for ip in ${IPS:-1 3 7 15}; do
  echo "== INT_PATH=$ip" | tee -a $OUT/log.txt
  ZOPFLI_AMD_INT_PATH=$ip timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "squeeze_runs or chain_task_paths or run_paths_fuzz" 2>&1 | tail -4 | tee -a $OUT/log.txt
  for cls in ${CLASSES:-Z M}; do
    ZOPFLI_AMD_INT_PATH=$ip timeout -k 5 300 python bench.py --cls $cls --steps 2 --warmup 1 --no-cpu-baseline --entry resident --no-blocksplitting1 2>$OUT/$cls.$ip.err | grep '^{"metric"' > $OUT/$cls.$ip.json
    python - $OUT/$cls.$ip.json $cls <<'PY' | tee -a $OUT/log.txt
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]; b=d["breakdown_s_per_step"]
    print(f'class {sys.argv[2]}: {d["value"]} MB/s, chain {r["avg_launch_ms"]} ms per run, bitexact {d["bitexact_vs_reference"]}, chain-tasks {r.get("chain")}')
except Exception as e: print("ERR", sys.argv[2], e)
PY
  done
done
