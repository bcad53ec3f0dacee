#!/bin/bash
# round 5: the cooperative run-task job (zmx_dp6.h) — parity first (the run paths' tests and fuzz against the real reference),
# then classes Z and M at 100 MB with ZOPFLI_AMD_COOP = 1 / 0 side by side
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r05_coop}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "squeeze_runs or chain_task or run_paths or tie_rule or golden or guard" > $OUT/parity.log 2>&1; grep -a "passed\|failed\|error" $OUT/parity.log | tail -3
grep -a "^E " $OUT/parity.log | head -20
timeout -k 10 200 python tools/fuzz_runs.py ${FUZZ_CASES:-60} 21 > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
for coop in 1 0; do
  for cls in Z M; do
    ZOPFLI_AMD_COOP=$coop timeout -k 5 120 python bench.py --cls $cls --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/bench_${cls}_coop$coop.json 2> $OUT/bench_${cls}_coop$coop.err
    python - $OUT/bench_${cls}_coop$coop.json $cls $coop <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print("class", sys.argv[2], "coop", sys.argv[3], "MB/s", d["value"], "bitexact", d["bitexact_vs_reference"], "chain ms/run", r["avg_launch_ms"], r.get("chain"))
except Exception as e: print("ERR", sys.argv[1], e)
PY
  done
done
tail -5 $OUT/bench_Z_coop1.err
