#!/bin/bash
# round 5: mid snapshots for text tasks too — chain parity, then latency of small calls and the default line's chain
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r05_mid_text}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "squeeze_runs or chain_task or run_paths or tie_rule or golden or guard" > $OUT/parity.log 2>&1; grep -a "passed\|failed\|error" $OUT/parity.log | tail -2
for mid in 1 0; do
  echo "== ZOPFLI_AMD_SEG_MID=$mid"
  ZOPFLI_AMD_SEG_MID=$mid timeout 200 python tools/latency.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); b=r['breakdown_ms']; print('  ', r['cls'], r['size'], r['numiterations'], 'ms', r['ms_min'], 'dp_kernel', b['dp_kernel'], 'squeeze', b['squeeze'])"
  ZOPFLI_AMD_SEG_MID=$mid timeout -k 5 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-blocksplitting1 > $OUT/bench_T_mid$mid.json 2> $OUT/bench_T_mid$mid.err
  python - $OUT/bench_T_mid$mid.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("   T 100 MB: MB/s", d["value"], "resident", d["value_resident"], "bitexact", d["bitexact_vs_reference"], "chain ms/run", r["avg_launch_ms"], r.get("chain"))
PY
done
