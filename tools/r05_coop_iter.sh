#!/bin/bash
# round 5: one iteration on the cooperative job: run-path parity, the PROF breakdown on class Z (20 MB), Z / M at 100 MB
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r05_coop_iter}
mkdir -p $OUT
cd $REPO
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "squeeze_runs or chain_task or run_paths" > $OUT/parity.log 2>&1; grep -a "passed\|failed\|error" $OUT/parity.log | tail -2
ZOPFLI_AMD_PROF=1 timeout -k 5 120 python bench.py --cls Z --size 20000000 --steps 1 --warmup 0 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/prof_Z.json 2> $OUT/prof_Z.err
grep -a "coop prof" $OUT/prof_Z.err | tail -1
for cls in ${CLASSES:-Z M}; do
  timeout -k 5 120 python bench.py --cls $cls --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/bench_$cls.json 2> $OUT/bench_$cls.err
  python - $OUT/bench_$cls.json $cls <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print("class", sys.argv[2], "MB/s", d["value"], "bitexact", d["bitexact_vs_reference"], "chain ms/run", r["avg_launch_ms"])
except Exception as e: print("ERR", sys.argv[1], e)
PY
done
