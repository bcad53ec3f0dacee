#!/bin/bash
# round 4, last call: the lines that carry k_match5's own counts (roofline_match.skip_walk), the profile set of the default
# line on the final device sources, and the GPU suite
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r04_final2
mkdir -p $OUT
for cls in P B M; do
  timeout -k 5 60 python bench.py --cls $cls --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 > $OUT/bench_$cls.json 2> $OUT/bench_$cls.err
  python - $OUT/bench_$cls.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(d["config"]["workload"][:8], d["value"], "bitexact", d["bitexact_vs_reference"], json.dumps(d["roofline_match"].get("skip_walk"))[:400])
except Exception as e: print("ERR", e)
PY
done
BENCH_ARGS="--steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1" TAG=r04_final2/prof bash tools/collect_profiles.sh > $OUT/profiles.txt 2>&1; tail -8 $OUT/profiles.txt | cut -c1-250
timeout -k 5 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_T.json 2> $OUT/bench_T.err; cut -c1-200 $OUT/bench_T.json
timeout -k 10 260 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; grep -a "passed\|failed\|error" $OUT/suite.log | tail -3
