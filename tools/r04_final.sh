#!/bin/bash
# round 4: the measurement set of profiles/r04_*: class lines, configs[3] at size, latency, match kernels, and the
# profile set (kernel stats + PMC passes) of the default bench line
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r04_final}
mkdir -p $OUT
TAG=${TAG:-r04_final}/classes STEPS=2 bash tools/r03_classes.sh 2>&1 | tee $OUT/classes.txt
timeout 600 python bench.py --cls M --size 200000000 --numiterations 50 --blocksplitting 1 --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 > $OUT/config3_M200_n50.json 2> $OUT/config3.err
python - $OUT/config3_M200_n50.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("configs[3] M 200 MB n=50 bs=1:", d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"])
except Exception as e: print("ERR", e)
PY
timeout 200 python tools/latency.py > $OUT/latency.jsonl 2> $OUT/latency.err; cut -c1-200 $OUT/latency.jsonl
TAG=${TAG:-r04_final}/match bash tools/r04_match.sh > $OUT/match.txt 2>&1; tail -9 $OUT/match.txt | cut -c1-300
BENCH_ARGS="--steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1" TAG=${TAG:-r04_final}/prof bash tools/collect_profiles.sh > $OUT/profiles.txt 2>&1; tail -30 $OUT/profiles.txt | cut -c1-250
