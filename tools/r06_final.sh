#!/bin/bash
# round 6: the measurement set of profiles/r06_*: the GPU suite, the default bench line (with small_files), the class lines
# (blocksplitting 0 and 1), configs[3] at size, latency, PNG at size with the reference timed on the same host, and the profile
# sets (kernel stats + PMC passes) of the default line (with an LDS pass), of class Z and of configs[3]
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
T=${TAG:-r06_final}
OUT=$REPO/gpurun_out/$T
mkdir -p $OUT
if [ "${SUITE:-1}" = 1 ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; grep -a "passed\|failed\|error" $OUT/suite.log | tail -3
fi
timeout -k 5 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-260 $OUT/bench_default.json
if [ "${CLASSLINES:-1}" = 1 ]; then
TAG=$T/classes STEPS=2 bash tools/r03_classes.sh 2>&1 | tee $OUT/classes.txt
timeout 600 python bench.py --cls M --size 200000000 --numiterations 50 --blocksplitting 1 --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --no-small-files > $OUT/config3_M200_n50.json 2> $OUT/config3.err
python - $OUT/config3_M200_n50.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("configs[3] M 200 MB n=50 bs=1:", d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"])
except Exception as e: print("ERR", e)
PY
timeout 200 python tools/latency.py > $OUT/latency.jsonl 2> $OUT/latency.err; cut -c1-200 $OUT/latency.jsonl
TMPDIR=/tmp timeout -k 5 400 python tools/png_at_size.py 4096 --ref > $OUT/png4096_same_host.json 2> $OUT/png4096.err; cat $OUT/png4096_same_host.json
fi
[ "${PROFILES:-1}" = 1 ] || exit 0
EXTRA_SET="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" BENCH_ARGS="--steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1" TAG=$T/prof bash tools/collect_profiles.sh > $OUT/profiles.txt 2>&1; tail -30 $OUT/profiles.txt | cut -c1-250
BENCH_ARGS="--cls Z --steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1" TAG=$T/profZ bash tools/collect_profiles.sh > $OUT/profilesZ.txt 2>&1; tail -12 $OUT/profilesZ.txt | cut -c1-250
BENCH_ARGS="--cls M --size 200000000 --numiterations 50 --blocksplitting 1 --steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1" TAG=$T/profC3 bash tools/collect_profiles.sh > $OUT/profilesC3.txt 2>&1; tail -12 $OUT/profilesC3.txt | cut -c1-250
