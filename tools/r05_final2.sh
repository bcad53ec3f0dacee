#!/bin/bash
# round 5, last call: the GPU suite on the final device sources, the profile sets of the default line and of class Z
# (their PMC files carry the sources' hash: bench.py quotes roofline.traffic only for the same build), the default line
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r05_final2}
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; grep -a "passed\|failed\|error" $OUT/suite.log | tail -3
BENCH_ARGS="--steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1" TAG=${TAG:-r05_final2}/prof bash tools/collect_profiles.sh > $OUT/profiles.txt 2>&1; tail -22 $OUT/profiles.txt | cut -c1-250
BENCH_ARGS="--cls Z --steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1" TAG=${TAG:-r05_final2}/profZ bash tools/collect_profiles.sh > $OUT/profilesZ.txt 2>&1; tail -8 $OUT/profilesZ.txt | cut -c1-250
cp $OUT/prof/pmc.json profiles/r05_bench100MB_pmc.json; cp $OUT/profZ/pmc.json profiles/r05_classZ100MB_pmc.json
timeout -k 5 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-260 $OUT/bench_default.json
timeout -k 5 200 python bench.py --cls Z --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/bench_Z.json 2> $OUT/bench_Z.err; cut -c1-200 $OUT/bench_Z.json
timeout 200 python tools/latency.py > $OUT/latency.jsonl 2> $OUT/latency.err; cut -c1-160 $OUT/latency.jsonl
