#!/bin/bash
# end-of-round check after the cut-point change: GPU suite, default line, class lines, kernel stats
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-final2}
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-160 $OUT/bench_default.json
TAG=${TAG:-final2}/classes CLASSES="${CLASSES:-T X M}" SPLITS="${SPLITS:-0 1}" tools/r02_classes.sh
