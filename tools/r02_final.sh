#!/bin/bash
# end-of-round record: the GPU suite, the default bench line, kernel stats of the same command, bench lines per class
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-final}
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-200 $OUT/bench_default.json
TAG=${TAG:-final}/classes CLASSES="${CLASSES:-T X M}" tools/r02_classes.sh
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
cp $OUT/stats/r_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python - $OUT/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5: print(f'{r["Name"][:50]:50s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]:>6s}%')
PY
