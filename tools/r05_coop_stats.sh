#!/bin/bash
# round 5: per-kernel durations of one squeeze step on class Z (and M) with the cooperative run-task job on and off
# (rocprofv3 --kernel-trace --stats), after a quick parity check of the build
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r05_coop_stats}
mkdir -p $OUT
cd $REPO
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "squeeze_runs or chain_task or run_paths" > $OUT/parity.log 2>&1; grep -a "passed\|failed\|error" $OUT/parity.log | tail -2
cd /tmp && export TMPDIR=/tmp
for coop in 1 0; do
  for SPEC in ${SPECS:-Z:100000000}; do
    cls=${SPEC%%:*}; size=${SPEC##*:}
    ZOPFLI_AMD_COOP=$coop timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${cls}_$coop -o r -- python $REPO/bench.py --cls $cls --size $size --steps 1 --warmup 0 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/stats_${cls}_$coop.log 2>&1
    echo "== class $cls $size coop $coop: $(grep -a -o '"value": [0-9.]*' $OUT/stats_${cls}_$coop.log | head -1)"
    python - $OUT/stats_${cls}_$coop/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5: print(f'{r["Name"][:64]:64s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.1f} ms {r["Percentage"]:>6s}%')
PY
  done
done
