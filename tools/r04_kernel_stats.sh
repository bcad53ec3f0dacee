#!/bin/bash
# round 4: per-kernel times of the table builds (tools/r04_match5.py time), rocprofv3 kernel stats
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r04_m5prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SPEC in ${SPECS:-T:50000000 P:20000000}; do
  n=${SPEC%%:*}
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$n -o r -- python $REPO/tools/r04_match5.py time $SPEC > $OUT/stats_$n.log 2>&1
  echo "== $SPEC"
  python - $OUT/stats_$n/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(f'{r["Name"][:60]:60s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]:>6s}%')
PY
done
