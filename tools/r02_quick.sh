#!/bin/bash
# quick GPU check: chain-task probe, bench lines for the classes given, kernel stats for class T
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-quick}
mkdir -p $OUT
cd $REPO
timeout 600 python tests/seg_probe.py 2>&1 | tail -2
for c in ${CLASSES:-T X}; do
  timeout 300 python bench.py --cls $c --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b_$c.json 2> $OUT/b_$c.err
  python - $OUT/b_$c.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(d["config"]["workload"][:8], d["value"], d["ms_per_step"], d["bitexact_vs_reference"], d["roundtrip_ok"], d["chain_tasks_per_step"], d["breakdown_s_per_step"])
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
if [ "${STATS:-1}" = "1" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats.log 2>&1
python - $OUT/stats/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(f'{r["Name"][:50]:50s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]:>6s}%')
PY
fi
