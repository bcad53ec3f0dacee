#!/bin/bash
# integer chain step on/off: probe vs the oracle, then bench lines
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-intpath}
mkdir -p $OUT
cd $REPO
for ip in ${IPS:-1 0}; do
  echo "== INT_PATH=$ip"
  ZOPFLI_AMD_INT_PATH=$ip timeout 600 python tests/seg_probe.py 2>&1 | tail -1 | cut -c1-300
  for c in ${CASES:-T:100000000:0 X:100000000:0 T:100000000:1}; do
    IFS=: read cls sz bs <<< "$c"
    ZOPFLI_AMD_INT_PATH=$ip ZOPFLI_AMD_PROF=${PROF:-} timeout 600 python bench.py --cls $cls --size $sz --blocksplitting $bs --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b_${cls}_${bs}_$ip.json 2> $OUT/b_${cls}_${bs}_$ip.err
    python - $OUT/b_${cls}_${bs}_$ip.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]
    print(d["config"]["workload"][:28], d["value"], "MB/s", d["ms_per_step"], "ms rt", d["roundtrip_ok"], "bitexact", d["bitexact_vs_reference"], "dp", b["dp_kernel"], "tables", b["tables"], "acc", d["roofline"]["chain"]["accepted_frac"])
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
  done
done
if [ "${STATS:-0}" = "1" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats.log 2>&1
python - $OUT/stats/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5: print(f'{r["Name"][:50]:50s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]:>6s}%')
PY
fi
