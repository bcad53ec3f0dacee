#!/bin/bash
# bench lines per class and block-splitting mode, plus kernel stats for class M
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-classes}
mkdir -p $OUT
cd $REPO
for cls in ${CLASSES:-T X M}; do
  for bs in ${SPLITS:-0 1}; do
    timeout 300 python bench.py --cls $cls --blocksplitting $bs --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_${cls}_bs${bs}.json 2> $OUT/bench_${cls}_bs${bs}.err
    python - $OUT/bench_${cls}_bs${bs}.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]; c=d["chain_tasks_per_step"]
    print(d["config"]["workload"][:8], "bs", d["config"]["workload"].split("blocksplitting=")[1][0], d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"],
          {k: round(v*1e3,1) for k,v in b.items() if k in ("tables","squeeze","dp_kernel","trace_kernel","split","encode","cost_model","match_kernel","hash_kernels")},
          "accepted", round(c["accepted"]/max(c["tasks"],1),4), "pos_rerun", round(c["positions_rerun"]/max(c["positions"],1),4))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
  done
done
if [ -n "${STATCLS:-}" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/bench.py --cls $STATCLS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats.log 2>&1
python - $OUT/stats/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5: print(f'{r["Name"][:50]:50s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]:>6s}%')
PY
fi
