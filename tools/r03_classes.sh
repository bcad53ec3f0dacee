#!/bin/bash
# profiles/r03_classes.json: one bench line per class of synthetic data (100 MB, numiterations 15; each line holds the
# ZopfliCompress rate, the resident rate, the blocksplitting=1 rate and whether the output is the reference's)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r03_classes}
mkdir -p $OUT
cd $REPO
: > $OUT/classes.jsonl
for cls in ${CLASSES:-T X R P B Z M}; do
  timeout 900 python bench.py --cls $cls --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-small-files > $OUT/bench_${cls}.json 2> $OUT/bench_${cls}.err
  cat $OUT/bench_${cls}.json >> $OUT/classes.jsonl
  python - $OUT/bench_${cls}.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]; r=d["roofline"] or {}; c=r.get("chain",{}); m=d["roofline_match"] or {}
    s1=d.get("blocksplitting1") or {}
    print(d["config"]["workload"][:8], d["value"], "MB/s", d["ms_per_step"], "ms | resident", d["value_resident"], "| bs1", s1.get("value"), s1.get("bitexact_vs_reference"),
          "| bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"], "| chain ms/run", r.get("avg_launch_ms"), "match ms", round(m.get("seconds_per_step",0)*1e3,1),
          "accepted", c.get("accepted_frac"), "pos_rerun", c.get("positions_rerun_frac"))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
