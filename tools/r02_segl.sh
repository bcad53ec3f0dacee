#!/bin/bash
# chain task length (ZOPFLI_AMD_SEG_L) against time, class T/X 100 MB
cd ${GRAFT_REPO_ROOT:-/root/repo}
for l in ${LS:-2048 4096 8192}; do for c in ${CLASSES:-T X}; do
  ZOPFLI_AMD_SEG_L=$l python bench.py --cls $c --steps 2 --warmup 1 --no-cpu-baseline > /tmp/w.json 2>/dev/null
  python - $l $c <<'PY'
import json,sys
d=json.load(open("/tmp/w.json")); b=d["breakdown_s_per_step"]; ch=d["roofline"]["chain"]
print("L", sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["bitexact_vs_reference"], "dp", b["dp_kernel"], "accepted", ch["accepted_frac"], "rerun pos", ch["positions_rerun_frac"])
PY
done; done
