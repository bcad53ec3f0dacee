#!/bin/bash
# chain task warm-up length (ZOPFLI_AMD_SEG_WARM) against acceptance and time, per class
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in ${WARMS:-512 384 256}; do for c in ${CLASSES:-T X P}; do
  ZOPFLI_AMD_SEG_WARM=$w python bench.py --cls $c --size ${SIZE:-50000000} --steps 2 --warmup 1 --no-cpu-baseline > /tmp/w.json 2>/dev/null
  python - $w $c <<'PY'
import json,sys
d=json.load(open("/tmp/w.json")); b=d["breakdown_s_per_step"]; ch=d["roofline"]["chain"]
print("warm", sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], "dp", b["dp_kernel"], "accepted", ch["accepted_frac"], "state", ch["rerun_state_frac"], "rerun pos", ch["positions_rerun_frac"])
PY
done; done
