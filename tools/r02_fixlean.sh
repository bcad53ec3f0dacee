#!/bin/bash
# which job for the serial re-runs of k_dp4_fix: lean one-wave job from N generic windows on (ZOPFLI_AMD_FIX_LEAN)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-fixlean}
mkdir -p $OUT
cd $REPO
for fl in ${LEANS:-0 24 100000}; do
  echo "== FIX_LEAN=$fl"
  ZOPFLI_AMD_FIX_LEAN=$fl timeout 600 python tests/seg_probe.py 2>&1 | tail -1 | cut -c1-400
  for c in ${CASES:-Z:10000000 B:2000000 M:100000000 X:100000000}; do
    cls=${c%%:*}; sz=${c##*:}
    ZOPFLI_AMD_FIX_LEAN=$fl ZOPFLI_AMD_PROF=${PROF:-} timeout 600 python bench.py --cls $cls --size $sz --steps 1 --warmup 1 --no-cpu-baseline > $OUT/b_${cls}_$fl.json 2> $OUT/b_${cls}_$fl.err
    python - $OUT/b_${cls}_$fl.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]
    print(d["config"]["workload"][:28], d["value"], "MB/s", d["ms_per_step"], "ms rt", d["roundtrip_ok"], "bitexact", d["bitexact_vs_reference"], "dp", b["dp_kernel"], "match", b["match_kernel"], "tables", b["tables"], {k:v for k,v in d["chain_tasks_per_step"].items() if k in ("tasks","accepted","positions_rerun")})
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
  done
done
