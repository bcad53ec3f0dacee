#!/bin/bash
# k_match2 candidate filter on/off: parity tests of the match table, then bench lines (match_kernel seconds)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-matchf}
mkdir -p $OUT
cd $REPO
for f in ${FS:-1 0}; do
  echo "== MATCH_FILTER=$f"
  if [ "${TESTS:-1}" = "1" ] && [ "$f" = "1" ]; then
    ZOPFLI_AMD_MATCH_FILTER=$f timeout 600 python -m pytest tests -m gpu -x -q -k "match_table or change_point or stream_golden or full_size" 2>&1 | grep -E "passed|failed|error" | tail -2
  fi
  for c in ${CASES:-T X M:20000000}; do
    IFS=: read cls sz <<< "$c"
    ZOPFLI_AMD_MATCH_FILTER=$f ZOPFLI_AMD_PROF=${PROF:-} timeout 600 python bench.py --cls $cls --size ${sz:-100000000} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b_${cls}_$f.json 2> $OUT/b_${cls}_$f.err
    python - $OUT/b_${cls}_$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]
    print(d["config"]["workload"][:28], d["value"], "MB/s", d["ms_per_step"], "ms rt", d["roundtrip_ok"], "bitexact", d["bitexact_vs_reference"], "match", b["match_kernel"], "tables", b["tables"])
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
    grep -h "k_match2" $OUT/b_${cls}_$f.err | tail -2
  done
done
