#!/bin/bash
# tasks started at cut points of the DP (k_cutpoints): parity probe, then bench lines with and without
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-cuts}
mkdir -p $OUT
cd $REPO
for cfg in ${CFGS:-"1024:0" "0:0" "1024:2048"}; do
  IFS=: read cuts segl <<< "$cfg"
  export ZOPFLI_AMD_SEG_CUTS=$cuts
  if [ "$segl" != "0" ]; then export ZOPFLI_AMD_SEG_L=$segl; else unset ZOPFLI_AMD_SEG_L; fi
  echo "== SEG_CUTS=$cuts SEG_L=${segl}"
  if [ "$cuts" != "0" ] && [ "${PROBE:-1}" = "1" ]; then timeout 600 python tests/seg_probe.py 2>&1 | tail -1 | cut -c1-300; fi
  for c in ${CASES:-T X}; do
    timeout 600 python bench.py --cls $c --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b_${c}_${cuts}_${segl}.json 2> $OUT/b_${c}_${cuts}_${segl}.err
    python - $OUT/b_${c}_${cuts}_${segl}.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]; ch=d["roofline"]["chain"]
    print(d["config"]["workload"][:8], d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"], "dp", b["dp_kernel"], "chain ms/run", d["roofline"]["avg_launch_ms"], "acc", ch["accepted_frac"], "state", ch["rerun_state_frac"], "level", ch["rerun_level_frac"], "pos_rerun", ch["positions_rerun_frac"])
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1200:])
PY
  done
done
ZOPFLI_AMD_SEG_CUTS=1024 ZOPFLI_AMD_PROF=1 timeout 300 python bench.py --cls T --size 20000000 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | grep k_cutpoints | head -2
