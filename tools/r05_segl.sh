#!/bin/bash
# round 5: task length again, now that k_dp4_fix judges a chunk's tasks at once (the per-task walk was what made short tasks lose)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
for L in ${LS_SMALL:-default 512 256}; do
  echo "== small calls, ZOPFLI_AMD_SEG_L=$L"
  if [ $L = default ]; then unset ZOPFLI_AMD_SEG_L; else export ZOPFLI_AMD_SEG_L=$L; fi
  timeout 200 python tools/latency.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); b=r['breakdown_ms']; print('  ', r['cls'], r['size'], r['numiterations'], 'ms', r['ms_min'], 'dp_kernel', b['dp_kernel'], 'squeeze', b['squeeze'])"
done
for L in ${LS_BIG:-default 1024 512}; do
  if [ $L = default ]; then unset ZOPFLI_AMD_SEG_L; else export ZOPFLI_AMD_SEG_L=$L; fi
  for cls in T P; do
  timeout -k 5 200 python bench.py --cls $cls --steps 2 --warmup 1 --no-cpu-baseline --no-blocksplitting1 > /tmp/b.json 2> /tmp/b.err
  python - /tmp/b.json $L $cls <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("== 100 MB class", sys.argv[3], "SEG_L", sys.argv[2], ": MB/s", d["value"], "resident", d["value_resident"], "bitexact", d["bitexact_vs_reference"], "chain ms/run", r["avg_launch_ms"], "pos_rerun", r["chain"]["positions_rerun_frac"])
PY
  done
done
