#!/bin/bash
# round 4: small-call latency against the task length of the chain
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r04_lat}
mkdir -p $OUT
for L in default 256 512; do
  echo "== SEG_L $L"
  if [ $L = default ]; then timeout -k 5 120 python tools/latency.py > $OUT/lat_$L.jsonl 2>$OUT/lat_$L.err
  else ZOPFLI_AMD_SEG_L=$L timeout -k 5 120 python tools/latency.py > $OUT/lat_$L.jsonl 2>$OUT/lat_$L.err; fi
  python - $OUT/lat_$L.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); b=d["breakdown_ms"]
    print(d["cls"], d["size"], "n", d["numiterations"], "ms", d["ms_min"], {k:b[k] for k in ("tables","greedy","squeeze","split","dp_kernel","trace_kernel","cost_model","encode")})
PY
done
