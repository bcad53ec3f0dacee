#!/bin/bash
# round 6: many small files through K concurrent ZopfliCompress callers (bench.py --small-files-only), variants by environment
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_small}
mkdir -p $OUT
for v in "default:" "lanes3:ZOPFLI_AMD_SMALL_LANES=3" "lanes16:ZOPFLI_AMD_SMALL_LANES=16" "threads1:ZOPFLI_AMD_THREADS=1"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout -k 5 400 python bench.py --small-files-only > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    for ln in open(sys.argv[1]):
        if ln.startswith('{"small_files"'):
            d=json.loads(ln)["small_files"]
            for s in d["sets"]:
                print(sys.argv[2], s["files"], "x", s["bytes_each"], {k:v["value"] for k,v in s["callers"].items()}, "ref all cores", s.get("reference_all_cores",{}).get("value"), "bitexact", s.get("bitexact_vs_reference_first4"))
except Exception as e: print("ERR", sys.argv[2], e)
PY
done
