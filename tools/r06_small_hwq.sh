#!/bin/bash
# round 6: many small files through 16 callers against the HIP runtime's number of hardware queues (GPU_MAX_HW_QUEUES, default 4)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_small_hwq}
mkdir -p $OUT
for v in "default:" "hwq8:GPU_MAX_HW_QUEUES=8" "hwq16:GPU_MAX_HW_QUEUES=16" "hwq24:GPU_MAX_HW_QUEUES=24"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout -k 5 400 python bench.py --small-files-only > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY' | tee -a $OUT/log.txt
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
