#!/bin/bash
# where the host's block-split phase goes: thread budget and heap policy against the split / encode timers (block splitting on)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-split}
mkdir -p $OUT
cd $REPO
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
run() {
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --cls ${CLS:-T} --blocksplitting 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b_$tag.json 2> $OUT/b_$tag.err
  python - $OUT/b_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); b=d["breakdown_s_per_step"]
    print(sys.argv[2], d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "split", b["split"], "encode", b["encode"], "greedy", b["greedy"], "cost_model", b["cost_model"], "download", b["download"])
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
run default X=1
run keepheap ZOPFLI_AMD_KEEP_HEAP=1
run t16 ZOPFLI_AMD_THREADS=16
run t32 ZOPFLI_AMD_THREADS=32
run t64 ZOPFLI_AMD_THREADS=64
run t200 ZOPFLI_AMD_THREADS=200
