#!/bin/bash
# round 6: the block-split search's block sizes on the device (ZOPFLI_AMD_DEVICE_SPLIT=1, the default) against the host's (=0):
# class lines with the reference's default block splitting, and the call's phases (ZOPFLI_AMD_TRACE_CALL)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r06_devsplit}
mkdir -p $OUT
cd $REPO
for cls in ${CLASSES:-R P T X}; do
  for ds in ${DS:-0 1}; do
    ZOPFLI_AMD_DEVICE_SPLIT=$ds timeout 900 python bench.py --cls $cls --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-small-files --entry zopfli_compress > $OUT/bench_${cls}_$ds.json 2> $OUT/bench_${cls}_$ds.err
    python - $OUT/bench_${cls}_$ds.json $ds <<'PY' | tee -a $OUT/log.txt
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    s1=d.get("blocksplitting1") or {}
    b=s1.get("breakdown_s_per_step") or {}
    print(d["config"]["workload"][:8], "device_split", sys.argv[2], "| bs0", d["value"], "MB/s | bs1", s1.get("value"), "MB/s", s1.get("ms_per_step"), "ms bitexact", s1.get("bitexact_vs_reference"), "| split", b.get("split"), "greedy", b.get("greedy"), "squeeze", b.get("squeeze"))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
  done
done
if [ "${TRACE:-1}" = "1" ]; then
  for ds in 0 1; do
    echo "== trace R device_split $ds" | tee -a $OUT/log.txt
    ZOPFLI_AMD_DEVICE_SPLIT=$ds ZOPFLI_AMD_TRACE_CALL=1 timeout 300 python bench.py --cls R --steps 1 --warmup 1 --no-cpu-baseline --no-small-files --entry zopfli_compress --blocksplitting 1 2>&1 >/dev/null | grep -v "^$" | tail -25 | tee -a $OUT/log.txt
  done
fi
