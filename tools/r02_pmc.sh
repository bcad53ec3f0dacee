#!/bin/bash
# PMC passes over a reduced workload (SIZE bytes): counters of one kernel, per launch
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r set; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $REPO/bench.py --size ${SIZE:-40000000} --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p$i.log 2>&1
  python - $OUT/p$i/p_counter_collection.csv "${KERNEL:-k_dp5_spec}" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    for k in acc: print(f"{k:32s} per launch {acc[k]/max(n[k],1):16.0f}  ({n[k]} launches)")
except Exception as e: print("ERR",e)
PY
done
