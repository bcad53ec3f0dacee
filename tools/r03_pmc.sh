#!/bin/bash
# PMC passes (one rocprofv3 run per line of tools/pmc_sets.txt) over ONE bench step of a class: counters of one kernel,
# per launch.   CLS=P SIZE=20000000 KERNEL=k_match3 ZOPFLI_AMD_MATCH=3 TAG=pmc_p3 bash tools/r03_pmc.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r set; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $REPO/bench.py --cls ${CLS:-T} --size ${SIZE:-40000000} --steps 1 --warmup 0 --numiterations 1 --entry resident --no-cpu-baseline > $OUT/p$i.log 2>&1
  python - $OUT/p$i/p_counter_collection.csv "${KERNEL:-k_match2}" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    for k in acc: print(f"{k:32s} per launch {acc[k]/max(n[k],1):16.0f}  ({n[k]} launches)")
except Exception as e: print("ERR",e)
PY
done < $REPO/tools/pmc_sets.txt
