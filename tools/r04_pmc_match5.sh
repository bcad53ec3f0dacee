#!/bin/bash
# round 4: PMC passes over table builds with the skip-walk forced (tools/r04_match5.py time, no torch): what k_match5
# waits for.   SPEC=T:50000000 KERNEL=k_match5 bash tools/r04_pmc_match5.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r04_pmc5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while IFS= read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $REPO/tools/r04_match5.py time ${SPEC:-T:50000000} > $OUT/p$i.log 2>&1
  python - $OUT/p$i/p_counter_collection.csv "${KERNEL:-k_match5}" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(set)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in acc: print(f"{k:36s} per launch {acc[k]/max(len(n[k]),1):18.0f}  ({len(n[k])} launches)")
except Exception as e: print("ERR",e)
PY
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM_RD
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_CYCLES
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SETS
