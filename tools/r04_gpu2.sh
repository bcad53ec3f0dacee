#!/bin/bash
# round 4: per-kernel times and counters of the kernel-5 table build (tools/r04_match5.py time)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r04_m5prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPEC="${SPEC:-T:50000000}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/tools/r04_match5.py time $SPEC > $OUT/stats.log 2>&1
python - $OUT/stats/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(f'{r["Name"][:60]:60s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]:>6s}%')
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $REPO/tools/r04_match5.py time $SPEC > $OUT/pmc_$i.log 2>&1
done
python - $OUT <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+"/pmc_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:40]
        if not any(x in k for x in ("k_match","k_levels","k_rank2","k_chain","k_same")): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in sorted(agg):
    print(k, {c: round(v/max(1,n[k][c])*( 1 if "SIZE" not in c else 1),1) for c,v in sorted(agg[k].items())})
PY
