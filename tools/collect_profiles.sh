#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace/stats + PMC passes of the default bench (or BENCH_ARGS).
# Outputs under gpurun_out/$TAG; copy the summaries to profiles/ afterwards.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-profile_run}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 1 --warmup 0 --no-cpu-baseline}"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
grep '^{"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line.json
i=0
# (EXTRA_SET: one more pass, e.g. the LDS set "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS")
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "${EXTRA_SET:-}"; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_$i.log 2>&1
done
python $REPO/tools/pmc_summary.py $OUT/pmc_*/p_counter_collection.csv > $OUT/pmc.json
cut -c1-300 $OUT/bench_line.json
python - $OUT/stats/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5: print(f'{r["Name"][:50]:50s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  {r["Percentage"]:>6s}%')
PY
python - $OUT/pmc.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("chain","k_dp5_spec","k_dp4_fix","k_match2","k_chain","k_codes"):
    v=d.get(k,{})
    print(k, {c: (round(x/1e9,3) if "bytes" in c else round(x)) for c,x in v.items() if c in ("launches","fetch_bytes","write_bytes","hbm_bytes","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_ACTIVE_INST_ANY","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_ACTIVE_INST_LDS")})
PY
