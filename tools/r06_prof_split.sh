#!/bin/bash
# round 6: rocprofv3 kernel stats of a ZopfliCompress call with the reference's default options on random data (the block-split
# search's block sizes on the device: k_block_cost, k_cost_*)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r06_prof_split}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $REPO/bench.py --cls R --blocksplitting 1 --steps 1 --warmup 0 --no-cpu-baseline --no-small-files --entry zopfli_compress > $OUT/stats.log 2>&1
python - $OUT/stats/r_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3 or "cost" in r["Name"]: print(f'{r["Name"][:60]:60s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms  total {float(r["TotalDurationNs"])/1e6:8.2f} ms {r["Percentage"]:>6s}%')
PY
