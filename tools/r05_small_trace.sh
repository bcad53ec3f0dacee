#!/bin/bash
# round 5: per-kernel durations of small ZopfliCompress calls (1 MB and 64 KiB of text, n = 15, default options) —
# rocprofv3 --kernel-trace --stats over tools/latency.py's loop
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r05_small}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/small.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
from zopfli_amd import ZopfliOptions, api, generate
size = int(sys.argv[2])
lib = api.library()
data = generate("T", size)
opt = ZopfliOptions(15)
api.compress(data, 0, opt, lib=lib)
t0 = time.perf_counter()
for _ in range(10):
    api.compress(data, 0, opt, lib=lib)
print("ms per call", (time.perf_counter() - t0) * 100)
PY
for size in 1000000 65536; do
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$size -o r -- python /tmp/small.py $REPO $size > $OUT/stats_$size.log 2>&1
  echo "== $size bytes: $(grep -a 'ms per call' $OUT/stats_$size.log)"
  python - $OUT/stats_$size/r_kernel_stats.csv <<'PY'
import csv,sys
tot=0
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: tot+=float(r["TotalDurationNs"])
print("  kernel time per call %.2f ms (11 calls)" % (tot/11/1e6))
for r in rows:
    if float(r["Percentage"]) > 1.5: print(f'  {r["Name"][:56]:56s} calls/call {int(r["Calls"])/11:6.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  per call {float(r["TotalDurationNs"])/11/1e6:6.2f} ms {r["Percentage"]:>6s}%')
PY
done
