#!/bin/bash
# round 5: data with long runs stays on ONE context without block splitting (round 3: two contexts 61 -> 35 MB/s on class Z).
# With shards of equal cost and stream priorities, is that still right?  ZOPFLI_AMD_STREAM_PRIO=2 uses the priorities
# without block splitting too (and lets data with < 30 % run probes be dealt); ZOPFLI_AMD_SPLIT_RUNS=1 deals whatever the data.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {
  timeout 300 python bench.py --cls $1 --steps 2 --warmup 1 --no-cpu-baseline --entry zopfli_compress --no-blocksplitting1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  class $1 $2: bs0', j['value'], 'MB/s', j['ms_per_step'], 'ms exact', j['bitexact_vs_reference'])"
}
for cls in ${CLASSES:-M Z}; do
  run $cls "one context (default)"
  ZOPFLI_AMD_SPLIT_RUNS=1 run $cls "three contexts, no priorities"
  ZOPFLI_AMD_SPLIT_RUNS=1 ZOPFLI_AMD_STREAM_PRIO=2 run $cls "three contexts, three priorities"
  ZOPFLI_AMD_SPLIT_RUNS=1 ZOPFLI_AMD_STREAM_PRIO=2 ZOPFLI_AMD_SPLIT_WAYS=2 run $cls "two contexts, two priorities"
done
