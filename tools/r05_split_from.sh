#!/bin/bash
# round 5: from how many master blocks on should a call be dealt over three contexts?  (ZOPFLI_AMD_SPLIT_MB; 32 until now)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {
  timeout 300 python bench.py --cls $1 --steps 3 --warmup 1 --no-cpu-baseline --entry zopfli_compress --size $2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j.get('blocksplitting1',{})
print('  class $1 $2 bytes SPLIT_MB=$3: bs0', j['ms_per_step'], 'ms | bs1', b.get('ms_per_step'), 'ms')"
}
for cls in ${CLASSES:-T P}; do
  for sz in ${SIZES:-2000000 3000000 4000000 6000000 8000000 12000000}; do
    for s in ${FROMS:-32 2 4}; do ZOPFLI_AMD_SPLIT_MB=$s run $cls $sz $s; done
  done
done
