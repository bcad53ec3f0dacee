#!/bin/bash
# round 5: with block splitting the three contexts of a call run at three stream priorities, so they end one after the
# other — does giving the first more and the last fewer master blocks shorten the call?  (ZOPFLI_AMD_SHARD_WEIGHTS)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cls in ${CLASSES:-R T P M}; do
  for W in ${WEIGHTS:-default 38,33,29 43,33,24 48,32,20}; do
    if [ $W = default ]; then unset ZOPFLI_AMD_SHARD_WEIGHTS; else export ZOPFLI_AMD_SHARD_WEIGHTS=$W; fi
    timeout 200 python bench.py --cls $cls --steps 2 --warmup 1 --no-cpu-baseline --entry zopfli_compress 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j.get('blocksplitting1',{})
print('  class $cls weights $W: bs0', j['value'], 'MB/s', j['ms_per_step'], 'ms | bs1', b.get('value'), 'MB/s', b.get('ms_per_step'), 'ms exact', j['bitexact_vs_reference'], b.get('bitexact_vs_reference'))"
  done
done
