#!/bin/bash
# class Z 20 MB: the chain under a few settings (redo passes, merge on/off), non-PROF timing + one PROF line each
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for envs in "$@"; do
  echo "== $envs"
  env $envs timeout 600 python bench.py --cls ${CLS:-Z} --size ${SIZE:-20000000} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown_s_per_step']; c=d['chain_tasks_per_step']
print(d['value'],'MB/s dp_kernel',round(b['dp_kernel']*1e3,1),'ms; tasks',c['tasks'],'accepted',c['accepted'],'pos_rerun',c['positions_rerun'])"
  env $envs ZOPFLI_AMD_PROF=1 timeout 600 python bench.py --cls ${CLS:-Z} --size ${SIZE:-20000000} --steps 1 --warmup 0 --numiterations 2 --no-cpu-baseline 2>&1 | grep -E "longest task|windows:" | tail -2
done
