#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
run() {
  env "$@" timeout 300 python bench.py --cls ${CLS:-T} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['breakdown_s_per_step']; c=d['chain_tasks_per_step']
print('$*', d['value'], d['ms_per_step'], d['bitexact_vs_reference'], 'dp', b['dp_kernel'], 'tasks', c['tasks'], 'rerun', c['rerun_state'], c['rerun_values'], c['rerun_level'], 'pos', c['positions_rerun'])
"
}
while IFS= read -r line; do run $line; done
