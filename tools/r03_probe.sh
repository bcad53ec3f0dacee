#!/bin/bash
# seg_probe.py (chained squeeze runs against the oracle) on chosen classes under a few environments
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for envs in "${@}"; do
  echo "== $envs"
  env $envs SEG_PROBE_CASES=${CASES:-Z} timeout 600 python tests/seg_probe.py 2>&1 | tail -3
done
