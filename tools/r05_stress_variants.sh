#!/bin/bash
# the cooperative job's stress run (tools/r05_coop_stress.py) per compile-time variant (tools/build_variant.py)
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp zopfli_amd/libzopfli_amd.so /tmp/lib_default.so
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so; else cp tools/_build/libzopfli_amd_$v.so zopfli_amd/libzopfli_amd.so; fi
  echo "== variant $v: $(REPS=${REPS:-24} timeout 200 python tools/r05_coop_stress.py 2>&1 | tail -1)"
done
cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so
