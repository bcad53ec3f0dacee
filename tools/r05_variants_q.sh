#!/bin/bash
# quick parity per variant (one test)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r05_variants_q}
mkdir -p $OUT
cd $REPO
cp zopfli_amd/libzopfli_amd.so /tmp/lib_default.so
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so; else cp tools/_build/libzopfli_amd_$v.so zopfli_amd/libzopfli_amd.so; fi
  timeout -k 10 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "squeeze_runs and Z" > $OUT/parity_$v.log 2>&1
  echo "variant $v: $(grep -a 'passed\|failed' $OUT/parity_$v.log | tail -1) $(grep -a 'first diff at' $OUT/parity_$v.log | grep AssertionError | head -1)"
done
cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so
