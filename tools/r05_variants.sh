#!/bin/bash
# A/B of compile-time variants of the library (tools/build_variant.py): the run-path parity tests and class Z per variant
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r05_variants}
mkdir -p $OUT
cd $REPO
cp zopfli_amd/libzopfli_amd.so /tmp/lib_default.so
for v in ${VARIANTS:-default A C D}; do
  if [ $v = default ]; then cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so; else cp tools/_build/libzopfli_amd_$v.so zopfli_amd/libzopfli_amd.so; fi
  cp zopfli_amd/libzopfli_amd.so zopfli_amd/libzopfli.so.1
  timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "squeeze_runs or chain_task or run_paths" > $OUT/parity_$v.log 2>&1
  echo "variant $v: $(grep -a 'passed\|failed' $OUT/parity_$v.log | tail -1)"
  grep -a "^FAILED" $OUT/parity_$v.log | head -5
  timeout -k 5 120 python bench.py --cls Z --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/bench_Z_$v.json 2> $OUT/bench_Z_$v.err
  python - $OUT/bench_Z_$v.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print("  variant", sys.argv[2], "class Z MB/s", d["value"], "bitexact", d["bitexact_vs_reference"], "chain ms/run", r["avg_launch_ms"])
except Exception as e: print("  ERR", sys.argv[1], e)
PY
done
cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so
