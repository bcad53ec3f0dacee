#!/bin/bash
# round 5: the host pools A/B on one box — r04 (a wide pool of 128 threads per shard lane), lanes3 (a third each), default
# (ONE wide pool, any number of jobs): ZopfliCompress with the reference's default block splitting, classes R / T / P
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r05_pool_ab}
mkdir -p $OUT
cp zopfli_amd/libzopfli_amd.so /tmp/lib_default.so
for rep in 1 2; do
for v in ${VARIANTS:-default pool_r04 pool_lanes3}; do
  if [ $v = default ]; then cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so; else cp tools/_build/libzopfli_amd_$v.so zopfli_amd/libzopfli_amd.so; fi
  for cls in ${CLASSES:-R T P}; do
    timeout -k 5 120 python bench.py --cls $cls --blocksplitting 1 --steps 3 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --entry zopfli_compress > $OUT/b_${v}_${cls}_$rep.json 2> $OUT/b_${v}_${cls}_$rep.err
    echo "rep $rep $v class $cls bs1: $(grep -o '"value": [0-9.]*' $OUT/b_${v}_${cls}_$rep.json | head -1)"
  done
done
done
cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so
