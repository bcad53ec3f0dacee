#!/bin/bash
# round 6: k_match2 variants (tools/build_variant.py, ZOPFLI_AMD_LIB) on classes T and X, 100 MB, resident, match kernel ms
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_match_ab}
mkdir -p $OUT
for lib in ${LIBS:-default m2t256 m2t128}; do
  for cls in ${CLASSES:-T X}; do
    if [ $lib = default ]; then unset ZOPFLI_AMD_LIB; else export ZOPFLI_AMD_LIB=$REPO/tools/_build/libzopfli_amd_$lib.so; fi
    ZOPFLI_AMD_PROF_MATCH=${PROFM:-0} timeout -k 5 200 python bench.py --cls $cls --steps 2 --warmup 1 --no-cpu-baseline --entry resident --no-blocksplitting1 2>$OUT/${lib}_$cls.err | grep '^{"metric"' > $OUT/${lib}_$cls.json
    python - $OUT/${lib}_$cls.json "$lib $cls" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); m=d["roofline_match"]; b=d["breakdown_s_per_step"]
    print(f'{sys.argv[2]}: {d["value"]} MB/s, match kernel {m["seconds_per_step"]*1e3:.2f} ms, hash {m["hash_kernels_seconds_per_step"]*1e3:.2f} ms, tables {b.get("tables")}, bitexact {d["bitexact_vs_reference"]}')
except Exception as e: print("ERR", sys.argv[2], e)
PY
  done
done
