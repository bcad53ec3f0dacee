#!/bin/bash
# k_match2 tile size A/B: the library built with MT = 4096 / 8192 (tests/_build/ab/lib_mt*.so) against the default 2048
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
cp zopfli_amd/libzopfli_amd.so /tmp/lib_default.so
for v in default mt4096 mt8192; do
  if [ $v != default ]; then cp tests/_build/ab/lib_$v.so zopfli_amd/libzopfli_amd.so; else cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so; fi
  for cls in T X; do
    python bench.py --cls $cls --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown_s_per_step']
print('$v $cls', d['value'],'MB/s', d['ms_per_step'], 'ms match', round(b['match_kernel']*1e3,1), 'tables', round(b['tables']*1e3,1), 'bitexact', d['bitexact_vs_reference'])"
  done
done
cp /tmp/lib_default.so zopfli_amd/libzopfli_amd.so
