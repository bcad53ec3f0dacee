#!/bin/bash
# k_bucket + k_match3 against k_chain + k_match2: parity tests, then bench lines (T X 100 MB, P B Z 20 MB) with both.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r03_match3}
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTS:-match_table or hash_links or pool_overflow or greedy or guard_mode}" > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log | cut -c1-400
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]; c=d["chain_tasks_per_step"]
    print(d["config"]["workload"][:26], d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"],
          {k: round(v*1e3,1) for k,v in b.items() if k in ("tables","squeeze","dp_kernel","match_kernel","hash_kernels")})
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
for m in ${MODES:-4 2}; do
  for spec in ${SPECS:-T:100000000 X:100000000 P:20000000 B:20000000 Z:20000000}; do
    cls=${spec%%:*}; size=${spec##*:}
    ZOPFLI_AMD_MATCH=$m timeout 600 python bench.py --cls $cls --size $size --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_m${m}_${cls}.json 2> $OUT/bench_m${m}_${cls}.err
    echo -n "match=$m "; summ $OUT/bench_m${m}_${cls}.json
    ZOPFLI_AMD_PROF=1 ZOPFLI_AMD_MATCH=$m timeout 600 python bench.py --cls $cls --size $size --steps 1 --warmup 0 --numiterations 1 --no-cpu-baseline > $OUT/prof_m${m}_${cls}.json 2> $OUT/prof_m${m}_${cls}.err
    grep -E "k_match[23]:" $OUT/prof_m${m}_${cls}.err | head -2
  done
done
