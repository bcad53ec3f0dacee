#!/bin/bash
# the chain on long-run data: parity (chain tests incl. class Z), then bench lines Z / M / B / P / T with PROF for Z
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r03_zchain}
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTS:-squeeze_runs or chain_task_paths or tie_rule or match_table}" > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log | cut -c1-600
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]; c=d["chain_tasks_per_step"]
    print(d["config"]["workload"][:26], d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"],
          {k: round(v*1e3,1) for k,v in b.items() if k in ("tables","squeeze","dp_kernel","trace_kernel","match_kernel","hash_kernels","cost_model")},
          "tasks", c["tasks"], "accepted", round(c["accepted"]/max(c["tasks"],1),4), "pos_rerun", round(c["positions_rerun"]/max(c["positions"],1),4))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
for spec in ${SPECS:-Z:20000000 M:100000000 B:20000000 P:20000000 T:100000000}; do
  cls=${spec%%:*}; size=${spec##*:}
  timeout 600 python bench.py --cls $cls --size $size --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_${cls}.json 2> $OUT/bench_${cls}.err
  summ $OUT/bench_${cls}.json
done
ZOPFLI_AMD_PROF=1 timeout 600 python bench.py --cls Z --size 20000000 --steps 1 --warmup 0 --numiterations 3 --no-cpu-baseline > $OUT/prof_Z.json 2> $OUT/prof_Z.err
grep -E "k_cutpoints|squeeze prof|windows:|longest task" $OUT/prof_Z.err | head -12
