#!/bin/bash
# Round 3, first GPU call: the two new exactness tests, baseline lines of the classes that were never run at size
# (R Z B P at 20 MB), and the memory-fault hunt: bench steps under ZOPFLI_AMD_GUARD=1 over the SEG_CUTS sweep.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r03_base}
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tie_rule or guard_mode" > $OUT/pytest_new.log 2>&1
tail -5 $OUT/pytest_new.log
summ() {
python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    b=d["breakdown_s_per_step"]; c=d["chain_tasks_per_step"]
    print(d["config"]["workload"][:40], d["value"], "MB/s", d["ms_per_step"], "ms bitexact", d["bitexact_vs_reference"], "rt", d["roundtrip_ok"],
          {k: round(v*1e3,1) for k,v in b.items() if k in ("tables","squeeze","dp_kernel","trace_kernel","split","encode","cost_model","match_kernel","hash_kernels")},
          "accepted", round(c["accepted"]/max(c["tasks"],1),4), "pos_rerun", round(c["positions_rerun"]/max(c["positions"],1),4), c)
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
}
for cls in ${CLASSES:-R Z B P}; do
  timeout 600 python bench.py --cls $cls --size 20000000 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_${cls}.json 2> $OUT/bench_${cls}.err
  summ $OUT/bench_${cls}.json
  ZOPFLI_AMD_PROF=1 timeout 600 python bench.py --cls $cls --size 20000000 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_${cls}.json 2> $OUT/prof_${cls}.err
  grep -E "k_match2:|k_cutpoints" $OUT/prof_${cls}.err | head -3
  grep -E "squeeze prof|windows:" $OUT/prof_${cls}.err | sed -n '3,4p;29,30p'
done
# the fault of profiles/README.md (r02): SEG_CUTS 512 on 100 MB class T.  Under the guard, with the sweep.
for cuts in 64 512 1024; do
  ZOPFLI_AMD_GUARD=1 ZOPFLI_AMD_SEG_CUTS=$cuts timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/guard_T_cuts$cuts.json 2> $OUT/guard_T_cuts$cuts.err
  echo "guard T cuts=$cuts rc=$?"; summ $OUT/guard_T_cuts$cuts.json; tail -2 $OUT/guard_T_cuts$cuts.err
done
for cls in X M; do
  ZOPFLI_AMD_GUARD=1 timeout 900 python bench.py --cls $cls --steps 1 --warmup 0 --no-cpu-baseline > $OUT/guard_$cls.json 2> $OUT/guard_$cls.err
  echo "guard $cls rc=$?"; summ $OUT/guard_$cls.json; tail -2 $OUT/guard_$cls.err
done
