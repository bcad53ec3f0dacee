#!/bin/bash
# what returning freed memory to the kernel costs with 64+ worker threads (TLB shootdowns): glibc knobs on/off
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {
  python bench.py --cls $1 --size $2 --blocksplitting $3 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/w.json 2>/dev/null
  python - "$4" $1 $3 <<'PY'
import json,sys
d=json.load(open("/tmp/w.json")); b=d["breakdown_s_per_step"]
print(sys.argv[1], sys.argv[2], "bs", sys.argv[3], d["value"], "MB/s", d["ms_per_step"], "ms", {k: round(v*1e3,1) for k,v in b.items() if k in ("split","encode","download","greedy","tables","squeeze")})
PY
}
for cfg in "R 50000000 0" "T 100000000 1" "T 100000000 0"; do set -- $cfg
  run $1 $2 $3 default
  MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=68719476736 MALLOC_TOP_PAD_=268435456 run $1 $2 $3 keep
done
