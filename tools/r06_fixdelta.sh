#!/bin/bash
# round 6: what shift do the tasks that k_dp4_fix re-runs on long-run data need?  (ZOPFLI_AMD_SEG_DEBUG=1 prints every task's check)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/${TAG:-r06_fixdelta}
mkdir -p $OUT
for cls in ${CLASSES:-Z}; do
  ZOPFLI_AMD_SEG_DEBUG=1 timeout -k 5 300 python bench.py --cls $cls --size ${SIZE:-10000000} --numiterations 4 --steps 1 --warmup 0 --no-cpu-baseline --entry resident --no-blocksplitting1 --no-small-files 2>/dev/null | grep "^fix b" | grep -v " ok 1 " > $OUT/$cls.fix.txt
  wc -l $OUT/$cls.fix.txt
  python - $OUT/$cls.fix.txt <<'PY'
import sys, re, struct, collections
h = collections.Counter(); n = 0
for l in open(sys.argv[1]):
    m = re.search(r"match (\d+) d (\S+) delta (\S+) vmin (\S+) vmax (\S+) why (\d+)", l)
    if not m: continue
    match, d, delta, vmin, vmax, why = int(m[1]), float(m[2]), float(m[3]), float(m[4]), float(m[5]), int(m[6])
    if match != 1 or why != 1: h[("other", match, why)] += 1; continue
    # ulp of a float at vmax
    b = struct.unpack("<I", struct.pack("<f", vmax))[0]
    ulp = struct.unpack("<f", struct.pack("<I", (b & 0x7f800000)))[0] * 2.0 ** -23
    k = delta / ulp
    h[("ulps", round(k, 2) if abs(k) < 20 else ("big+" if k > 0 else "big-"))] += 1
    n += 1
for k, v in sorted(h.items(), key=lambda kv: -kv[1])[:40]: print(k, v)
print("level failures", n)
PY
done
