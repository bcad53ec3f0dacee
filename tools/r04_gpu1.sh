#!/bin/bash
# round 4, GPU call 1: k_match5 parity (digests) and timing against k_match2
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r04_m5b}
mkdir -p $OUT
timeout 900 python tools/r04_match5.py parity > $OUT/parity.log 2>&1; echo "parity rc $?" >> $OUT/parity.log
tail -20 $OUT/parity.log
timeout 900 python tools/r04_match5.py time > $OUT/time.log 2>&1; echo "time rc $?" >> $OUT/time.log
tail -12 $OUT/time.log
ZOPFLI_AMD_PROF=1 timeout 600 python tools/r04_match5.py time T:20000000 X:20000000 P:20000000 B:20000000 M:20000000 > $OUT/prof.log 2>&1
grep -E "k_match|rc" $OUT/prof.log | tail -24
