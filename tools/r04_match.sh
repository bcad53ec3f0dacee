#!/bin/bash
# round 4: k_match5 / auto dispatch: parity (digests) and timing against k_match2
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r04_m5}
mkdir -p $OUT
timeout -k 5 200 python tools/r04_match5.py parity > $OUT/parity.log 2>&1; echo "parity rc $?" >> $OUT/parity.log
tail -14 $OUT/parity.log
timeout -k 5 200 python tools/r04_match5.py time ${SPECS:-} > $OUT/time.log 2>&1; echo "time rc $?" >> $OUT/time.log
tail -9 $OUT/time.log | cut -c1-420
