#!/bin/bash
# round 5: ranks only where the 8192-hit cap can bind (k_hits maxima -> k_rank2): parity of the three match paths
# (digests, adversarial suite) and the match / hash kernel times per class
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r05_match}
mkdir -p $OUT
timeout -k 5 200 python tools/r04_match5.py parity > $OUT/parity.log 2>&1; echo "parity rc $?" >> $OUT/parity.log
tail -14 $OUT/parity.log
timeout -k 10 400 python -m pytest tests/test_gpu_match_adversarial.py tests/test_gpu_parity.py -m gpu -x -q -k "match" > $OUT/tests.log 2>&1; grep -a "passed\|failed" $OUT/tests.log | tail -2
timeout -k 5 200 python tools/r04_match5.py time ${SPECS:-T:100000000 X:100000000 P:20000000 B:20000000 M:20000000} > $OUT/time.log 2>&1; echo "time rc $?" >> $OUT/time.log
tail -12 $OUT/time.log | cut -c1-420
ZOPFLI_AMD_RANK_ALL=1 timeout -k 5 200 python tools/r04_match5.py time T:100000000 > $OUT/time_rankall.log 2>&1
tail -3 $OUT/time_rankall.log | cut -c1-420
