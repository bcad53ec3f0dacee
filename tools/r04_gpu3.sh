#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${TAG:-r04_m5prof2}
mkdir -p $OUT
ZOPFLI_AMD_PROF=1 timeout -k 5 60 python tools/r04_match5.py time ${SPECS:-T:300000} > $OUT/prof.log 2>&1; echo "prof rc $?" >> $OUT/prof.log
grep -E "k_match5|rc|Error|error|did not end" $OUT/prof.log | tail -8 | cut -c1-400
