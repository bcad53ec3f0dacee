#!/bin/bash
# round 4: kernel timeline of ZopfliCompress calls (rocprofv3 --kernel-trace): when is the device idle?
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG:-r04_timeline}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for bs in ${BS:-1 0}; do
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/bs$bs -o t -- python $REPO/tools/r04_cpu.py ${CLS:-T} ${SIZE:-100000000} $bs 2 > $OUT/bs$bs.log 2>&1
  grep '^{' $OUT/bs$bs.log | cut -c1-600
  python $REPO/tools/r04_timeline.py $OUT/bs$bs/t_kernel_trace.csv | tee $OUT/bs$bs.txt
done
