#!/bin/bash
# round 5, first call: the GPU suite on the round's first commit (new: PNG goldens at size, adversarial oracle test at every
# position), the run-task baseline (classes Z and M with ZOPFLI_AMD_PROF), the reference's zopflipng timed on THIS box's
# host beside libzopflipng_amd.so, and the bounded repro of round 4's k_match5 hang (last: it may hang)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r05_call1
mkdir -p $OUT
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; grep -a "passed\|failed\|error" $OUT/suite.log | tail -3
for cls in Z M; do
  timeout -k 5 120 python bench.py --cls $cls --steps 1 --warmup 1 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/bench_$cls.json 2> $OUT/bench_$cls.err
  cut -c1-230 $OUT/bench_$cls.json
done
ZOPFLI_AMD_PROF=1 timeout -k 5 120 python bench.py --cls Z --size 20000000 --steps 1 --warmup 0 --no-cpu-baseline --no-blocksplitting1 --entry resident > $OUT/prof_Z.json 2> $OUT/prof_Z.err
grep -a "k_dp5_spec generic windows, cycles:\|k_dp5_spec positions\|squeeze prof\|longest task" $OUT/prof_Z.err | tail -4 | cut -c1-400
TMPDIR=/tmp timeout -k 5 400 python tools/png_at_size.py 4096 --ref > $OUT/png4096_same_host.json 2> $OUT/png4096.err; cat $OUT/png4096_same_host.json
timeout -k 5 90 python tools/m5_atomic_repro.py --run > $OUT/m5_atomic.log 2>&1; echo "m5 repro rc $?"; tail -4 $OUT/m5_atomic.log
