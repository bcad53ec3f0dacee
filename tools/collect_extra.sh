#!/bin/bash
# Runs on the GPU box (via gpurun): kernel stats of BASELINE configs[2] (block splitting on) and one
# PMC pass with the LDS counters of the default workload.  Outputs under gpurun_out/extra_run/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/extra_run
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_split1 -o r -- python $REPO/bench.py --blocksplitting 1 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats_split1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_lds -o p -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_lds.log 2>&1
tail -1 $OUT/stats_split1.log | cut -c1-200
tail -2 $OUT/pmc_lds.log | cut -c1-200
ls $OUT
