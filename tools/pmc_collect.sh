#!/bin/bash
# rocprofv3 passes for the step kernels of one bench configuration (run on the GPU box through gpurun):
#   kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ counters, L2 hit / miss -- each counter set in its own pass (PMC passes carry
#   --kernel-trace only: gpurun refuses PMC together with the runtime / hip / hsa trace domains).
# usage: tools/pmc_collect.sh <tag> [bench.py args...]      -> gpurun_out/pmc_<tag>/{stats,fetch,write,sq,l2}
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $OLDPWD/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2 -- $B > $OUT/l2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/inst -- $B > $OUT/inst.log 2>&1 || true
find $OUT -name "*.csv" | head -40
