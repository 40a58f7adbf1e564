#!/bin/bash
# SQ counters of the wave kernel on 1 GiB of items of one size each (device descriptors):
#   gpurun --timeout 900 -- 'bash tools/collect_batch_pmc.sh r02'
# then python tools/summarize_batch_pmc.py r02
set -e
R=$PWD
RND=${1:-r03}
export TMPDIR=/tmp
cd /tmp
for size in 64 100 256 512 1024 3000 4096 8192 16384; do
  rm -rf $R/gpurun_out/${RND}_batch_pmc_$size
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --output-format csv -d $R/gpurun_out/${RND}_batch_pmc_$size -- $R/tools/bin/batch_bench items $size > $R/gpurun_out/${RND}_batch_pmc_$size.json 2> $R/gpurun_out/${RND}_batch_pmc_$size.log
done
echo collected
