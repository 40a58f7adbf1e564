#!/bin/bash
# Mid-size single buffers: who owns the per-call floor, host or GPU?
#   gpurun --timeout 600 -- 'bash tools/collect_midsize.sh r03'
# then python tools/start_to_start.py r03
set -e
R=$PWD
RND=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 120 tools/bin/batch_bench compute > gpurun_out/${RND}_mid_plain.json
timeout -k 10 60 tools/bin/api_cost > gpurun_out/${RND}_api_cost.txt
cd /tmp
for size in 1048576 4194304 16777216 67108864; do
  rm -rf $R/gpurun_out/${RND}_mid_$size
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${RND}_mid_$size -- $R/tools/bin/batch_bench compute $size > $R/gpurun_out/${RND}_mid_$size.json 2> $R/gpurun_out/${RND}_mid_$size.log
done
echo collected
