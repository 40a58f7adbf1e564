#!/bin/bash
# Collects one round's measurements on the GPU box (run through gpurun from the
# repo root: `gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r03'`), then
# `python tools/summarize_profiles.py r03` here condenses them into profiles/.
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
R=$PWD
RND=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${RND}_trace gpurun_out/${RND}_trace20 gpurun_out/${RND}_pmc_fetch gpurun_out/${RND}_pmc_write gpurun_out/${RND}_pmc_sq gpurun_out/${RND}_batch_trace
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${RND}.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${RND}_trace -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $R/gpurun_out/${RND}_trace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${RND}_trace20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${RND}_trace20.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${RND}_pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/${RND}_pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${RND}_pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/${RND}_pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${RND}_pmc_sq -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/${RND}_pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${RND}_batch_trace -- $R/tools/bin/batch_bench > $R/gpurun_out/${RND}_batch_bench_traced.json 2> $R/gpurun_out/${RND}_batch_trace.log
cd $R
timeout -k 10 300 tools/bin/batch_bench > gpurun_out/${RND}_batch_bench.json
timeout -k 10 400 python tools/size_sweep.py > gpurun_out/${RND}_size_sweep.json 2> gpurun_out/${RND}_size_sweep.log
timeout -k 10 200 python tools/async_sweep.py > gpurun_out/${RND}_async_sweep.json
SIFT_AMD_LIB=$R/tools/bin/trace/libsift_amd.so timeout -k 10 120 python tools/trace_groups.py > gpurun_out/${RND}_group_trace.txt
bash tools/collect_midsize.sh ${RND}
bash tools/cut_grid.sh > gpurun_out/${RND}_cut_grid.json
timeout -k 10 300 python tools/lone_long_item.py > gpurun_out/${RND}_lone_long_item.txt
bash tools/collect_batch_pmc.sh ${RND}
echo collected
