# Per-kernel medians (rocprofv3 kernel trace) of `batch_bench items <bytes> [count]` for library
# builds under tools/bin/<name>/ ("new" = the tree's own): tools/ab_kernels.sh "prev new" 1048576 [count]
variants="$1"; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in $variants; do
  if [ $v = new ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$R/tools/bin/$v; fi
  rm -rf $R/gpurun_out/kt_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$v -- $R/tools/bin/batch_bench items "$@" > $R/gpurun_out/kt_$v.txt 2>&1 || exit 1
  echo "== $v items $@"; grep -h crc_batch $R/gpurun_out/kt_$v.txt | cut -c1-200
  python3 $R/tools/kernel_medians.py $R/gpurun_out/kt_$v | grep sift::
done
