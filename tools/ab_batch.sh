# A/B of two builds of the library under tools/bin/batch_bench (one box, alternating)
set -e
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export LD_LIBRARY_PATH=$PWD/tools/bin/old; else unset LD_LIBRARY_PATH; fi
    timeout -k 10 200 tools/bin/batch_bench > gpurun_out/abb_$v$i.txt
  done
done
