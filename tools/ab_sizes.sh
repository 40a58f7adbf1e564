# A/B of two builds over call sizes (back-to-back async calls, us per call), one box, alternating
set -e
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export SIFT_AMD_LIB=$PWD/tools/bin/old/libsift_amd.so; else unset SIFT_AMD_LIB; fi
    echo -n "$v $i "; timeout -k 10 200 python tools/async_sweep.py
  done
done
