# A/B of two builds of the library on one box, alternating: tools/bin/old/libsift_amd.so
# (built from the commit to compare with) against the one in the tree.
set -e
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export SIFT_AMD_LIB=$PWD/tools/bin/old/libsift_amd.so; else unset SIFT_AMD_LIB; fi
    timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/ab_$v$i.json
    python - <<PY
import json; d=json.load(open("gpurun_out/ab_$v$i.json")); r=d["roofline"]; print("$v", $i, r["kernel_us"], r["kernel_us_min"], r["kernel_us_median"], r["kernel_us_max"], d["value"])
PY
  done
done
