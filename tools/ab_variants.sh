# A/B of library builds under tools/bin/<name>/ (plus "new" = the tree's own) with
# tools/bin/batch_bench: tools/ab_variants.sh "old v1 new" [items <bytes>]...
set -e
variants="$1"; shift
for i in 1 2; do
  for v in $variants; do
    if [ $v = new ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$PWD/tools/bin/$v; fi
    timeout -k 10 200 tools/bin/batch_bench "$@" > gpurun_out/abb_$v$i.txt
  done
done
