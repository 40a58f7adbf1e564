# Where a batch of equal items is better cut into pieces and where kept by waves: device time
# per call of sift_crc32_batch_device (descriptors in HBM) for sizes x counts, with every item
# kept by a wave, every item cut, and the library's own choice; floor = bytes / 6.9 TB/s + 4 us.
# JSON on stdout.
t() { grep -o '"kind": "crc_batch"[^}]*' | grep -o '"device_us_per_call": [0-9.]*' | cut -d' ' -f2; }
echo '{"rows": ['
sep=""
for sz in 65536 131072 262144 524288 1048576 4194304; do
  for cnt in 128 512 1024 2048 4000; do
    tot=$((sz*cnt)); if [ $tot -gt 1073741824 ]; then continue; fi
    a=$(SIFT_BENCH_WAVE_MAX=1073741824 timeout -k 10 60 tools/bin/batch_bench items $sz $cnt | t) || exit 1
    b=$(SIFT_BENCH_WAVE_MAX=4096 timeout -k 10 60 tools/bin/batch_bench items $sz $cnt | t) || exit 1
    c=$(timeout -k 10 60 tools/bin/batch_bench items $sz $cnt | t) || exit 1
    echo "$sep {\"item_bytes\": $sz, \"items\": $cnt, \"waves_us\": $a, \"pieces_us\": $b, \"default_us\": $c, \"floor_us\": $(python3 -c "print(round($tot/6.9e6+4,1))")}"
    sep=","
  done
done
echo ']}'
