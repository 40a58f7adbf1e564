# Where a batch of equal items is better cut into pieces and where kept by waves:
# device time per call of sift_crc32_batch_device for sizes x counts, both ways.
for sz in 65536 131072 262144 524288 1048576 4194304; do
  for cnt in 128 512 1024 2048 4000; do
    tot=$((sz*cnt)); if [ $tot -gt 1073741824 ]; then continue; fi
    a=$(SIFT_BENCH_WAVE_MAX=1073741824 timeout -k 10 60 tools/bin/batch_bench items $sz $cnt | grep -o '"device_us_per_call": [0-9.]*' | cut -d' ' -f2) || exit 1
    b=$(SIFT_BENCH_WAVE_MAX=4096 timeout -k 10 60 tools/bin/batch_bench items $sz $cnt | grep -o '"device_us_per_call": [0-9.]*' | cut -d' ' -f2) || exit 1
    c=$(timeout -k 10 60 tools/bin/batch_bench items $sz $cnt | grep -o '"device_us_per_call": [0-9.]*' | cut -d' ' -f2) || exit 1
    echo "bytes $sz count $cnt waves_us $a pieces_us $b default_us $c floor_us $(python3 -c "print(round($tot/6.9e6+4,1))")"
  done
done
