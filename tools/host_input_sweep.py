#!/usr/bin/env python3
"""Pageable host input (sift_crc32_compute_host) by number of copy threads.
usage: host_input_sweep.py <threads>   (the knob is read once per device, so one process per value)"""
import os, sys, time, zlib, statistics as st
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sift_amd
from sift_amd import _lib
threads = int(sys.argv[1])
_lib.check(_lib.lib.sift_tune(_lib.TUNE_HOST_COPY_THREADS, threads))
host = np.random.default_rng(0x51F7).integers(0, 256, size=1 << 30, dtype=np.uint8)
row = {"threads": threads}
for n in (4 << 20, 16 << 20, 64 << 20, 256 << 20, 1 << 30):
    h = host[:n]
    got = sift_amd.compute(0xFFFFFFFF, h)
    if n <= (64 << 20):
        assert got == zlib.crc32(h.tobytes()) ^ 0xFFFFFFFF
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); sift_amd.compute(0xFFFFFFFF, h); t.append(time.perf_counter() - t0)
    row[f"{n >> 20}MiB_GBps"] = round(n / st.median(t) / 1e9, 1)
print(row, flush=True)
