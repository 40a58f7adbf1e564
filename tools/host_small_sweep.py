#!/usr/bin/env python3
"""Pageable host input, 64 KiB .. 16 MiB: microseconds per synchronous call (median of 30)."""
import os, sys, time, zlib, statistics as st
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sift_amd
host = np.random.default_rng(0x51F7).integers(0, 256, size=16 << 20, dtype=np.uint8)
row = {}
for n in (64 << 10, 128 << 10, 256 << 10, 384 << 10, 512 << 10, 768 << 10, 1 << 20, 1536 << 10, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 16 << 20):
    h = host[5:5 + n]
    assert sift_amd.compute(0xFFFFFFFF, h) == zlib.crc32(h.tobytes()) ^ 0xFFFFFFFF
    t = []
    for _ in range(30):
        t0 = time.perf_counter(); sift_amd.compute(0xFFFFFFFF, h); t.append(time.perf_counter() - t0)
    row[f"{n >> 10}KiB_us"] = round(st.median(t) * 1e6, 1)
print(row, flush=True)
