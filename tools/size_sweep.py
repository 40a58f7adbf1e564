#!/usr/bin/env python3
"""Size sweep (SURVEY.md section 8(d) sizes): GPU call latency / throughput for
device-resident, pageable and pinned input, next to Python's zlib.crc32 on one
core.  Results are checked against zlib.crc32.  The reference's own CPU loop
is timed by bench.py (`cpu_baseline`, ~0.62 GB/s, size-independent); the
crossover in DESIGN.md uses that figure.  One JSON document on stdout (progress on stderr)."""
import json
import os
import statistics as st
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import sift_amd

SIZES = [4, 64, 1 << 10, 4 << 10, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20,
         64 << 20, 256 << 20, 1 << 30]
rng = np.random.default_rng(0x51F7)
host = rng.integers(0, 256, size=max(SIZES), dtype=np.uint8)
dev = torch.from_numpy(host).cuda()
pinned = torch.from_numpy(host).pin_memory()
ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device="cuda")
out = torch.zeros(1, dtype=torch.int32, device="cuda")
rows = []
for n in SIZES:
    h, d = host[:n], dev[:n]
    hb = h.tobytes()
    want = zlib.crc32(hb) ^ 0xFFFFFFFF
    assert sift_amd.compute(0xFFFFFFFF, d) == want
    reps = 200 if n <= (16 << 20) else 30
    # synchronous device-resident call (launch + 4-byte read-back + sync)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sift_amd.compute(0xFFFFFFFF, d)
        t.append(time.perf_counter() - t0)
    sync_us = st.median(t) * 1e6
    # asynchronous back-to-back
    for _ in range(5):
        sift_amd.compute_async(0xFFFFFFFF, d, out, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sift_amd.compute_async(0xFFFFFFFF, d, out, ws)
    torch.cuda.synchronize()
    async_us = (time.perf_counter() - t0) / reps * 1e6
    # host-resident input (PCIe inclusive)
    hreps = 20 if n <= (16 << 20) else 3
    assert sift_amd.compute(0xFFFFFFFF, h) == want
    t = []
    for _ in range(hreps):
        t0 = time.perf_counter()
        sift_amd.compute(0xFFFFFFFF, h)
        t.append(time.perf_counter() - t0)
    host_us = st.median(t) * 1e6
    # pinned host input: streamed in place over PCIe
    hp = pinned[:n].numpy()
    assert sift_amd.compute(0xFFFFFFFF, hp) == want
    t = []
    for _ in range(hreps):
        t0 = time.perf_counter()
        sift_amd.compute(0xFFFFFFFF, hp)
        t.append(time.perf_counter() - t0)
    pin_us = st.median(t) * 1e6
    # CPU: zlib.crc32 on one core
    creps = max(1, min(200, int(2e8 / max(n, 1)) // 50))
    t0 = time.perf_counter()
    for _ in range(creps):
        zlib.crc32(hb)
    z_us = (time.perf_counter() - t0) / creps * 1e6
    rows.append({"bytes": n, "gpu_sync_us": round(sync_us, 2), "gpu_async_us": round(async_us, 2),
                 "gpu_async_GBps": round(n / async_us / 1e3, 3), "gpu_host_input_us": round(host_us, 2),
                 "gpu_host_input_GBps": round(n / host_us / 1e3, 3),
                 "gpu_pinned_input_us": round(pin_us, 2), "gpu_pinned_input_GBps": round(n / pin_us / 1e3, 3),
                 "zlib_us": round(z_us, 2)})
    print(rows[-1], file=sys.stderr, flush=True)
json.dump({"note": "device-resident unless 'host_input' / 'pinned_input'; zlib = Python zlib.crc32, 1 core; the "
                   "reference's own loop runs ~0.62 GB/s (bench.py cpu_baseline)", "rows": rows}, sys.stdout, indent=1)
print()
