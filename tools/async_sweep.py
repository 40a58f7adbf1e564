#!/usr/bin/env python3
"""Asynchronous device-resident compute() per size: back-to-back calls over distinct
slices of a 2 GiB pool (more than the Infinity Cache holds per round), wall time per call."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sift_amd
pool = torch.randint(0, 256, (2 << 30,), dtype=torch.uint8, device="cuda")
ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device="cuda")
out = torch.zeros(1, dtype=torch.int32, device="cuda")
rows = {}
for size in (1 << 20, 2 << 20, 4 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 1 << 30):
    ns = max(2, min(pool.numel() // size, max(8, (512 << 20) // size)))
    stride = (pool.numel() // ns) // 4096 * 4096
    sl = [pool[k * stride:k * stride + size] for k in range(ns)]
    calls = 2000 if size <= (16 << 20) else 300
    best = 1e9
    for rep in range(3):
        for k in range(20):
            sift_amd.compute_async(0xFFFFFFFF, sl[k % ns], out, ws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(calls):
            sift_amd.compute_async(0xFFFFFFFF, sl[k % ns], out, ws)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / calls)
    rows[f"{size >> 20}MiB_us"] = round(best * 1e6, 2)
print(json.dumps(rows), flush=True)
