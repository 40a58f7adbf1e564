#!/usr/bin/env python3
"""Throughput of the batch entry points (one wave per buffer) over chunk sizes;
results spot-checked against zlib.crc32.  Writes gpurun_out/batch_sweep.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import sift_amd
from sift_amd import container
import zlib

TOTAL = 1 << 30
blob = torch.randint(0, 256, (TOTAL,), dtype=torch.uint8, device="cuda")
host = blob[: 64 << 20].cpu().numpy()
rows = []
for size in (64, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 4 << 20, 16 << 20):
    n = min(TOTAL // size, 1 << 20)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(size)
    lens = np.full(n, size, dtype=np.uint64)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    for _ in range(3):
        sift_amd.compute_batch(0xFFFFFFFF, blob, offs, lens, 0xFFFFFFFF, out=out)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        sift_amd.compute_batch(0xFFFFFFFF, blob, offs, lens, 0xFFFFFFFF, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # spot-check a few against zlib
    got = out.cpu().numpy().view(np.uint32)
    for i in (0, 1, min(n, (64 << 20) // size) - 1):
        assert int(got[i]) == zlib.crc32(host[i * size:(i + 1) * size].tobytes()), (size, i)
    # same batch with the (offset, length) arrays already in HBM: no descriptor copy
    doff = torch.from_numpy(offs.astype(np.int64)).cuda()
    dlen = torch.from_numpy(lens.astype(np.int64)).cuda()
    for _ in range(3):
        sift_amd.compute_batch(0xFFFFFFFF, blob, doff, dlen, 0xFFFFFFFF, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sift_amd.compute_batch(0xFFFFFFFF, blob, doff, dlen, 0xFFFFFFFF, out=out)
    torch.cuda.synchronize()
    ddt = (time.perf_counter() - t0) / reps
    assert (out.cpu().numpy().view(np.uint32) == got).all()
    rows.append({"buffer_bytes": size, "buffers": n, "gpu_ms": round(dt * 1e3, 3),
                 "device_descriptors_ms": round(ddt * 1e3, 3),
                 "device_descriptors_GBps": round(n * size / ddt / 1e9, 1),
                 "device_descriptors_Mbuffers_per_s": round(n / ddt / 1e6, 2),
                 "gpu_GBps": round(n * size / dt / 1e9, 1), "gpu_Mbuffers_per_s": round(n / dt / 1e6, 2)})
    print(rows[-1], flush=True)

# Chunk::read on ONE long serialized chunk (its length is only in its header, on the device)
import struct
chunk_rows = []
for n in (1 << 20, 16 << 20, 64 << 20, 256 << 20, (1 << 30) - 16):
    ser = torch.empty(n + 12, dtype=torch.uint8, device="cuda")
    ser[8:8 + n] = blob[:n]
    ser[:8] = torch.tensor(list(struct.pack(">I", n) + b"AMRG"), dtype=torch.uint8, device="cuda")
    crc = sift_amd.compute(0xFFFFFFFF, ser[4:8 + n]) ^ 0xFFFFFFFF
    ser[8 + n:] = torch.tensor(list(struct.pack(">I", crc)), dtype=torch.uint8, device="cuda")
    doff = torch.zeros(1, dtype=torch.int64, device="cuda")
    dav = torch.full((1,), n + 12, dtype=torch.int64, device="cuda")
    for _ in range(3):
        st, _ = container.read_chunks_device(ser, doff, dav)
    assert st.cpu().tolist() == [0]
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        container.read_chunks_device(ser, doff, dav)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    chunk_rows.append({"chunk_data_bytes": n, "read_chunks_device_us": round(dt * 1e6, 1),
                       "GBps": round((n + 4) / dt / 1e9, 1)})
    print(chunk_rows[-1], flush=True)
    del ser

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "batch_sweep.json"), "w") as f:
    json.dump({"note": "sift_crc32_batch_device, buffers back to back in one 1 GiB device blob; time "
                       "includes the host->device copy of the (offset,len) table", "rows": rows}, f, indent=1)
