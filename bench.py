#!/usr/bin/env python3
"""bench.py -- CRC-32 hot-path throughput on MI355X.

BASELINE.json names no metric (it reads "N/A: exclude repo"); SURVEY.md section 8(d)
defines the contingency workload measured here: compute(0xFFFFFFFF, buf)
(sift-persist/src/crc.rs:28-34) over one 1 GiB buffer of uniform random bytes
(seed 0x51F7), 1 GiB being the reference's chunk cap (container.rs:49).

A step is one compute() over the whole buffer, resident in HBM before the
timed region.  With N > 1 (one process per GPU) the logical buffer is N GiB,
rank r owns range r, and the job ends with one all-gather of every step's
(state, length) pair per rank over RCCL and a stitch per step -- weak scaling,
no data-path collective.

Prints one JSON line (rank 0).  `roofline` is for the tile kernel, timed with
HIP events on its own stream during the timed steps (the event pair rides on the
kernel's dispatch, see sift_profile_begin in include/sift_amd.h); `cpu_baseline` is the C
port of the reference's byte-at-a-time loop (oracle/sift_oracle.c), 1 core, on
a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x51F7
GIB = 1 << 30
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(host_sample) -> dict:
    """The oracle (C port of crc.rs:28-34) on one host core, bounded sample."""
    from oracle import c_oracle
    L = c_oracle.lib()
    ptr = host_sample.ctypes.data
    n = host_sample.size
    t0 = time.perf_counter()
    L.oracle_crc_repeat(ptr, n, 1)
    one = time.perf_counter() - t0
    passes = max(1, min(64, int(12.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    L.oracle_crc_repeat(ptr, n, passes)
    dt = time.perf_counter() - t0
    return {
        "value": round(n * passes / dt / 1e9, 4),
        "unit": "GB/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{passes} passes over the first {n >> 20} MiB of the workload buffer "
                  f"({n * passes / GIB:.2f} GiB hashed, {dt:.1f} s)",
    }


def load_traffic(workload: str):
    """HBM bytes per tile-kernel launch from the committed PMC run, or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("workload") == workload:
            return t.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def dry_collective() -> None:
    """The N > 1 exchange on CPU tensors over gloo, no GPU: every rank hashes its
    range of one logical buffer with zlib (standing in for the kernel), then the
    same exchange_and_stitch the GPU path runs; the result must equal zlib over
    the whole buffer.  Launch under torch.distributed.run like the real bench."""
    import zlib

    import numpy as np
    import torch
    import torch.distributed as dist

    from sift_amd import sharded
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    steps, n = 3, 70001
    M = 0xFFFFFFFF
    whole = [np.random.default_rng(SEED + i).integers(0, 256, size=n * world + 5, dtype=np.uint8)
             for i in range(steps)]
    pairs = torch.zeros(steps, 2, dtype=torch.int64)
    for i in range(steps):
        lo = rank * n
        hi = (rank + 1) * n if rank < world - 1 else whole[i].size  # ragged last range
        state = zlib.crc32(whole[i][lo:hi].tobytes(), (M if rank == 0 else 0) ^ M) ^ M
        pairs[i, 0] = state
        pairs[i, 1] = hi - lo
    got = sharded.exchange_and_stitch(pairs)
    want = [zlib.crc32(w.tobytes()) ^ M for w in whole]
    assert got == want, (rank, got, want)
    if rank == 0:
        print(json.dumps({"dry_collective": "ok", "world": world, "steps": steps}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--bytes", type=int, default=GIB, help="bytes per GPU per step")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct buffers per GPU, used in turn (seeds 0x51F7 + i): with 4 x 1 GiB no "
                         "step can be served from the 256 MiB Infinity Cache")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 1 MiB / 64 MiB lines")
    ap.add_argument("--preheat-ms", type=float, default=80.0,
                    help="untimed clock pre-heat before the warmup steps")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="bracket every n-th launch's tile kernel with HIP events; 0 = every launch "
                         "when steps < 64, else about 64 samples")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="with one GPU: still open the RCCL process group (one rank) and run every "
                         "collective of the N > 1 leg, so that those lines have run on hardware")
    ap.add_argument("--dry-collective", action="store_true",
                    help="run only the N > 1 exchange, on CPU tensors over gloo (no GPU)")
    args = ap.parse_args()
    if args.dry_collective:
        dry_collective()
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.rehearse_collective
    if collective:
        if world == 1:  # rehearsal without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    import sift_amd
    from sift_amd import _lib, sharded
    _lib.check(_lib.lib.sift_init(local_rank))

    n = args.bytes
    R = max(1, args.rotate)
    # one pool, R buffers back to back; buffer i of rank r is seeded 0x51F7 + i + 64 r
    pool = torch.empty(n * R, dtype=torch.uint8, device=dev)
    bufs = []
    for i in range(R):
        g = torch.Generator(device=dev).manual_seed(SEED + i + 64 * rank)
        bufs.append(pool[i * n:(i + 1) * n])
        bufs[i].copy_(torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g))
    ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device=dev)
    K, W = args.steps, args.warmup
    pairs = torch.zeros(max(K, 1), 2, dtype=torch.int64, device=dev)
    pairs[:, 1] = n

    def run(steps):
        """`steps` passes of the hot path plus the read-back and host stitch of
        every pass's result: the whole job, nothing cached between passes
        (pass i reads buffer i mod R).  With N > 1 the (state, length) pairs of all
        passes are exchanged in ONE all-gather after the last pass (fewer, larger
        collectives: 16 bytes per rank per pass), inside the timed region, and
        every pass is stitched."""
        for i in range(steps):
            sharded.local_state_async(0xFFFFFFFF, bufs[i % R], rank, pairs[i], ws)
        if collective:
            return sharded.exchange_and_stitch(pairs[:steps])
        torch.cuda.synchronize()
        return [int(s) & 0xFFFFFFFF for s in pairs[:steps, 0].cpu().tolist()]

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock pre-heat (untimed, reported in config.preheat_ms): the GPU ramps its
    # clocks over the first tens of milliseconds of load, far longer than a short
    # --warmup lasts; without this a small K measures the ramp, not the kernel.
    t_heat = time.perf_counter()
    while (time.perf_counter() - t_heat) * 1e3 < args.preheat_ms:
        run(min(32, max(K, 1)))
    done = 0
    while done < W:  # untimed: same code path, so lazy kernel loads happen here
        c = min(W - done, max(K, 1))
        run(c)
        done += c
    pairs[:, 0] = 0
    fence()
    stride = args.event_stride if args.event_stride > 0 else (1 if K < 64 else max(1, K // 64))
    _lib.check(_lib.lib.sift_profile_begin(local_rank, K, stride))
    t0 = time.perf_counter()
    results = run(K)
    fence()
    dt = time.perf_counter() - t0
    kms = C.c_double(0.0)
    kl = C.c_int(0)
    samples = (C.c_float * max(K, 1))()
    _lib.check(_lib.lib.sift_profile_end_samples(local_rank, C.byref(kms), C.byref(kl), samples, K))

    if collective:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    for i in range(R):  # every pass over the same buffer must agree
        assert len(set(results[i::R])) <= 1, "steps over the same buffer must agree"

    if rank == 0:
        total_bytes = float(n) * world * K
        workload = f"compute(0xFFFFFFFF, {n >> 20} MiB uniform random bytes, seed 0x51F7) per GPU"
        kernel_s = kms.value / 1e3 / max(kl.value, 1)
        achieved = n / kernel_s / 1e9 if kl.value else None
        ks = sorted(float(samples[i]) * 1e3 for i in range(min(kl.value, K)))
        out = {
            "metric": "crc32_compute_throughput",
            "value": round(total_bytes / dt / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "note": "BASELINE.json metric is N/A; workload per SURVEY.md section 8(d): "
                        "1 GiB = chunk cap of container.rs:49",
                "bytes_per_gpu_per_step": n,
                "rotate": R,
                "rotate_note": f"{R} distinct buffers per GPU used in turn ({R * n >> 20} MiB): "
                               "no step re-reads bytes that could still sit in the 256 MiB Infinity Cache",
                "preheat_ms": args.preheat_ms,
                "result": f"0x{results[0]:08x}",
                "parallelism": f"range-sharded x{world}, one (state,len) all-gather per job" if world > 1
                               else ("single GPU (one-rank RCCL group rehearsed)" if collective else "single GPU"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "crc32_tiles_kernel<true>",
                "achieved": round(achieved, 1) if achieved else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                "kernel_us": round(kernel_s * 1e6, 2),
                "kernel_us_min": round(ks[0], 2) if ks else None,
                "kernel_us_median": round(ks[len(ks) // 2], 2) if ks else None,
                "kernel_us_max": round(ks[-1], 2) if ks else None,
                "launches_timed": kl.value,
                "event_stride": stride,
                "timing": ("HIP event pair recorded before and behind each timed launch (SIFT_PROF_EXT=0)"
                           if os.environ.get("SIFT_PROF_EXT") == "0" else
                           "HIP event pair attached to each timed launch's dispatch (hipExtLaunchKernelGGL): "
                           "the dispatch's own start/end timestamps, what rocprofv3 reports"),
                "algorithmic_bytes_per_launch": n,
                "traffic": load_traffic(workload),
            },
        }
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_sizes(torch, sift_amd, pool, ws)
        if world == 1 and not args.no_cpu_baseline:
            sample = bufs[0][: min(n, 256 << 20)].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample)
        print(json.dumps(out), flush=True)

    if collective:
        dist.barrier()
        dist.destroy_process_group()


def secondary_sizes(torch, sift_amd, pool, ws) -> list:
    """SURVEY.md section 8(d)'s other sizes, 1 MiB and 64 MiB: asynchronous calls back
    to back on one stream, each over a different slice of the pool (the slices
    used in one round total more than the 256 MiB Infinity Cache holds), wall
    time per call.  Not the headline: reported beside it."""
    rows = []
    out = torch.zeros(1, dtype=torch.int32, device=pool.device)
    for size, calls in ((1 << 20, 2000), (64 << 20, 400)):
        nslices = max(1, min(pool.numel() // size, max(8, (512 << 20) // size)))
        stride = (pool.numel() // nslices) // 4096 * 4096
        slices = [pool[k * stride:k * stride + size] for k in range(nslices)]
        for k in range(min(calls, 50)):
            sift_amd.compute_async(0xFFFFFFFF, slices[k % nslices], out, ws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(calls):
            sift_amd.compute_async(0xFFFFFFFF, slices[k % nslices], out, ws)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        rows.append({"workload": f"compute(0xFFFFFFFF, {size >> 20} MiB) x {calls} back to back, "
                                 f"{nslices} distinct slices in turn",
                     "us_per_call": round(dt * 1e6, 2), "GBps": round(size / dt / 1e9, 1),
                     "frac_of_hbm_peak": round(size / dt / 1e9 / HBM_PEAK_GBS, 4)})
    return rows


if __name__ == "__main__":
    main()
