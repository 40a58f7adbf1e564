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

With N > 1 nothing but this file runs on the node, so the job checks its own
exchange first (untimed): every rank reduces a small seeded range on its GPU, the
pairs go through the same exchange_and_stitch over RCCL, and every rank compares
the stitched register with zlib over the concatenation it regenerates on the
host (what crc.rs:28-34 gives over the whole buffer); a mismatch ends the job
with a non-zero exit code and no result line.

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
    passes = max(1, min(64, int(3.0 / max(one, 1e-3))))
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


VERIFY_BYTES = (1 << 20) + 5


def verify_ranges(world: int):
    """The seeded ranges of the self-check: rank r owns VERIFY_BYTES + 13 r bytes
    (ragged on purpose), seed 0x51F7 + 1000 + r.  Every rank can regenerate all."""
    import numpy as np
    return [np.random.default_rng(SEED + 1000 + r).integers(0, 256, size=VERIFY_BYTES + 13 * r, dtype=np.uint8)
            for r in range(world)]


def verify_exchange(local_state, rank: int, world: int, device) -> dict:
    """Untimed self-check of the N > 1 leg: local_state(state, numpy bytes) -> u32 is
    this rank's reduction (the GPU path, or zlib in --dry-collective); then the
    bench's own exchange_and_stitch; then zlib over the whole concatenation."""
    import zlib

    import numpy as np
    import torch
    import torch.distributed as dist

    from sift_amd import sharded
    M = 0xFFFFFFFF
    parts = verify_ranges(world)
    pairs = torch.zeros(2, 2, dtype=torch.int64, device=device)
    for step, start in enumerate((M, 0x12345678)):  # two registers: the initial one and an arbitrary one
        pairs[step, 0] = local_state(start if rank == 0 else 0, parts[rank])
        pairs[step, 1] = parts[rank].size
    got = sharded.exchange_and_stitch(pairs)
    whole = np.concatenate(parts).tobytes()
    want = [zlib.crc32(whole, start ^ M) ^ M for start in (M, 0x12345678)]
    ok = torch.tensor([1 if got == want else 0], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return {"verified": bool(int(ok.item())), "ranks_seen": dist.get_world_size(),
            "got": [f"0x{g:08x}" for g in got], "want": [f"0x{w:08x}" for w in want]}


def dry_collective() -> None:
    """The N > 1 exchange on CPU tensors over gloo, no GPU: every rank hashes its
    range of one logical buffer with zlib (standing in for the kernel), then the
    same exchange_and_stitch the GPU path runs; the result must equal zlib over
    the whole buffer.  The bench's self-check (verify_exchange) runs the same way.
    Launch under torch.distributed.run like the real bench."""
    import zlib

    import numpy as np
    import torch
    import torch.distributed as dist

    from sift_amd import sharded
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    M = 0xFFFFFFFF
    v = verify_exchange(lambda st, a: zlib.crc32(a.tobytes(), st ^ M) ^ M, rank, world, torch.device("cpu"))
    assert v["verified"] and v["ranks_seen"] == world, v
    # and the check must be able to fail: a rank that reduces the wrong bytes is caught
    bad = verify_exchange(lambda st, a: zlib.crc32(a.tobytes()[:-1] if rank == world - 1 else a.tobytes(), st ^ M) ^ M,
                          rank, world, torch.device("cpu"))
    assert not bad["verified"], bad
    steps, n = 3, 70001
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
        print(json.dumps({"dry_collective": "ok", "world": world, "steps": steps, "verified": v["verified"],
                          "ranks_seen": v["ranks_seen"]}), flush=True)
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

    stream = torch.cuda.current_stream(local_rank).cuda_stream  # fetched once, not per call

    verification = None
    if collective:
        def gpu_state(state, a):
            t = torch.from_numpy(a).to(dev)
            return sift_amd.compute(state, t)
        verification = verify_exchange(gpu_state, rank, world, dev)
        if not verification["verified"]:
            if rank == 0:
                print(json.dumps({"error": "the sharded exchange does not reproduce compute() over the whole buffer",
                                  **verification}), file=sys.stderr, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)

    def run(steps):
        """`steps` passes of the hot path plus the read-back and host stitch of
        every pass's result: the whole job, nothing cached between passes
        (pass i reads buffer i mod R).  With N > 1 the (state, length) pairs of all
        passes are exchanged in ONE all-gather after the last pass (fewer, larger
        collectives: 16 bytes per rank per pass), inside the timed region, and
        every pass is stitched."""
        for i in range(steps):
            sharded.local_state_async(0xFFFFFFFF, bufs[i % R], rank, pairs[i], ws, stream)
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
                **({"verified": verification["verified"], "ranks_seen": verification["ranks_seen"],
                    "verification": f"untimed: {VERIFY_BYTES} + 13 r bytes per rank (seed 0x51F7 + 1000 + r) reduced on "
                                    "each GPU, stitched by the same exchange over RCCL, equal to zlib.crc32 over the "
                                    "concatenation on every rank"} if verification else {}),
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
                           "the dispatch's own start/end timestamps, what rocprofv3 reports; a timed launch "
                           "costs the job about 4 us (--steps 20: 164.7 us a step with every launch timed, "
                           "160.4 with --event-stride 20)"),
                "algorithmic_bytes_per_launch": n,
                "traffic": load_traffic(workload),
            },
        }
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_sizes(torch, sift_amd, pool, ws) + secondary_batch(torch, sift_amd, pool)
        if world == 1 and not args.no_cpu_baseline:
            sample = bufs[0][: min(n, 256 << 20)].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample)
        print(json.dumps(out), flush=True)

    if collective:
        dist.barrier()
        dist.destroy_process_group()


def _roof(kernel: str, alg_bytes: float, kernel_us: float, launches: int) -> dict:
    gbs = alg_bytes / (kernel_us * 1e-6) / 1e9 if kernel_us else None
    return {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1) if gbs else None, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4) if gbs else None,
            "kernel_us": round(kernel_us, 2), "launches_timed": launches,
            "algorithmic_bytes_per_launch": int(alg_bytes), "traffic": None}


def _timed(_lib, dev_index: int, call, reps: int):
    """reps calls back to back with the library's dispatch-attached event pairs on every
    kernel: (wall us per call, [tile, batch, piece] kernel us per call, launches)."""
    import torch
    # the memory clock falls back within tens of milliseconds of light load (the host-side
    # set-up of a row): ~30 ms of the call itself first, untimed
    t_heat = time.perf_counter()
    while (time.perf_counter() - t_heat) < 0.03:
        for _ in range(3):
            call()
        torch.cuda.synchronize()
    _lib.check(_lib.lib.sift_profile_begin(dev_index, 2 * reps + 2, 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    ms = (C.c_double * 3)()
    n = (C.c_int * 3)()
    _lib.check(_lib.lib.sift_profile_end_kernels(dev_index, ms, n, None, 0))
    return wall, [ms[k] * 1e3 / reps for k in range(3)], [n[k] for k in range(3)]


def secondary_sizes(torch, sift_amd, pool, ws) -> list:
    """SURVEY.md section 8(d)'s other sizes, 1 MiB and 64 MiB: asynchronous calls back
    to back on one stream, each over a different slice of the pool (the slices
    used in one round total more than the 256 MiB Infinity Cache holds), wall
    time per call and the tile kernel's own duration.  Not the headline."""
    from sift_amd import _lib
    rows = []
    dev_index = pool.device.index
    stream = torch.cuda.current_stream(dev_index).cuda_stream
    out = torch.zeros(1, dtype=torch.int32, device=pool.device)
    for size, calls in ((1 << 20, 2000), (64 << 20, 400)):
        nslices = max(1, min(pool.numel() // size, max(8, (512 << 20) // size)))
        stride = (pool.numel() // nslices) // 4096 * 4096
        slices = [pool[k * stride:k * stride + size] for k in range(nslices)]
        for k in range(min(calls, 50)):
            sift_amd.compute_async(0xFFFFFFFF, slices[k % nslices], out, ws, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(calls):
            sift_amd.compute_async(0xFFFFFFFF, slices[k % nslices], out, ws, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        # the kernel's own duration, separately (64 launches with an event pair on the dispatch)
        _lib.check(_lib.lib.sift_profile_begin(dev_index, 64, 1))
        for k in range(64):
            sift_amd.compute_async(0xFFFFFFFF, slices[k % nslices], out, ws, stream)
        torch.cuda.synchronize()
        ms = (C.c_double * 3)()
        n = (C.c_int * 3)()
        _lib.check(_lib.lib.sift_profile_end_kernels(dev_index, ms, n, None, 0))
        kus = ms[0] * 1e3 / max(n[0], 1)
        rows.append({"workload": f"compute(0xFFFFFFFF, {size >> 20} MiB) x {calls} back to back, "
                                 f"{nslices} distinct slices in turn",
                     "entry_point": "sift_crc32_compute_device",
                     "us_per_call": round(dt * 1e6, 2), "GBps": round(size / dt / 1e9, 1),
                     "frac_of_hbm_peak": round(size / dt / 1e9 / HBM_PEAK_GBS, 4),
                     "roofline": _roof("crc32_tiles_kernel", size, kus, n[0])})
    return rows


def _build_documents(torch, pool, region_off: int, budget: int, max_files: int):
    """~max_files save files of 256 distinct kinds (payload 40..4000 bytes, opaque: the
    automerge bytes are out of scope) packed back to back into pool[region_off:], every
    alignment and tail; framing built on the host with struct + zlib (independent of
    the library).  Returns (offsets, lens, payload_lens) as int64 device tensors + total."""
    import struct
    import zlib

    import numpy as np
    dev = pool.device
    rng = np.random.default_rng(SEED + 77)
    magic = bytes([0x89, 0x53, 0x49, 0x46, 0x54, 0x0D, 0x0A, 0x1A, 0x0A])  # container.rs:104

    def chunk(t, d):  # container.rs:68-77
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    plen = rng.integers(40, 4001, size=256)
    files = [magic + chunk(b"AMRG", rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes()) + chunk(b"SEND", b"")
             for n in plen]
    flen = np.array([len(f) for f in files], dtype=np.int64)
    table = np.zeros((256, int(flen.max())), dtype=np.uint8)
    for k, f in enumerate(files):
        table[k, :len(f)] = np.frombuffer(f, dtype=np.uint8)
    kinds = ((np.arange(max_files, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(24)
    kinds = kinds.astype(np.int64)
    lens = flen[kinds]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    n = int(np.searchsorted(offs + lens, budget, side="right"))
    kinds, lens, offs = kinds[:n], lens[:n], offs[:n]
    total = int(offs[-1] + lens[-1])
    t_table = torch.from_numpy(table).to(dev)
    d_kinds, d_lens, d_offs = (torch.from_numpy(a).to(dev) for a in (kinds, lens, offs))
    step = 16384
    for a in range(0, n, step):  # assemble on the device, 16 K files at a time
        b = min(n, a + step)
        which = torch.repeat_interleave(torch.arange(b - a, device=dev), d_lens[a:b])
        within = torch.arange(which.numel(), device=dev) - (d_offs[a:b] - d_offs[a])[which]
        lo = region_off + int(offs[a])
        pool[lo:lo + which.numel()] = t_table[d_kinds[a:b][which], within]
    return d_offs + region_off, d_lens, torch.from_numpy(plen[kinds]).to(dev), total


def secondary_batch(torch, sift_amd, pool) -> list:
    """The batch, document and single-chunk entry points (SURVEY.md section 8 (f)-1, (f)-2 framing;
    BASELINE config "bulk insert/load of synthetic tasks" with the payload opaque), descriptors
    in HBM, each kernel timed by an event pair on its dispatch.  Algorithmic bytes count the
    payload plus, per item, the 16-byte descriptor read and every 4-byte result word written."""
    import zlib

    import numpy as np
    from sift_amd import _lib, container
    L = _lib.lib
    dev = pool.device
    di = dev.index
    stream = torch.cuda.current_stream(di).cuda_stream
    rows = []
    reps = 20
    names = ["crc32_tiles_kernel", "crc32_batch_kernel", "crc32_pieces_kernel"]

    def row(workload, entry, items, payload, per_item_extra, wall, kus, launches, checked):
        alg = payload + per_item_extra * items
        dom = max((1, 2), key=lambda k: kus[k])
        return {"workload": workload, "entry_point": entry, "items": items, "payload_bytes": int(payload),
                "us_per_call_wall": round(wall, 2),
                "kernel_us": {names[k]: round(kus[k], 2) for k in (1, 2) if launches[k]},
                "items_per_s": round(items / (sum(kus[1:]) * 1e-6), 1) if sum(kus[1:]) else None,
                "GBps_payload_over_kernel_time": round(payload / (sum(kus[1:]) * 1e-6) / 1e9, 1),
                "roofline": _roof(names[dom], alg, kus[dom], launches[dom]), "checked": checked}

    # -- many plain buffers: 1 Mi x 64 B, 1 Mi x 1 KiB and 1024 x 1 MiB, back to back in the pool
    out = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
    host = None
    for n, size in ((1 << 20, 64), (1 << 20, 1024), (1024, 1 << 20)):
        offs = torch.arange(n, dtype=torch.int64, device=dev) * size
        lens = torch.full((n,), size, dtype=torch.int64, device=dev)

        def call():
            _lib.check(L.sift_crc32_batch_device(di, 0xFFFFFFFF, 0xFFFFFFFF, pool.data_ptr(), offs.data_ptr(),
                                                 lens.data_ptr(), n, out.data_ptr(), stream))
        wall, kus, launches = _timed(_lib, di, call, reps)
        pick = np.linspace(0, n - 1, 257).astype(np.int64)
        got = out[torch.from_numpy(pick).to(dev)].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        if host is None:
            host = pool[: 1 << 30].cpu().numpy()
        want = np.array([zlib.crc32(host[int(i) * size:(int(i) + 1) * size].tobytes()) for i in pick])
        assert (got == want).all(), f"batch {size} B: results differ from zlib"
        rows.append(row(f"crc32 of {n} buffers of {size} B (uniform random bytes), descriptors in HBM", "sift_crc32_batch_device",
                        n, n * size, 16 + 4, wall, kus, launches, "257 sampled results == zlib.crc32"))
    del host

    # -- read_document's framing on ~0.5 M save files of mixed 73..4033 B
    region = 1 << 30  # second GiB of the pool
    offs, lens, plens, total = _build_documents(torch, pool, region, 1 << 30, 500_000)
    nf = offs.numel()
    status = torch.zeros(nf, dtype=torch.int32, device=dev)
    dlen = torch.zeros(nf, dtype=torch.int32, device=dev)

    def call_docs():
        _lib.check(L.sift_read_documents_device(di, pool.data_ptr(), offs.data_ptr(), lens.data_ptr(), nf,
                                                status.data_ptr(), dlen.data_ptr(), stream))
    # the check must be able to fail: damage three files, expect exactly those to be refused
    victims = [5, nf // 2, nf - 1]
    saved = []
    for v in victims:
        at = int(offs[v].item()) + 30
        saved.append((at, int(pool[at].item())))
        pool[at] ^= 1
    call_docs()
    torch.cuda.synchronize()
    bad = torch.nonzero(status).flatten().cpu().tolist()
    assert bad == victims and all(int(status[v].item()) == _lib.ERR_READ_CRC_MISMATCH for v in victims), bad
    for at, val in saved:
        pool[at] = val
    wall, kus, launches = _timed(_lib, di, call_docs, reps)
    assert int(status.abs().sum().item()) == 0 and bool((dlen.to(torch.int64) == plens).all().item())
    rows.append(row(f"read_document framing (document.rs:62-77) of {nf} save files of mixed 73..4033 B packed back to "
                    "back (256 kinds, payload opaque), descriptors in HBM", "sift_read_documents_device", nf, total,
                    16 + 4 + 4, wall, kus, launches,
                    "every status OK and AMRG length right; three damaged files refused with ReadCrcMismatch"))

    # -- Chunk::read on ONE chunk whose length is only in its header: 64 MiB and the 1 GiB cap
    for data_len in (64 << 20, (1 << 30) - 16):
        at = (2 << 30) + 16  # third GiB of the pool onwards
        hdr = np.frombuffer(int(data_len).to_bytes(4, "big") + b"AMRG", dtype=np.uint8)
        pool[at:at + 8] = torch.from_numpy(hdr.copy()).to(dev)
        crc = sift_amd.compute(0xFFFFFFFF, pool[at + 4:at + 8 + data_len]) ^ 0xFFFFFFFF  # container.rs:79-84
        pool[at + 8 + data_len:at + 12 + data_len] = torch.from_numpy(
            np.frombuffer(int(crc).to_bytes(4, "big"), dtype=np.uint8).copy()).to(dev)
        o1 = torch.tensor([at], dtype=torch.int64, device=dev)
        a1 = torch.tensor([data_len + 12], dtype=torch.int64, device=dev)
        st1 = torch.zeros(2, dtype=torch.int32, device=dev)

        def call_chunk():
            _lib.check(L.sift_read_chunks_device(di, pool.data_ptr(), o1.data_ptr(), a1.data_ptr(), 1,
                                                 st1.data_ptr(), st1.data_ptr() + 4, stream))
        pool[at + 8 + data_len // 2] ^= 0x10
        call_chunk()
        torch.cuda.synchronize()
        assert st1.cpu().tolist() == [_lib.ERR_READ_CRC_MISMATCH, 0], st1
        pool[at + 8 + data_len // 2] ^= 0x10
        wall, kus, launches = _timed(_lib, di, call_chunk, reps)
        assert st1.cpu().tolist() == [0, data_len], st1
        rows.append(row(f"Chunk::read (container.rs:47-66) of one chunk of {data_len} data bytes, length known only "
                        "from its header, descriptor in HBM", "sift_read_chunks_device", 1, data_len + 12, 16 + 8,
                        wall, kus, launches, "status OK and length right; a flipped data bit gives ReadCrcMismatch"))
    return rows


if __name__ == "__main__":
    main()
