"""CPU, world_size 2 over gloo: the multi-GPU exchange and stitch of
sift_amd.sharded.  The per-rank reduction is injected (the
oracle stands in for the GPU kernel here); what is under test is the
(state, length) all-gather and the GF(2) stitch through the C ABI."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cuts, seed, init_state, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle
        from sift_amd import sharded
        full = np.random.default_rng(seed).integers(0, 256, size=cuts[-1], dtype=np.uint8)
        local = torch.from_numpy(full[cuts[rank]:cuts[rank + 1]].copy())
        got = sharded.sharded_compute(init_state, local,
                                      local_compute=lambda s, t: c_oracle.compute(s, t.numpy()))
        q.put((rank, got, c_oracle.compute(init_state, full)))
    finally:
        dist.destroy_process_group()


def _worker_many(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle
        from sift_amd import sharded
        rng = np.random.default_rng(3)
        steps = 5
        fulls = [rng.integers(0, 256, size=int(n), dtype=np.uint8) for n in (10, 4000, 70001, 1, 33333)]
        pairs = torch.zeros(steps, 2, dtype=torch.int64)
        for i, full in enumerate(fulls):
            cut = full.size // 3
            part = full[:cut] if rank == 0 else full[cut:]
            pairs[i, 0] = c_oracle.compute(0xFFFFFFFF if rank == 0 else 0, part)
            pairs[i, 1] = part.size
        got = sharded.exchange_and_stitch(pairs)
        q.put((rank, got, [c_oracle.compute(0xFFFFFFFF, f) for f in fulls]))
    finally:
        dist.destroy_process_group()


def test_exchange_many_buffers_one_all_gather():
    # bench.py's N > 1 exchange: every step's pair in one all-gather, then a stitch per step
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_many, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, want in res:
        assert got == want, rank


@pytest.mark.parametrize("cuts,init_state", [
    ([0, 1000, 2500], 0xFFFFFFFF),
    ([0, 0, 77], 0xFFFFFFFF),          # empty first shard
    ([0, 4097, 4097], 0x12345678),     # empty last shard
    ([0, 1 << 20, (1 << 20) + 3], 0),
])
def test_two_rank_stitch(cuts, init_state):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cuts, 7, init_state, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, want in res:
        assert got == want, (rank, hex(got), hex(want))


def test_stitch_many_ranges_single_process(oracle):
    from sift_amd import sharded
    rng = np.random.default_rng(5)
    full = rng.integers(0, 256, size=200000, dtype=np.uint8)
    cuts = [0] + sorted(int(c) for c in rng.integers(0, full.size, size=7)) + [full.size]
    pairs = []
    for r, (a, b) in enumerate(zip(cuts, cuts[1:])):
        pairs.append((oracle.compute(0xFFFFFFFF if r == 0 else 0, full[a:b]), b - a))
    assert sharded.stitch(pairs) == oracle.compute(0xFFFFFFFF, full)


def test_bench_dry_collective_two_ranks():
    # bench.py's own N > 1 exchange, launched the way the driver launches the bench
    # (torch.distributed.run, 2 ranks), on CPU tensors over gloo: everything of the
    # multi-GPU bench leg except the backend string and the kernel.
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
         os.path.join(root, "bench.py"), "--dry-collective"],
        capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # "verified": the bench's untimed self-check of the exchange (verify_exchange) passed on
    # both ranks -- and, inside the dry run, a deliberately wrong rank was caught by it
    assert lines and json.loads(lines[-1]) == {"dry_collective": "ok", "world": 2, "steps": 3,
                                               "verified": True, "ranks_seen": 2}
