"""The RCCL leg on hardware, as far as one GPU allows: a one-rank process group
with backend "nccl" (RCCL on ROCm) runs the same calls bench.py makes with
N > 1 -- init with device_id, the int64 all_gather_into_tensor of (state, length)
pairs, the MAX all-reduce of the step time -- on device tensors.  Two ranks on
one GPU are refused by RCCL ("Duplicate GPU detected"), so the world size stays
1 here; the 2-rank exchange is covered over gloo in test_sharded.py."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys, zlib
    sys.path.insert(0, os.environ["SIFT_ROOT"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", device_id=dev)
    import sift_amd
    from sift_amd import sharded
    g = torch.Generator(device=dev).manual_seed(7)
    n = (96 << 20) + 12345
    buf = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
    want = zlib.crc32(buf.cpu().numpy().tobytes()) ^ 0xFFFFFFFF
    got = sharded.sharded_compute(0xFFFFFFFF, buf)
    assert got == want, (hex(got), hex(want))
    steps = 5
    pairs = torch.zeros(steps, 2, dtype=torch.int64, device=dev)
    pairs[:, 1] = n
    ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device=dev)
    for i in range(steps):
        sharded.local_state_async(0xFFFFFFFF, buf, 0, pairs[i], ws)
    out = sharded.exchange_and_stitch(pairs[:steps])
    assert out == [want] * steps, out
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.5
    dist.barrier()
    dist.destroy_process_group()
    print("rccl one rank ok")
""")


def test_rccl_backend_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), SIFT_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "rccl one rank ok" in r.stdout


def test_bench_rehearses_its_collectives_on_one_gpu():
    # bench.py's own N > 1 lines (nccl init with device_id, exchange_and_stitch inside the
    # timed region, barrier, MAX all-reduce, destroy) with a one-rank group, and the same
    # register as the plain one-GPU run.
    import json
    outs = []
    for extra in ([], ["--rehearse-collective"]):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, MASTER_PORT=str(port))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2",
                            "--bytes", str(256 << 20), "--no-cpu-baseline", "--no-secondary"] + extra,
                           capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    assert outs[0]["config"]["result"] == outs[1]["config"]["result"]
    assert "rehearsed" in outs[1]["config"]["parallelism"]
    assert outs[1]["n_gpus"] == 1 and outs[1]["value"] > 0
    # the untimed self-check of the exchange ran over RCCL (one rank) and passed
    assert outs[1]["config"]["verified"] is True and outs[1]["config"]["ranks_seen"] == 1
    assert "verified" not in outs[0]["config"]
