"""Per-launch kernel times of the 1 GiB compute call, in launch order, with the
buffer each launch read: shows whether the spread between launches follows the
buffer (address placement) or the launch (clock / neighbours)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sift_amd
from sift_amd import _lib, sharded


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    skew = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # bytes added to every buffer's start
    dev = torch.device("cuda:0")
    _lib.check(_lib.lib.sift_init(0))
    pool = torch.empty(n * R + skew + 4096, dtype=torch.uint8, device=dev)
    pool.random_(0, 256)
    bufs = [pool[skew + i * n: skew + (i + 1) * n] for i in range(R)]
    ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device=dev)
    pairs = torch.zeros(K, 2, dtype=torch.int64, device=dev)
    for rep in range(3):
        for i in range(K):
            sharded.local_state_async(0xFFFFFFFF, bufs[i % R], 0, pairs[i], ws)
    torch.cuda.synchronize()
    _lib.check(_lib.lib.sift_profile_begin(0, K, 1))
    for i in range(K):
        sharded.local_state_async(0xFFFFFFFF, bufs[i % R], 0, pairs[i], ws)
    torch.cuda.synchronize()
    kms, kl = C.c_double(0.0), C.c_int(0)
    samples = (C.c_float * K)()
    _lib.check(_lib.lib.sift_profile_end_samples(0, C.byref(kms), C.byref(kl), samples, K))
    per = [[] for _ in range(R)]
    for i in range(kl.value):
        per[i % R].append(samples[i] * 1e3)
    print("base address mod 2 MiB:", [b.data_ptr() % (2 << 20) for b in bufs])
    for b in range(R):
        v = per[b]
        print(f"buffer {b}: " + " ".join(f"{x:6.1f}" for x in v) + f"   mean {sum(v) / len(v):.1f}")
    allv = [x for v in per for x in v]
    print(f"all: mean {sum(allv) / len(allv):.2f} min {min(allv):.2f} max {max(allv):.2f}")


if __name__ == "__main__":
    main()
