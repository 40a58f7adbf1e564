"""Needs a -DSIFT_TRACE_FINISH build of the library
(make -C sift_amd/csrc EXTRA=-DSIFT_TRACE_FINISH, then SIFT_AMD_LIB=<that .so>):
the tile kernel then leaves four 100 MHz timestamps per workgroup in the
workspace's residue area (start, tables staged, stream done, before the
exchange).  Prints, per launch, the event-bracketed kernel time and how the 256
workgroups -- and the eight XCDs, workgroup g runs on XCD g mod 8 -- spread."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sift_amd
from sift_amd import _lib, sharded


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device("cuda:0")
    _lib.check(_lib.lib.sift_init(0))
    R = 4
    pool = torch.empty(n * R, dtype=torch.uint8, device=dev)
    pool.random_(0, 256)
    ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device=dev)
    pairs = torch.zeros(K, 2, dtype=torch.int64, device=dev)
    for i in range(32):
        sharded.local_state_async(0xFFFFFFFF, pool[(i % R) * n:(i % R + 1) * n], 0, pairs[0], ws)
    torch.cuda.synchronize()
    rows = []
    for i in range(K):
        _lib.check(_lib.lib.sift_profile_begin(0, 3, 1))
        for j in range(3):  # back to back, the last one is looked at
            b = (3 * i + j) % R
            sharded.local_state_async(0xFFFFFFFF, pool[b * n:(b + 1) * n], 0, pairs[i], ws)
        torch.cuda.synchronize()
        kms, kl = C.c_double(0.0), C.c_int(0)
        samples = (C.c_float * 3)()
        _lib.check(_lib.lib.sift_profile_end_samples(0, C.byref(kms), C.byref(kl), samples, 3))
        w = ws.view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        t = w[:1024].reshape(4, 256)
        us = (t - t[0].min()) / 100.0
        ev = samples[2] * 1e3
        rows.append((ev, us[2].max(), us[2].mean(), us[3].max()))
        print(f"launch {i}: events {ev:6.1f} us; start spread {us[0].max():4.2f}; staged {us[1].min():4.2f}.."
              f"{us[1].max():4.2f}; stream done {us[2].min():6.1f}..{us[2].max():6.1f} (mean {us[2].mean():6.1f}); "
              f"last group at the exchange {us[3].max():6.1f}")
        print("    per XCD mean stream-done:", [round(float(us[2][x::8].mean()), 1) for x in range(8)],
              " spread inside an XCD:", [round(float(us[2][x::8].max() - us[2][x::8].min()), 1) for x in range(8)])
    a = np.array(rows[1:])
    print(f"means: events {a[:, 0].mean():6.1f}  last group's stream done {a[:, 1].mean():6.1f}  "
          f"stream done over groups {a[:, 2].mean():6.1f}  last group at the exchange {a[:, 3].mean():6.1f}")


if __name__ == "__main__":
    main()
