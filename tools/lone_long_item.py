import sys, time
sys.path.insert(0, '/root/repo')
import torch, sift_amd
from sift_amd import crc
dev = torch.device('cuda:0')
blob = torch.randint(0, 256, (64 << 20,), dtype=torch.uint8, device=dev)
def run(n, big):
    offs = torch.arange(n, dtype=torch.int64, device=dev) * 64
    lens = torch.full((n,), 64, dtype=torch.int64, device=dev)
    if big:
        offs[n // 2] = 1 << 20
        lens[n // 2] = big
    for _ in range(5): crc.compute_batch(0xFFFFFFFF, blob, offs, lens)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): crc.compute_batch(0xFFFFFFFF, blob, offs, lens)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / 50
for n in (1000, 2000, 3500, 8000, 100000):
    print(n, 'tiny only %.1f us' % run(n, 0), ' + one 100 KiB %.1f' % run(n, 100 << 10), ' + one 250 KiB %.1f' % run(n, 250 << 10), ' + one 2 MiB %.1f' % run(n, 2 << 20), flush=True)
