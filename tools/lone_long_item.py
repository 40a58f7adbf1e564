"""One long item among many short ones: device time per call of the batch entry point for n
items of 64 bytes, alone and with ONE item of 100 KiB / 250 KiB / 2 MiB among them (descriptors
in HBM, through the Python binding: about 11 us of every figure is the binding).  The long
item must cost what its bytes cost, whatever n is (round 3: with the cut threshold chosen
from n alone, 3500 items + one of 250 KiB took 62 us)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
