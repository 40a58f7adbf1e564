import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sift_amd
n = 1 << 30
buf = torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda')
ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device='cuda')
K = 200
out = torch.zeros(K, dtype=torch.int64, device='cuda')
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for i in range(K):
        sift_amd.compute_async(0xFFFFFFFF, buf, out[i:i+1].view(torch.int32), ws)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    r = out.cpu().tolist()
    t3 = time.perf_counter()
    print(f"rep {rep}: launch {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t0):.2f} ms ({1e6*(t2-t0)/K:.1f} us/step), readback {1e3*(t3-t2):.3f} ms")
# per-kernel timeline with events on every 20th
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
torch.cuda.synchronize()
for i in range(K):
    if i % 20 == 0: ev[i][0].record()
    sift_amd.compute_async(0xFFFFFFFF, buf, out[i:i+1].view(torch.int32), ws)
    if i % 20 == 19: ev[i-19][1].record()
torch.cuda.synchronize()
print([round(ev[i][0].elapsed_time(ev[i][1])/20*1e3,1) for i in range(0,K,20)])
