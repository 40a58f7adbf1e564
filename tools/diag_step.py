import sys, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sift_amd
from sift_amd import _lib
n = 1<<30
buf = torch.randint(0,256,(n,),dtype=torch.uint8,device='cuda')
ws = torch.zeros(sift_amd.workspace_bytes(), dtype=torch.uint8, device='cuda')
out = torch.zeros(2, dtype=torch.int32, device='cuda')
for prof in (False, True):
    for _ in range(5): sift_amd.compute_async(0xFFFFFFFF, buf, out, ws)
    torch.cuda.synchronize()
    if prof: _lib.lib.sift_profile_begin(0, 100, 1)
    t0=time.perf_counter()
    for _ in range(50): sift_amd.compute_async(0xFFFFFFFF, buf, out, ws)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    if prof:
        ms=C.c_double(); k=C.c_int(); _lib.lib.sift_profile_end(0, C.byref(ms), C.byref(k)); print('kernel ms avg', ms.value/k.value)
    print('prof',prof,'launch loop %.1f us/step, total %.1f us/step'%((t1-t0)/50*1e6,(t2-t0)/50*1e6))
