import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
for (m,n) in [(1280,1024),(640,512),(2560,2048),(512,512)]:
    A = cm_empty(m,n); J = torch.zeros(n, dtype=torch.int64, device='cuda'); tau = torch.zeros(min(m,n), dtype=torch.float64, device='cuda')
    ts=[]
    for it in range(3):
        ctx.fill_dense(A, m, n, key=(3,0)); ctx.sync(); t0=time.time()
        ctx.lib.rlhip_geqp3_f64(ctx.h, m, n, A.data_ptr(), m, J.data_ptr(), tau.data_ptr()); ctx.sync(); ts.append(time.time()-t0)
    print(f'geqp3 {m}x{n}: {min(ts)*1e3:.2f} ms = {min(ts)*1e6/min(m,n):.1f} us/step', flush=True)
