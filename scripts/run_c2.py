import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd.device import *
ctx = Context(0)
m,n,k = 200000, 20000, 256
A = cm_empty(m,n); ctx.fill_dense(A, m, n, key=(7,0)); ctx.sync()
for it in range(3):
    torch.cuda.synchronize(); t0=time.time()
    r = drv_rsvd(ctx, A, m, n, k, k, 1e-12, 0, 1, key=(0,0)); torch.cuda.synchronize(); dt=time.time()-t0
    print(f'C2 rsvd: {dt*1e3:.1f} ms', flush=True)
