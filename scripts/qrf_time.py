import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from randlapack_amd.device import *
import scipy.linalg.lapack as ll
ctx = Context(0)
for (m,n) in [(512,512),(1280,1024),(2048,2048),(20000,128),(100000,64),(640,5000)]:
    A = cm_empty(m,n); tau = torch.zeros(min(m,n), dtype=torch.float64, device='cuda')
    ts=[]
    for it in range(3):
        ctx.fill_dense(A, m, n, key=(3,0)); ctx.sync(); t0=time.time()
        rc = ctx.lib.rlhip_geqrf_f64(ctx.h, m, n, A.data_ptr(), m, tau.data_ptr()); ctx.sync(); ts.append(time.time()-t0)
    msg = f'geqrf {m}x{n}: rc {rc} {min(ts)*1e3:.2f} ms = {min(ts)*1e6/min(m,n):.1f} us/step'
    if m*n <= 3e6:
        A0 = cm_empty(m,n); ctx.fill_dense(A0, m, n, key=(3,0)); ctx.sync()
        qr_ref, tau_ref, _, _ = ll.dgeqrf(cm_to_numpy(A0))
        msg += f' | diff {np.abs(cm_to_numpy(A)-qr_ref).max()/np.abs(qr_ref).max():.1e} tau {np.abs(tau.cpu().numpy()-tau_ref).max():.1e}'
    print(msg, flush=True)
