import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
m,n = 1048576, 1024
A = cm_empty(m,n)
for it in range(3):
    ctx.fill_dense(A, m, n, key=(3,0)); ctx.sync()
    t0=time.time(); r = drv_cqrrpt(ctx, A, m, n, 1.25, 4, timing=(it==2)); ctx.sync(); dt=time.time()-t0
    fl = 2*4*m*n + (2*1280*n*n - 2/3*n**3) + 3*m*n*n + 4/3*n**3
    print(f'C3 cqrrpt: {dt*1e3:.1f} ms rank={r["rank"]} -> {fl/dt/1e12:.1f} TFLOP/s times(us) {r.get("times_us")}', flush=True)
