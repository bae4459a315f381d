import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
n = 1024
for m in (1048576, 1048576 + 256, 1000000):
    A = cm_empty(m,n)
    for it in range(2):
        ctx.fill_dense(A, m, n, key=(3,0)); ctx.sync()
        t0=time.time(); r = drv_cqrrpt(ctx, A, m, n, 1.25, 4, timing=(it==1)); ctx.sync(); dt=time.time()-t0
    print(f'm={m}: {dt*1e3:.1f} ms times(us) {r.get("times_us")}', flush=True)
    del A
