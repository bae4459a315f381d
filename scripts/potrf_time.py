import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
rng = np.random.default_rng(0)
for n in (64, 256, 300, 448):
    B = rng.standard_normal((2000,n)); G = B.T@B
    Gd = cm_from_numpy(G); 
    ctx.potrf(n, Gd, n); ctx.sync()
    R = np.triu(cm_to_numpy(Gd)); Rref = np.linalg.cholesky(G).T
    ts=[]
    for rep in range(5):
        Gd = cm_from_numpy(G); ctx.sync(); t0=time.time(); ctx.potrf(n, Gd, n); ctx.sync(); ts.append(time.time()-t0)
    print(f'potrf n={n} err {np.abs(R-Rref).max()/np.abs(Rref).max():.2e} t={min(ts)*1e6:.0f} us')
