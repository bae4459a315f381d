import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
rng = np.random.default_rng(0)
B = rng.standard_normal((20000,256)); A = np.linalg.qr(B)[1].T.copy()
for rep in range(3):
    Ad = cm_from_numpy(A); S = torch.empty(256, dtype=torch.float64, device='cuda'); VT = cm_empty(256,256)
    ctx.sync(); t0=time.time(); info, sw = ctx.gesvdj(256,256,Ad,256,S,VT,256); ctx.sync(); dt=time.time()-t0
print(f'dbg={os.environ.get("RLHIP_JACOBI_DBG")} sweeps={sw} t={dt*1e3:.2f}ms per-launch {dt*1e6/(sw*16):.1f} us')
