import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
rng = np.random.default_rng(0)
for (m,n,kind) in [(256,256,'RT'),(256,256,'graded'),(200,100,'gauss'),(256,33,'gauss'),(64,64,'RT'),(130,130,'RT'),(300,64,'gauss')]:
    if kind=='RT':
        B = rng.standard_normal((2000,n))@np.diag(np.logspace(0,-6,n))@np.linalg.qr(rng.standard_normal((n,n)))[0]
        A = np.linalg.qr(B)[1].T.copy()[:m,:n]
    elif kind=='graded':
        A = rng.standard_normal((m,n))@np.diag(np.logspace(0,-8,n))
    else:
        A = rng.standard_normal((m,n))
    Ad = cm_from_numpy(A); S = torch.empty(n, dtype=torch.float64, device='cuda'); VT = cm_empty(n,n)
    ctx.sync(); t0=time.time(); info, sw = ctx.gesvdj(m,n,Ad,m,S,VT,n); ctx.sync(); dt=time.time()-t0
    U = cm_to_numpy(Ad); s = S.cpu().numpy(); vt = cm_to_numpy(VT)
    sref = np.linalg.svd(A, compute_uv=False)
    print(f'gesvdj {kind} {m}x{n} info={info} sweeps={sw} t={dt*1e3:.2f}ms recon={np.abs(U*s@vt-A).max()/np.abs(A).max():.2e} sabs={np.max(np.abs(s-sref))/sref[0]:.2e} orthU={np.abs(U.T@U-np.eye(n)).max():.2e} orthV={np.abs(vt@vt.T-np.eye(n)).max():.2e}', flush=True)
# bench-like: R^T of a CholQR of a Gaussian 20000 x 256
B = rng.standard_normal((20000,256)); A = np.linalg.qr(B)[1].T.copy()
for rep in range(3):
    Ad = cm_from_numpy(A); S = torch.empty(256, dtype=torch.float64, device='cuda'); VT = cm_empty(256,256)
    ctx.sync(); t0=time.time(); info, sw = ctx.gesvdj(256,256,Ad,256,S,VT,256); ctx.sync(); dt=time.time()-t0
    s = S.cpu().numpy(); sref = np.linalg.svd(A, compute_uv=False); U = cm_to_numpy(Ad); vt = cm_to_numpy(VT)
    print(f'bench-like gesvdj sweeps={sw} t={dt*1e3:.2f}ms sabs={np.max(np.abs(s-sref))/sref[0]:.2e} recon={np.abs(U*s@vt-A).max()/np.abs(A).max():.2e} orthU={np.abs(U.T@U-np.eye(256)).max():.2e}')
