import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, scipy.linalg as sl
from randlapack_amd.device import *
ctx = Context(0); lib = ctx.lib
rng = np.random.default_rng(0)
for (m,n) in [(50,20),(300,64),(2000,100),(16384,512),(65536,512),(40,40),(64,100)]:
    A = rng.standard_normal((m,n))
    Ad = cm_from_numpy(A); ip = torch.zeros(min(m,n), dtype=torch.int64, device='cuda')
    ctx.sync(); t0=time.time(); info = lib.rlhip_getrf_f64(ctx.h, m, n, Ad.data_ptr(), m, ip.data_ptr()); ctx.sync(); dt=time.time()-t0
    lu, piv = sl.lu_factor(A) if m==n else (None,None)
    import scipy.linalg.lapack as ll
    lu_ref, piv_ref, info_ref = ll.dgetrf(A)
    LU = cm_to_numpy(Ad); ipg = ip.cpu().numpy()
    print(f'getrf {m}x{n} info={info} t={dt*1e3:.1f}ms pivots equal {np.array_equal(ipg-1, piv_ref)} LU diff {np.abs(LU-lu_ref).max()/np.abs(lu_ref).max():.2e}', flush=True)
