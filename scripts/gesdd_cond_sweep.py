import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0); lib = ctx.lib
rng = np.random.default_rng(1)
m, n = 3000, 96
for lc in range(4, 15):
    for shape in ('log', 'step'):
        cond = 10.0**lc
        s = np.logspace(0, -lc, n) if shape == 'log' else np.concatenate([np.ones(n//2), np.full(n - n//2, 1/cond)])
        A = (np.linalg.qr(rng.standard_normal((m,n)))[0]*s)@np.linalg.qr(rng.standard_normal((n,n)))[0].T
        Ad = cm_from_numpy(A); S = torch.zeros(n, dtype=torch.float64, device='cuda'); U = cm_empty(m,n); VT = cm_empty(n,n)
        import ctypes as C
        sw = C.c_int(0)
        rc = lib.rlhip_gesdd_f64(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, C.byref(sw)); ctx.sync()
        Un, Sn, VTn = cm_to_numpy(U), S.cpu().numpy(), cm_to_numpy(VT)
        sref = np.linalg.svd(A, compute_uv=False)
        print(f'cond 1e{lc} {shape}: rc {rc} sweeps {sw.value} orthU {np.linalg.norm(Un.T@Un-np.eye(n)):.1e} orthV {np.linalg.norm(VTn@VTn.T-np.eye(n)):.1e} recon {np.linalg.norm(Un*Sn@VTn-A)/np.linalg.norm(A):.1e} Sabs {np.max(np.abs(Sn-sref))/sref[0]:.1e} Srel {np.max(np.abs(Sn-sref)/sref):.1e}', flush=True)
