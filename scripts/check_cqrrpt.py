import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from randlapack_amd.device import *
from randlapack_amd import _lib
from _gen import poly_mat
import oracle
ctx = Context(0); lib = ctx.lib
rng = np.random.default_rng(0)
for (m,n) in [(10,7),(1000,200),(513,64),(4,6)]:
    A = rng.standard_normal((m,n)); J = rng.permutation(n)+1
    Ad = cm_from_numpy(A); Jd = torch.from_numpy(J).cuda()
    rc = lib.rlhip_col_swap_f64(ctx.h, m, n, n, Ad.data_ptr(), m, Jd.data_ptr()); ctx.sync()
    print('col_swap', m, n, rc, np.array_equal(cm_to_numpy(Ad), A[:, J-1]), np.array_equal(Jd.cpu().numpy(), J))
for (m,n) in [(50,20),(300,200),(1280,1024),(200,300)]:
    A = rng.standard_normal((m,n)) * np.logspace(0,-3,n)[rng.permutation(n)]
    Ad = cm_from_numpy(A); Jd = torch.zeros(n, dtype=torch.int64, device='cuda'); tau = torch.zeros(min(m,n), dtype=torch.float64, device='cuda')
    ctx.sync(); t0=time.time(); rc = lib.rlhip_geqp3_f64(ctx.h, m, n, Ad.data_ptr(), m, Jd.data_ptr(), tau.data_ptr()); ctx.sync(); dt=time.time()-t0
    info, Ao, Jo, tauo = oracle.geqp3(A)
    Rg = np.triu(cm_to_numpy(Ad))[:min(m,n)]; Ro = np.triu(Ao)[:min(m,n)]
    Jg = Jd.cpu().numpy()
    print(f'geqp3 {m}x{n} rc={rc} t={dt*1e3:.1f}ms pivots_equal={np.array_equal(Jg,Jo)} nmismatch={(Jg!=Jo).sum()} |R|diff={np.abs(np.abs(Rg)-np.abs(Ro)).max()/np.abs(Ro).max():.2e} tau diff={np.abs(tau.cpu().numpy()-tauo).max():.2e}', flush=True)
d, m, n, nnz = 40, 1000, 16, 4
S = C.c_void_p(); nxt = (C.c_uint32*4)()
rc = lib.rlhip_saso_create(ctx.h, d, m, nnz, Context._u32((0,0,0,0)), Context._u32((5,0)), nxt, C.byref(S))
Sd = cm_empty(d, m); lib.rlhip_saso_dense_f64(ctx.h, S, Sd.data_ptr()); Sh = cm_to_numpy(Sd)
print('saso dense: nnz per col', set((Sh!=0).sum(0)), 'values', set(np.unique(Sh)), 'row counts min/max', (Sh!=0).sum(1).min(), (Sh!=0).sum(1).max(), 'next', list(nxt))
A = rng.standard_normal((m,n)); Ad = cm_from_numpy(A); Bd = cm_zeros(d,n)
lib.rlhip_saso_apply_f64(ctx.h, S, n, 1.0, Ad.data_ptr(), m, 0.0, Bd.data_ptr(), d); ctx.sync()
print('saso apply err', np.abs(cm_to_numpy(Bd) - Sh@A).max())
lib.rlhip_saso_destroy(ctx.h, S)
for (m,n,rank) in [(4000,200,100),(10000,200,200),(2000,50,50)]:
    A = poly_mat(m,n,rank,rng,cond=1e6) if rank<n else rng.standard_normal((m,n))
    Ad = cm_from_numpy(A)
    r = drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, want_sketch=True, timing=True)
    sk = cm_to_numpy(r['sketch'])
    o = oracle.cqrrpt(A, sk, np.finfo(float).eps**0.85)
    Q = cm_to_numpy(Ad); R = cm_to_numpy(r['R']); J = r['J'].cpu().numpy(); k = r['rank']
    print(f"cqrrpt {m}x{n}: rc {r['rc']}/{o['rc']} rank {k}/{o['rank']} J equal {np.array_equal(J,o['J'])} |R-Ro| {np.abs(R[:k]-o['R'][:k]).max()/np.abs(o['R']).max():.2e} ||AP-QR|| {np.linalg.norm(A[:,J-1]-Q[:,:k]@R[:k])/np.linalg.norm(A):.2e} orth {np.linalg.norm(Q[:,:k].T@Q[:,:k]-np.eye(k)):.2e} times {r['times_us']}", flush=True)
m,n = 1048576, 1024
A = cm_empty(m,n)
for it in range(2):
    ctx.fill_dense(A, m, n, key=(3,0)); ctx.sync()
    t0=time.time(); r = drv_cqrrpt(ctx, A, m, n, 1.25, 4, timing=True); ctx.sync(); dt=time.time()-t0
    fl = 2*4*m*n + (2*1280*n*n - 2/3*n**3) + 3*m*n*n + 4/3*n**3
    print(f'C3 cqrrpt: {dt*1e3:.1f} ms rank={r["rank"]} -> {fl/dt/1e12:.1f} TFLOP/s times(us) {r["times_us"]}', flush=True)
