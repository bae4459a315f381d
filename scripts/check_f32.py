import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from randlapack_amd.device import *
from _gen import poly_mat
import oracle
ctx = Context(0)
rng = np.random.default_rng(0)
def cm32(a): return torch.from_numpy(np.ascontiguousarray(a.T.astype(np.float32))).cuda()
def verify(A, Aout, tau, J):
    m,n = A.shape; mn=min(m,n)
    Qf = oracle.ungqr(Aout.astype(np.float64), tau.astype(np.float64)); R = np.triu(Aout)[:mn].astype(np.float64)
    return np.linalg.norm(A[:, J-1]-Qf@R)/np.linalg.norm(A), np.linalg.norm(Qf.T@Qf-np.eye(mn))
for (m,n,b) in [(1000,400,100),(2048,2048,256),(600,900,128)]:
    A = rng.standard_normal((m,n)).astype(np.float32).astype(np.float64)
    for opts in [(0,1,1),(1,1,1),(0,2,0)]:
        Ad = cm32(A)
        r = drv_bqrrp(ctx, Ad, m, n, b, 1.0, qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2], key=(3,0))
        e1,e2 = verify(A, cm_to_numpy(Ad), r['tau'].cpu().numpy(), r['J'].cpu().numpy())
        print(f'bqrrp f32 {m}x{n} b={b} opts {opts}: rc {r["rc"]} rank {r["rank"]} resid {e1:.2e} orth {e2:.2e}', flush=True)
A = poly_mat(20000, 256, 256, rng, cond=1e3).astype(np.float32).astype(np.float64)
Ad = cm32(A); r = drv_cqrrpt(ctx, Ad, 20000, 256, 1.25, 4, key=(1,0)); k=r['rank']
Q = cm_to_numpy(Ad)[:, :k].astype(np.float64); R = cm_to_numpy(r['R'])[:k].astype(np.float64); J = r['J'].cpu().numpy()
print(f'cqrrpt f32 rank {k} resid {np.linalg.norm(A[:,J-1]-Q@R)/np.linalg.norm(A):.2e} orth {np.linalg.norm(Q.T@Q-np.eye(k)):.2e}', flush=True)
Ad = cm32(A); r = drv_rsvd(ctx, Ad, 20000, 256, 32, 32, 1e-5, 2, 1); U,S,V = cm_to_numpy(r['U']).astype(np.float64), r['S'].cpu().numpy().astype(np.float64), cm_to_numpy(r['V']).astype(np.float64)
sref = np.linalg.svd(A, compute_uv=False)[:32]
print(f'rsvd f32 k {r["k"]} qb {r["qb_rc"]} S relerr {np.max(np.abs(S-sref)/sref):.2e} orthU {np.linalg.norm(U.T@U-np.eye(32)):.2e}', flush=True)
Ad = cm32(A[:2000]); r = drv_hqrrp(ctx, Ad, 2000, 256, 32, 8); e1,e2 = verify(A[:2000], cm_to_numpy(Ad), r['tau'].cpu().numpy(), r['J'].cpu().numpy()); print(f'hqrrp f32 resid {e1:.2e} orth {e2:.2e}')
# C4-like single GPU: 32768^2 fp32, b = 2048
m = n = 32768; b = 2048
A = cm_empty(m, n, dtype=torch.float32); ctx.fill_dense(A, m, n, key=(4,0)); ctx.sync()
for opts in [(0,1,1)]:
    r = drv_bqrrp(ctx, A, m, n, b, 1.0, qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2]); ctx.sync()
    ctx.fill_dense(A, m, n, key=(4,0)); ctx.sync()
    t0=time.time(); r = drv_bqrrp(ctx, A, m, n, b, 1.0, timing=True, qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2]); ctx.sync(); dt=time.time()-t0
    fl = 2*b*m*n + 2*m*n*n - 2*n**3/3
    print(f'bqrrp f32 {m}x{n} b={b} opts {opts}: {dt*1e3:.1f} ms -> {fl/dt/1e12:.1f} TFLOP/s rank {r["rank"]} times {r["times_us"]}', flush=True)
