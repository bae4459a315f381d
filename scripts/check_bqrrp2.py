import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, scipy.linalg as sl, scipy.linalg.lapack as ll
from randlapack_amd.device import *
from _gen import poly_mat
import oracle
ctx = Context(0); lib = ctx.lib
rng = np.random.default_rng(0)
# geqrf / ungqr vs LAPACK
for (m,n) in [(60,12),(500,64),(2000,256),(300,300),(64,200),(512,4096)]:
    A = rng.standard_normal((m,n)); Ad = cm_from_numpy(A); k=min(m,n)
    tau = torch.zeros(k, dtype=torch.float64, device='cuda')
    rc = lib.rlhip_geqrf_f64(ctx.h, m, n, Ad.data_ptr(), m, tau.data_ptr()); ctx.sync()
    qr_, tau_, _, info = ll.dgeqrf(A)
    msg = f'geqrf {m}x{n} rc={rc} QR diff {np.abs(cm_to_numpy(Ad)-qr_).max()/np.abs(qr_).max():.2e} tau {np.abs(tau.cpu().numpy()-tau_).max():.2e}'
    if m >= n:
        rc = lib.rlhip_ungqr_f64(ctx.h, m, n, n, Ad.data_ptr(), m, tau.data_ptr()); ctx.sync()
        Q = cm_to_numpy(Ad); Qo = ll.dorgqr(qr_, tau_)[0]
        msg += f' ungqr rc={rc} Q diff {np.abs(Q-Qo).max():.2e} orth {np.linalg.norm(Q.T@Q-np.eye(n)):.2e}'
    print(msg, flush=True)
# HQRQ / PLUL through drv_stab
for (m,k) in [(1000,200),(4096,256)]:
    Y = rng.standard_normal((m,k)); Yd = cm_from_numpy(Y)
    rc,_ = drv_stab(ctx, 1, Yd, m, k); Q = cm_to_numpy(Yd)
    print(f'HQRQ {m}x{k} rc {rc} orth {np.linalg.norm(Q.T@Q-np.eye(k)):.2e} vs oracle {np.abs(Q-oracle.stab(1,Y)[1]).max():.2e}' if hasattr(oracle,'stab') else '', flush=True)
    Yd = cm_from_numpy(Y); rc,_ = drv_stab(ctx, 2, Yd, m, k); L = cm_to_numpy(Yd)
    print(f'PLUL {m}x{k} rc {rc} max|L| {np.abs(L).max():.3f} vs oracle {np.abs(L-oracle.stab(2,Y)[1]).max():.2e}', flush=True)
def verify(A, Aout, tau, J, name, o=None):
    m,n = A.shape; mn = min(m,n)
    Qf = oracle.ungqr(Aout, tau); R = np.triu(Aout)[:mn]
    e1 = np.linalg.norm(A[:, J-1] - Qf@R)/np.linalg.norm(A); e2 = np.linalg.norm(Qf.T@Qf - np.eye(mn))
    msg = f'{name}: resid {e1:.2e} orth {e2:.2e}'
    if o is not None:
        rk = o['rank']
        msg += f" | J equal {np.array_equal(J,o['J'])} J[:rank] {np.array_equal(J[:rk],o['J'][:rk])} rank {o['rank']} |R-Ro| {np.abs(np.abs(R)-np.abs(np.triu(o['A'])[:mn])).max()/np.abs(R).max():.2e} tau diff {np.abs(tau-o['tau']).max():.2e}"
    print(msg, flush=True)
for (m,n,b,kind) in [(500,200,50,'poly'),(5000,2000,500,'gauss'),(300,500,64,'poly'),(400,150,40,'lowrank'),(1024,1024,128,'step')]:
    if kind=='poly': A = poly_mat(m,n,min(m,n),rng,cond=1e4)
    elif kind=='lowrank': A = poly_mat(m,n,60,rng,cond=1e3)
    elif kind=='step':
        s = np.ones(n); s[n//2:] = 1e-10; A = (np.linalg.qr(rng.standard_normal((m,n)))[0]*s)@np.linalg.qr(rng.standard_normal((n,n)))[0].T
    else: A = rng.standard_normal((m,n))
    for (qw,qt,ap) in [(0,1,1),(0,2,0),(0,0,1),(1,0,0),(0,1,0)]:
        Ad = cm_from_numpy(A)
        t0=time.time(); r = drv_bqrrp(ctx, Ad, m, n, b, 1.0, want_sketch=True, timing=True, qrcp_wide=qw, qr_tall=qt, apply_trans_q=ap); ctx.sync(); dt=time.time()-t0
        o = oracle.bqrrp(A, b, 1.0, qrcp_wide=qw, qr_tall=qt, apply_trans_q=ap, sketch=cm_to_numpy(r['sketch']))
        print(f'bqrrp {m}x{n} b={b} {kind} opts {(qw,qt,ap)}: rc {r["rc"]} rank {r["rank"]}/{o["rank"]} t={dt*1e3:.1f}ms')
        verify(A, cm_to_numpy(Ad), r['tau'].cpu().numpy(), r['J'].cpu().numpy(), '   device', o)
m = n = 16384; b = 512
A = cm_empty(m, n)
for (qw,qt,ap) in [(0,1,1),(1,1,1)]:
    ctx.fill_dense(A, m, n, key=(4,0)); ctx.sync()
    r = drv_bqrrp(ctx, A, m, n, b, 1.0, qrcp_wide=qw, qr_tall=qt, apply_trans_q=ap); ctx.sync()
    ctx.fill_dense(A, m, n, key=(4,0)); ctx.sync()
    t0=time.time(); r = drv_bqrrp(ctx, A, m, n, b, 1.0, timing=True, qrcp_wide=qw, qr_tall=qt, apply_trans_q=ap); ctx.sync(); dt=time.time()-t0
    fl = 2*b*m*n + 2*m*n*n - 2*n**3/3
    print(f'bqrrp {m}x{n} b={b} opts {(qw,qt,ap)}: {dt*1e3:.1f} ms -> {fl/dt/1e12:.1f} TFLOP/s rank {r["rank"]} times {r["times_us"]}', flush=True)
