import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd.device import Context, cm_from_numpy, cm_to_numpy, cm_empty, cm_zeros
ctx = Context(0)
res = {}
res['mfma_f64_tflops'] = ctx.mfma_peak(True, 20000)
res['mfma_f32_tflops'] = ctx.mfma_peak(False, 20000)
big = torch.empty(4 << 30, dtype=torch.uint8, device='cuda:0')
big.zero_()
res['hbm_read_gbps'] = ctx.hbm_read_peak(big)
del big
print(res, flush=True)
# philox KAT
kat = ctx.philox(1, (0,0,0,0), (0,0)); print('philox0', [hex(x) for x in kat])
kat = ctx.philox(1, (0xffffffff,)*4, (0xffffffff,)*2); print('philoxF', [hex(x) for x in kat])
kat = ctx.philox(1, (0x243f6a88,0x85a308d3,0x13198a2e,0x03707344), (0xa4093822,0x299f31d0)); print('philoxPi', [hex(x) for x in kat])
rng = np.random.default_rng(0)
def chk(name, got, ref, tol):
    err = np.abs(got-ref).max()/max(1e-300, np.abs(ref).max())
    print(f'{name}: relerr {err:.3e}', 'OK' if err < tol else 'FAIL', flush=True)
# gemm all layouts, odd sizes
for (m,n,k) in [(300,70,45),(257,256,130),(128,256,64),(1000,17,33),(64,300,1000),(5,3,2), (513, 129, 4000)]:
    for ta in 'NT':
        for tb in 'NT':
            A = rng.standard_normal((m,k)); B = rng.standard_normal((k,n)); C0 = rng.standard_normal((m,n))
            Ad = cm_from_numpy(A if ta=='N' else A.T.copy()); Bd = cm_from_numpy(B if tb=='N' else B.T.copy()); Cd = cm_from_numpy(C0)
            lda = m if ta=='N' else k; ldb = k if tb=='N' else n
            ctx.gemm(ta,tb,m,n,k,1.5,Ad,lda,Bd,ldb,-0.5,Cd,m)
            chk(f'gemm{ta}{tb} {m}x{n}x{k}', cm_to_numpy(Cd), 1.5*A@B-0.5*C0, 1e-13)
# syrk
for (n,k) in [(256,5000),(100,300),(300,2000)]:
    A = rng.standard_normal((k,n)); Ad = cm_from_numpy(A); Cd = cm_zeros(n,n)
    ctx.syrk('U','T',n,k,1.0,Ad,k,0.0,Cd,n)
    chk(f'syrk {n} {k}', np.triu(cm_to_numpy(Cd)), np.triu(A.T@A), 1e-13)
# potrf
for n in [5, 32, 100, 256, 700]:
    X = rng.standard_normal((2*n,n)); G = X.T@X; Gd = cm_from_numpy(G)
    info = ctx.potrf(n, Gd, n); R = np.triu(cm_to_numpy(Gd))
    chk(f'potrf {n} info={info}', R.T@R, G, 1e-13)
G = np.eye(40); G[17,17] = -1.0; Gd = cm_from_numpy(G); print('potrf fail info', ctx.potrf(40, Gd, 40))
# trsm / trmm
for (m,n) in [(1000,256),(333,100),(2000,600),(70,5)]:
    U = np.triu(rng.standard_normal((n,n))) + 5*np.eye(n); B = rng.standard_normal((m,n))
    Ud = cm_from_numpy(U); Bd = cm_from_numpy(B)
    ctx.trsm(m,n,2.0,Ud,n,Bd,m); X = cm_to_numpy(Bd)
    chk(f'trsm {m} {n}', X@U, 2.0*B, 1e-12)
    Bd = cm_from_numpy(B); ctx.trmm(m,n,2.0,Ud,n,Bd,m)
    chk(f'trmm {m} {n}', cm_to_numpy(Bd), 2.0*B@U, 1e-13)
# lange
A = rng.standard_normal((1234,77)); print('lange', ctx.lange_fro(1234,77,cm_from_numpy(A),1234), np.linalg.norm(A))
# gesvdj
for (m,n) in [(300,64),(256,256),(2000,255),(50,7)]:
    A = rng.standard_normal((m,n)) @ np.diag(np.logspace(0,-8,n)) @ np.linalg.qr(rng.standard_normal((n,n)))[0]
    A = np.linalg.qr(A)[1].T.copy() if m==n else A
    Ad = cm_from_numpy(A); S = torch.empty(n, dtype=torch.float64, device='cuda'); VT = cm_empty(n,n)
    t0=time.time(); info, sw = ctx.gesvdj(m,n,Ad,m,S,VT,n); ctx.sync(); dt=time.time()-t0
    U = cm_to_numpy(Ad); s = S.cpu().numpy(); vt = cm_to_numpy(VT)
    sref = np.linalg.svd(A, compute_uv=False)
    print(f'gesvdj {m}x{n} info={info} sweeps={sw} t={dt*1e3:.1f}ms recon={np.abs(U*s@vt-A).max()/np.abs(A).max():.2e} srel={np.max(np.abs(s-sref)/sref):.2e} orthU={np.abs(U.T@U-np.eye(n)).max():.2e}', flush=True)
# fill_dense
buf = cm_empty(1000, 300); nxt = ctx.fill_dense(buf, 1000, 300); g = cm_to_numpy(buf)
print('fill mean/std', g.mean(), g.std(), 'next', nxt, 'first', g[:4,0])
# big gemm timing: A*Omega and A^T*Q
m, n, k = 100000, 20000, 256
A = cm_empty(m, n); ctx.fill_dense(A, m, n, key=(1,0))
Om = cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(2,0))
Y = cm_empty(m, k); BT = cm_empty(n, k)
for name, fn, fl in [('A*Om', lambda: ctx.gemm('N','N',m,k,n,1.0,A,m,Om,n,0.0,Y,m), 2.0*m*n*k), ('At*Y', lambda: ctx.gemm('T','N',n,k,m,1.0,A,m,Y,m,0.0,BT,n), 2.0*m*n*k), ('syrk', lambda: ctx.syrk('U','T',k,m,1.0,Y,m,0.0,cm_zeros(k,k),k), 1.0*m*k*k)]:
    fn(); ctx.sync()
    ctx.timer_start(); 
    for _ in range(3): fn()
    ms = ctx.timer_stop_ms()/3
    print(f'{name}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s', flush=True)
# verify big gemm on a sample of rows
Ah = A[:, :512].cpu().numpy().T  # rows 0..511 -> (512, n)
Yh = Y[:, :512].cpu().numpy().T
print('bigY relerr', np.abs(Ah@Om.cpu().numpy().T - Yh).max()/np.abs(Yh).max())
