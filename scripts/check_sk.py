import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
rng = np.random.default_rng(0)
def chk(name, got, ref, tol=1e-12):
    err = np.abs(got-ref).max()/np.abs(ref).max()
    print(f'{name}: relerr {err:.3e}', 'OK' if err < tol else 'FAIL', flush=True)
for (m,n,k,ta) in [(1280,256,32768,'N'),(1408+37,256,32768+0,'N'),(1280,512,16384,'N'),(256,256,131072,'T'),(640+5,256,65536,'T'),(128*300,256,1024,'N')]:
    A = rng.standard_normal((m,k)); B = rng.standard_normal((k,n)); C0 = rng.standard_normal((m,n))
    Ad = cm_from_numpy(A if ta=='N' else A.T.copy()); Bd = cm_from_numpy(B); Cd = cm_from_numpy(C0)
    ctx.gemm(ta,'N',m,n,k,1.5,Ad,m if ta=='N' else k,Bd,k,-0.5,Cd,m)
    r1 = cm_to_numpy(Cd)
    chk(f'sk gemm {ta}N {m}x{n}x{k}', r1, 1.5*A@B-0.5*C0)
    Cd = cm_from_numpy(C0); ctx.gemm(ta,'N',m,n,k,1.5,Ad,m if ta=='N' else k,Bd,k,-0.5,Cd,m)
    print('   deterministic', np.array_equal(r1, cm_to_numpy(Cd)))
m,n,k = 200000, 20000, 256
A = cm_empty(m,n); ctx.fill_dense(A, m, n, key=(7,0))
Om = cm_empty(n,k); ctx.fill_dense(Om, n, k, key=(8,0))
Y = cm_empty(m,k); BT = cm_empty(n,k)
for name, fn in [('A*Om', lambda: ctx.gemm('N','N',m,k,n,1.0,A,m,Om,n,0.0,Y,m)), ('At*Y', lambda: ctx.gemm('T','N',n,k,m,1.0,A,m,Y,m,0.0,BT,n))]:
    fn(); ctx.sync()
    ctx.timer_start()
    for _ in range(3): fn()
    ms = ctx.timer_stop_ms()/3
    print(f'{name}: {ms:.2f} ms  {2.0*m*n*k/ms/1e9:.1f} TFLOP/s', flush=True)
# verify big results on samples vs generic kernel
os.environ['X']='1'
Ah = A[:, :300].cpu().numpy().T; Yh = Y[:, :300].cpu().numpy().T
print('bigY relerr', np.abs(Ah@Om.cpu().numpy().T - Yh).max()/np.abs(Yh).max())
Ah = A[:, -70:].cpu().numpy().T; Yh = Y[:, -70:].cpu().numpy().T
print('bigY tail relerr', np.abs(Ah@Om.cpu().numpy().T - Yh).max()/np.abs(Yh).max())
cols = torch.arange(0, n, 501, device='cuda')
ref = (A[cols].double() @ Y.T)  # (len, m)@(m,k)
print('bigBT relerr', float((ref - BT[:, cols].T).abs().max() / ref.abs().max()))
ref = (A[-40:].double() @ Y.T); print('bigBT tail relerr', float((ref - BT[:, -40:].T).abs().max()/ref.abs().max()))
