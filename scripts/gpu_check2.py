import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd.device import *
import oracle
ctx = Context(0)
rng = np.random.default_rng(1)
def relsub(U1, U2):  # subspace distance
    return np.linalg.norm(U1 - U2 @ (U2.T @ U1))
for (m,n,k,b,p,q) in [(4096,512,64,64,2,1),(1000,300,40,10,2,1),(500,200,50,50,0,1),(2000,400,32,16,1,1)]:
    # polynomial-decay matrix
    U0 = np.linalg.qr(rng.standard_normal((m,n)))[0]; V0 = np.linalg.qr(rng.standard_normal((n,n)))[0]
    s0 = np.ones(n); t = np.arange(n - n//10); s0[n//10:] = (1.0/(1+t))**2 * 1.0 + 1e-6
    A = (U0*s0)@V0.T
    Ad = cm_from_numpy(A)
    tol = np.finfo(float).eps**0.5625
    t0=time.time(); r = drv_rsvd(ctx, Ad, m, n, k, b, tol, p, q); ctx.sync(); tg=time.time()-t0
    t0=time.time(); o = oracle.rsvd(A, k, b, tol, p, q); tc=time.time()-t0
    U = cm_to_numpy(r['U']); S = r['S'].cpu().numpy(); V = cm_to_numpy(r['V'])
    recon_g = np.linalg.norm(A-(U*S)@V.T)/np.linalg.norm(A); recon_o = np.linalg.norm(A-(o['U']*o['S'])@o['V'].T)/np.linalg.norm(A)
    print(f"rsvd {m}x{n} k={k} b={b} p={p}: gpu rc={r['rc']} qb={r['qb_rc']} k={r['k']} | cpu qb={o['qb_rc']} k={o['k']} | recon gpu {recon_g:.3e} cpu {recon_o:.3e} | sigma rel {np.max(np.abs(S-o['S'])/o['S']):.2e} | orthU {np.linalg.norm(U.T@U-np.eye(r['k'])):.2e} orthV {np.linalg.norm(V.T@V-np.eye(r['k'])):.2e} | next {r['next_ctr']} {o['next_ctr']} | t gpu {tg*1e3:.1f}ms cpu {tc*1e3:.1f}ms", flush=True)
# big: config 2 scaled (m=100000)
m,n,k = 200000, 20000, 256
A = cm_empty(m,n); ctx.fill_dense(A, m, n, key=(7,0)); ctx.sync()
for it in range(3):
    torch.cuda.synchronize(); t0=time.time()
    r = drv_rsvd(ctx, A, m, n, k, k, 1e-12, 0, 1, key=(0,0)); torch.cuda.synchronize(); dt=time.time()-t0
    fl = 2.0*m*n*k*2 + 4.0*m*k*k
    print(f'C2 rsvd: {dt*1e3:.1f} ms -> {fl/dt/1e12:.1f} TFLOP/s, qb={r["qb_rc"]} k={r["k"]} S0={float(r["S"][0]):.3f} S255={float(r["S"][-1]):.3f}', flush=True)
U=r['U']; print('orthU big', float(torch.linalg.norm(U@U.T - torch.eye(k,device='cuda',dtype=torch.float64))))
