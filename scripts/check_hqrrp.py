import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from randlapack_amd.device import *
from _gen import poly_mat
import oracle
ctx = Context(0); lib = ctx.lib
rng = np.random.default_rng(0)
def verify(A, Aout, tau, J):
    m,n = A.shape; mn = min(m,n)
    Qf = oracle.ungqr(Aout, tau); R = np.triu(Aout)[:mn]
    return np.linalg.norm(A[:, J-1] - Qf@R)/np.linalg.norm(A), np.linalg.norm(Qf.T@Qf - np.eye(mn))
for (m,n,nb,pp) in [(300,120,32,5),(200,200,64,10),(150,260,32,8),(500,70,16,4),(1280,1024,64,10)]:
    A = poly_mat(m,n,min(m,n),rng,cond=1e4)
    for (qt,ppv) in [(0,1),(0,0),(1,0),(2,0)]:
        Ad = cm_from_numpy(A)
        ctx.sync(); t0=time.time(); r = drv_hqrrp(ctx, Ad, m, n, nb, pp, ppv, qt, key=(7,0), want_G=True); ctx.sync(); dt=time.time()-t0
        o = oracle.hqrrp(A, nb, pp, ppv, qt, key=(7,0), G=cm_to_numpy(r['G']))
        Aout = cm_to_numpy(Ad); tau = r['tau'].cpu().numpy(); J = r['J'].cpu().numpy()
        e1,e2 = verify(A, Aout, tau, J)
        mn=min(m,n)
        print(f'hqrrp {m}x{n} nb={nb} qr_type={qt} pp={ppv}: rc {r["rc"]}/{o["rc"]} t={dt*1e3:.1f}ms resid {e1:.2e} orth {e2:.2e} | J equal {np.array_equal(J,o["J"])} |R-Ro| {np.abs(np.abs(np.triu(Aout)[:mn])-np.abs(np.triu(o["A"])[:mn])).max()/np.abs(Aout).max():.2e} ctr {r["next_ctr"]==o["next_ctr"]}', flush=True)
# CQRRPT with the three QRCP choices
m,n = 20000, 256
A = poly_mat(m,n,n,rng,cond=1e6)
for qrcp in (2,0,1):
    Ad = cm_from_numpy(A)
    r = drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, want_sketch=True, key=(11,0), qrcp=qrcp)
    # the inner QRCP continues from the state after the SASO: hand the oracle the same one
    o = oracle.cqrrpt(A, cm_to_numpy(r['sketch']), float(np.finfo(float).eps**0.85), qrcp=qrcp, ctr=r.get('ctr_after_saso',(0,0,0,0)), key=(11,0))
    k = r['rank']; Q = cm_to_numpy(Ad)[:, :k]; R = cm_to_numpy(r['R'])[:k]; J = r['J'].cpu().numpy()
    print(f'cqrrpt qrcp={qrcp}: rc {r["rc"]} rank {k}/{o["rank"]} resid {np.linalg.norm(A[:,J-1]-Q@R)/np.linalg.norm(A):.2e} orth {np.linalg.norm(Q.T@Q-np.eye(k)):.2e} J equal {np.array_equal(J,o["J"])}', flush=True)
