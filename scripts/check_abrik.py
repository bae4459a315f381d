import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from randlapack_amd.device import *
from _gen import poly_mat
import oracle
ctx = Context(0)
rng = np.random.default_rng(0)
for (m,n,k,iters) in [(400,300,8,2),(400,300,8,3),(400,300,8,6),(400,300,8,12),(1000,800,16,9),(300,500,4,10),(2000,2000,32,8)]:
    A = poly_mat(m,n,min(m,n),rng,cond=1e6)
    Ad = cm_from_numpy(A)
    t0=time.time(); r = drv_abrik(ctx, Ad, m, n, k, 1e-12, iters, key=(1,0)); ctx.sync(); dt=time.time()-t0
    o = oracle.abrik(A, k, 1e-12, iters, key=(1,0))
    U,S,V = cm_to_numpy(r['U']), r['S'].cpu().numpy(), cm_to_numpy(r['V'])
    t = r['triplets']; kk=min(t,k)
    sref = np.linalg.svd(A, compute_uv=False)
    print(f'abrik {m}x{n} k={k} iters {r["iters"]}/{o["iters"]} triplets {t}/{o["triplets"]} t={dt*1e3:.1f}ms S vs oracle {np.max(np.abs(S-o["S"])/o["S"][0]):.2e} S vs exact (top k) {np.max(np.abs(S[:kk]-sref[:kk])/sref[:kk]):.2e} orthU {np.linalg.norm(U.T@U-np.eye(t)):.2e} orthV {np.linalg.norm(V.T@V-np.eye(t)):.2e} normR {r["norm_R_end"]:.6e}/{o["norm_R_end"]:.6e} ctr {r["next_ctr"]==o["next_ctr"]}', flush=True)
# early termination: exactly low-rank operator
A = rng.standard_normal((300,20))@rng.standard_normal((20,200))
r = drv_abrik(ctx, cm_from_numpy(A), 300, 200, 8, 1e-12, 50, key=(2,0)); o = oracle.abrik(A, 8, 1e-12, 50, key=(2,0))
print('low-rank: iters', r['iters'], o['iters'], 'triplets', r['triplets'], o['triplets'])
