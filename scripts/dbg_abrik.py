import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from randlapack_amd.device import *
from _gen import poly_mat
import oracle
ctx = Context(0)
rng = np.random.default_rng(0)
m,n,k = 400,300,8
A = poly_mat(m,n,min(m,n),rng,cond=1e6)
sref = np.linalg.svd(A, compute_uv=False)
for iters in (4,5,6):
    r = drv_abrik(ctx, cm_from_numpy(A), m, n, k, 1e-12, iters, key=(1,0)); o = oracle.abrik(A, k, 1e-12, iters, key=(1,0))
    S = r['S'].cpu().numpy(); t=r['triplets']
    U,V = cm_to_numpy(r['U']), cm_to_numpy(r['V'])
    print('iters', iters, 't', t)
    print(' dev   ', np.array2string(S[:t], precision=5))
    print(' oracle', np.array2string(o['S'][:t], precision=5))
    print(' exact ', np.array2string(sref[:t], precision=5))
    print(' dev resid ||A^T U - V S||', np.linalg.norm(A.T@U - V*S), ' oracle', np.linalg.norm(A.T@o['U'] - o['V']*o['S']), ' ||AV - US|| dev', np.linalg.norm(A@V-U*S), 'oracle', np.linalg.norm(A@o['V']-o['U']*o['S']))
