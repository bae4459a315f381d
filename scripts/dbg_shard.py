import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from randlapack_amd import device as d
import oracle
m,n,k,p = 3001,256,32,0
rng = np.random.default_rng(2024)
A = rng.standard_normal((m, n)) @ np.diag(np.linspace(1.0, 0.05, n)) @ np.linalg.qr(rng.standard_normal((n, n)))[0]
ctx = d.Context(0)
r1 = d.drv_rsvd(ctx, d.cm_from_numpy(A), m, n, k, k, 1e-12, p, 1)
ref = oracle.rsvd(A, k, k, 1e-12, p, 1)
print(r1['S'].cpu().numpy()[:6], ref['S'][:6], r1['qb_rc'], ref['qb_rc'], r1['k'], ref['k'], r1['next_ctr'], ref['next_ctr'])
Om = d.cm_empty(n,k); ctx.fill_dense(Om, n, k); ctx.sync()
Oo,_ = oracle.fill_dense(n,k)
print('omega diff', np.abs(d.cm_to_numpy(Om)-Oo).max())
A2 = np.asfortranarray(A)
ref2 = oracle.rsvd(A2, k, k, 1e-12, p, 1); print(ref2['S'][:6])
