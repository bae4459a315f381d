import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from randlapack_amd import device as d
ctx = d.Context(0)
m = n = 200000; k = 32
rng = np.random.default_rng(77)
rows = np.repeat(np.arange(m), 10); cols = (rows + np.tile(np.arange(-4, 6), m)) % n
G = sp.csr_matrix((rng.standard_normal(m * 10), (rows, cols)), shape=(m, n)); G.sum_duplicates()
op = d.CsrOperator.from_scipy(G)
X = d.cm_empty(n, k); ctx.fill_dense(X, n, k, key=(9, 0))
Y = d.linop_apply(ctx, op, "L", "N", X, m, k, n)
for _ in range(5): d.linop_apply(ctx, op, "L", "N", X, m, k, n, C_in=Y)
for _ in range(5): d.linop_apply(ctx, op, "L", "T", Y, n, k, m, C_in=X)
ctx.sync()
