#!/usr/bin/env python3
"""A/B of the CSR product behind rlhip_linop_apply at ABRIK's shape (200000 x 200000, 10 nonzeros per row, 32 columns, column-major blocks).
The kernel choice is read once per process, so every variant runs in its own interpreter:  python scripts/spmm_ab.py  prints one line per variant.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = r"""
import numpy as np, scipy.sparse as sp, randlapack_amd.device as d
ctx = d.Context(0)
m = n = 200000; k = 32
rng = np.random.default_rng(77)
rows = np.repeat(np.arange(m), 10); cols = (rows + np.tile(np.arange(-4, 6), m)) % n
G = sp.csr_matrix((rng.standard_normal(m * 10), (rows, cols)), shape=(m, n)); G.sum_duplicates()
Gr = sp.random(m, n, 10.0 / n, random_state=np.random.default_rng(5), format="csr")
for name, S in (("banded", G), ("uniform", Gr)):
    op = d.CsrOperator.from_scipy(S)
    X = d.cm_empty(n, k); ctx.fill_dense(X, n, k, key=(9, 0))
    Y = d.linop_apply(ctx, op, "L", "N", X, m, k, n); ctx.sync()
    best = 1e9
    for rep in range(3):
        ctx.timer_start()
        for _ in range(20): d.linop_apply(ctx, op, "L", "N", X, m, k, n, C_in=Y)
        best = min(best, ctx.timer_stop_ms() / 20)
    nnz = S.nnz
    ba = (nnz * k + m * k) * 8.0 + 16.0 * nnz
    print(f"{name:8s} nnz={nnz} {best*1e3:7.1f} us per product = {ba / best / 1e6:7.1f} GB/s algorithmic = {ba / best / 8e9:.3f} of HBM peak")
"""
for tag, env in (("default (narrow + column-major store)", {}), ("RLHIP_SPMM_CMOUT=0 (narrow, transposed store)", {"RLHIP_SPMM_CMOUT": "0"}),
                 ("RLHIP_SPMM_NARROW=0 (wavefront per row)", {"RLHIP_SPMM_NARROW": "0"})):
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    r = subprocess.run([sys.executable, "-c", ONE], env=e, capture_output=True, text=True, cwd=ROOT)
    print(tag); print(r.stdout.rstrip() or r.stderr[-2000:])
