"""Where ABRIK's time goes on tall operators: the panel QR route (qr_exp) on a graded sparse operator and on a dense Gaussian one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from randlapack_amd import device as d
ctx = d.Context(0)
m = n = 200000; k = 32
rng = np.random.default_rng(77)
rows = np.repeat(np.arange(m), 10); cols = (rows + np.tile(np.arange(-4, 6), m)) % n
vals = rng.standard_normal(m * 10)
for name, d1 in (("graded", np.exp(-np.arange(m) / 4.0) + 1e-13), ("flat", np.ones(m))):
    G = sp.csr_matrix((vals * d1[rows] * d1[cols], (rows, cols)), shape=(m, n)); G.sum_duplicates()
    op = d.CsrOperator.from_scipy(G)
    for qr_exp in (-1, 0, 1):
        r = d.drv_abrik_linop(ctx, op, k, 1e-13, 8, key=(2, 0), qr_exp=qr_exp, timing=True)
        r = d.drv_abrik_linop(ctx, op, k, 1e-13, 8, key=(2, 0), qr_exp=qr_exp, timing=True)
        t = dict(zip(d.ABRIK_TIMES, r["times_us"]))
        print(f"sparse {name} qr_exp={qr_exp}: iters {r['iters']} triplets {r['triplets']} total {t['total']/1e3:.1f} ms  qr {t['qr']/1e3:.1f}  ungqr {t['ungqr']/1e3:.1f}  reorth {t['reorth']/1e3:.1f}  gemm_A {t['gemm_A']/1e3:.1f}  get_factors {t['get_factors']/1e3:.1f}", flush=True)
A = d.cm_empty(200000, 20000); ctx.fill_dense(A, 200000, 20000, key=(7, 0))
op = d.DenseOperator(A, 200000, 20000)
for qr_exp in (-1, 1):
    r = d.drv_abrik_linop(ctx, op, k, 1e-13, 8, key=(2, 0), qr_exp=qr_exp, timing=True)
    t = dict(zip(d.ABRIK_TIMES, r["times_us"]))
    print(f"dense qr_exp={qr_exp}: iters {r['iters']} total {t['total']/1e3:.1f} ms  qr {t['qr']/1e3:.1f}  ungqr {t['ungqr']/1e3:.1f}  reorth {t['reorth']/1e3:.1f}  gemm_A {t['gemm_A']/1e3:.1f}  get_factors {t['get_factors']/1e3:.1f}", flush=True)
