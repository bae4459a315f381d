"""Timing of the sparse-operator kernels and the linop QR drivers at benchmark scale (tall sparse A, bench_CQRRT_linops sizes
scaled to one MI355X).  usage: python scripts/linops_time.py [m n nnz_per_row]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys
import time

import numpy as np
import torch

from randlapack_amd import device as d

m, n, r = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1_000_000, 1000, 10)
ctx = d.Context(0)
rng = np.random.default_rng(0)
cols = np.sort(rng.integers(0, n, size=(m, r)), axis=1).astype(np.int64).ravel()
vals = rng.standard_normal(m * r)
rowptr = (np.arange(m + 1, dtype=np.int64) * r)
op = d.CsrOperator(m, n, torch.as_tensor(rowptr, device="cuda:0"), torch.as_tensor(cols, device="cuda:0"), torch.as_tensor(vals, device="cuda:0"))
nnz = m * r


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for b in (64, 256, 1000):
    if b > n:
        continue
    M = d.cm_from_numpy(rng.standard_normal((n, b)))
    X = d.cm_zeros(m, b)
    t = timed(lambda: d.linop_apply(ctx, op, "L", "N", M, m, b, n, C_in=X))
    byts = (nnz * b + m * b) * 8 + nnz * 16
    print(f"fwd  A*M   b={b:5d}: {t*1e3:8.2f} ms  alg {byts/1e9:7.2f} GB -> {byts/t/1e9:7.0f} GB/s   (includes transposes + CSR-transpose build)")
    Y = d.cm_zeros(n, b)
    t = timed(lambda: d.linop_apply(ctx, op, "L", "T", X, n, b, m, C_in=Y))
    byts = (nnz * b + n * b) * 8 + nnz * 16
    print(f"adj  A^T*X b={b:5d}: {t*1e3:8.2f} ms  alg {byts/1e9:7.2f} GB -> {byts/t/1e9:7.0f} GB/s")

# raw kernels without the operator construction (the transpose CSR is built once per operator in real use)
import ctypes as C
rpt = torch.zeros(n + 1, dtype=torch.int64, device="cuda:0")
cit = torch.zeros(nnz, dtype=torch.int64, device="cuda:0")
vt = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
t0 = time.perf_counter()
ctx.lib.rlhip_csr_transpose_f64(ctx.h, m, n, op.rowptr.data_ptr(), op.colidx.data_ptr(), op.vals.data_ptr(), rpt.data_ptr(), cit.data_ptr(), vt.data_ptr())
print(f"csr_transpose (host staged): {(time.perf_counter()-t0)*1e3:.0f} ms")
for b in (64, 256, 1000):
    if b > n:
        continue
    for lay in ("R", "C"):
        Bf = torch.randn(n * b, dtype=torch.float64, device="cuda:0")
        Cf = torch.zeros(m * b, dtype=torch.float64, device="cuda:0")
        ldb, ldc = (b, b) if lay == "R" else (n, m)
        t = timed(lambda: ctx.lib.rlhip_csr_spmm_f64(ctx.h, lay.encode(), m, b, n, 1.0, op.rowptr.data_ptr(), op.colidx.data_ptr(), op.vals.data_ptr(),
                                                     Bf.data_ptr(), ldb, 0.0, Cf.data_ptr(), ldc))
        byts = (nnz * b + m * b) * 8 + nnz * 16
        print(f"kernel fwd layout {lay} b={b:5d}: {t*1e3:8.2f} ms -> {byts/t/1e9:7.0f} GB/s algorithmic")
        Bt = torch.randn(m * b, dtype=torch.float64, device="cuda:0")
        Ct = torch.zeros(n * b, dtype=torch.float64, device="cuda:0")
        ldb, ldc = (b, b) if lay == "R" else (m, n)
        t = timed(lambda: ctx.lib.rlhip_csr_spmm_f64(ctx.h, lay.encode(), n, b, m, 1.0, rpt.data_ptr(), cit.data_ptr(), vt.data_ptr(),
                                                     Bt.data_ptr(), ldb, 0.0, Ct.data_ptr(), ldc))
        byts = (nnz * b + n * b) * 8 + nnz * 16
        print(f"kernel adj layout {lay} b={b:5d}: {t*1e3:8.2f} ms -> {byts/t/1e9:7.0f} GB/s algorithmic")

for alg in ("cholqr", "scholqr3", "cqrrt"):
    for blk in (0, 256):
        t = timed(lambda: d.drv_qr_linops(ctx, alg, op, block_size=blk, d_factor=2.0, nnz=4), reps=2)
        print(f"{alg:10s} block={blk:4d}: {t*1e3:9.1f} ms")
