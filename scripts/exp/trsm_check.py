import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n) in ((1048576, 1024), (262144, 1024), (1048576, 512)):
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(3, 0)); U = d.cm_empty(n, n); ctx.fill_dense(U, n, n, key=(2, 0))
    ctx.lib.rlhip_add_diag_f64(ctx.h, n, 40.0, U.data_ptr(), n)
    Um = torch.triu(U.T)          # U stored (n, n) column-major: U.T[i, j] = U_ij
    ref = None
    for it in range(6):
        B = A.clone()
        ctx.trsm(m, n, 1.0, U, n, B, m); ctx.sync()
        # residual: X U - A, X = B.T (m x n)
        R = (Um.T @ B) - A        # (n, m): column c of (X U) = sum_k X[:,k] U[k,c]
        err = R.abs().amax(dim=1)  # per column
        bad = (err > 1e-9).nonzero().flatten()
        same = True if ref is None else bool(torch.equal(ref, B))
        if ref is None: ref = B.clone()
        msg = f"inplace m={m} n={n} it={it} maxerr={float(err.max()):.3e} badcols={bad[:8].tolist()} nbad={bad.numel()} same_as_first={same}"
        if bad.numel():
            c = int(bad[0]); rows = (R[c].abs() > 1e-9).nonzero().flatten()
            msg += f" rows[{rows.numel()}]: {rows[:6].tolist()} .. {rows[-3:].tolist()}"
        print(msg, flush=True)
    ldw = m + 32
    W = torch.empty((n, ldw), dtype=torch.float64, device="cuda"); Jp = torch.randperm(n, device="cuda") + 1
    ref = None
    for it in range(6):
        W.zero_()
        ctx.trsm_gather(m, n, 1.0, U, n, A, m, Jp, W, ldw); ctx.sync()
        X = W[:, :m]
        R = (Um.T @ X) - A[Jp - 1]
        err = R.abs().amax(dim=1)
        bad = (err > 1e-9).nonzero().flatten()
        same = True if ref is None else bool(torch.equal(ref, X))
        if ref is None: ref = X.clone()
        msg = f"oop     m={m} n={n} it={it} maxerr={float(err.max()):.3e} badcols={bad[:8].tolist()} nbad={bad.numel()} same_as_first={same}"
        if bad.numel():
            c = int(bad[0]); rows = (R[c].abs() > 1e-9).nonzero().flatten()
            msg += f" rows[{rows.numel()}]: {rows[:6].tolist()} .. {rows[-3:].tolist()}"
        print(msg, flush=True)
    del A, B, W, R, ref
