"""hqrrp with pivoted panels at a size where the tall-panel split (pivots from the QRCP of the panel's R factor) is active: residual,
orthogonality and |diag R| against the singular values, with the split on and off (argv[1] = 0: the hqrrp_tall_panel option off)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import os, sys, numpy as np, torch
from randlapack_amd import device as d
from benchmarks import _common as c
ctx = d.Context(0)
SPLIT = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx.set_option('hqrrp_tall_panel', SPLIT)
m, n, nb = 8192, 2048, 256
for m_type, kw in [("polynomial", dict(cond_num=1e10, exponent=2.0)), ("step", dict(cond_num=1e10)), ("spiked", dict(scaling=1e10)), ("gaussian", {})]:
    A0 = c.regen(ctx, m_type, m, n, key=(5, 0), **kw)
    A = A0.clone()
    r = d.drv_hqrrp(ctx, A, m, n, nb_alg=nb, pp=10, panel_pivoting=1, qr_type=0)
    J, tau = r["J"], r["tau"]
    R = torch.triu(A[:, :n].T)                       # (n, n)
    Q = A[:n].clone()
    ctx.lib.rlhip_ungqr_f64(ctx.h, m, n, n, Q.data_ptr(), m, tau.data_ptr())
    AP = A0[(J - 1).long()]
    res = float(torch.linalg.norm(AP - R.T @ Q) / torch.linalg.norm(A0))
    orth = float(torch.linalg.norm(Q @ Q.T - torch.eye(n, dtype=Q.dtype, device=Q.device)))
    S = torch.linalg.svdvals(A0.T.contiguous().cpu()).numpy()
    dr = np.abs(np.diag(R.cpu().numpy()))
    q = dr / S
    print(f"{m_type:11s} split={SPLIT}  resid {res:.2e}  orth {orth:.2e}  |R_ii|/sigma_i in [{q[:n - n // 8].min():.3f}, {q[:n - n // 8].max():.3f}]  perm ok {sorted(J.cpu().tolist()) == list(range(1, n + 1))}")
