import numpy as np, torch
from randlapack_amd import device as d
from benchmarks import _common as c
import oracle
ctx = d.Context(0)
m = n = 512
A0 = c.regen(ctx, "kahan", m, n, theta=1.2, perturb=1e3)
A0n = d.cm_to_numpy(A0)
for qt in (1, 2):
    A = A0.clone()
    out = d.drv_bqrrp(ctx, A, m, n, 64, 1.0, qr_tall=qt, want_sketch=True)
    t = out["tau"].cpu().numpy(); J = out["J"].cpu().numpy()
    Qo = oracle.ungqr(d.cm_to_numpy(A), t)
    R = np.triu(d.cm_to_numpy(A))
    print("device qr_tall", qt, "rank", out["rank"], "orth", np.linalg.norm(Qo.T @ Qo - np.eye(n)), "resid", np.linalg.norm(A0n[:, J - 1] - Qo @ R) / np.linalg.norm(A0n), "tau max", t.max())
    o = oracle.bqrrp(A0n, 64, 1.0, qr_tall=qt, sketch=d.cm_to_numpy(out["sketch"]))
    print("oracle keys", {k: (v if np.isscalar(v) else getattr(v, 'shape', None)) for k, v in o.items()})
    Qr = oracle.ungqr(o["A"], o["tau"]); Rr = np.triu(o["A"])
    print("oracle qr_tall", qt, "rank", o["rank"], "orth", np.linalg.norm(Qr.T @ Qr - np.eye(n)), "resid", np.linalg.norm(A0n[:, o["J"] - 1] - Qr @ Rr) / np.linalg.norm(A0n), "tau max", o["tau"].max(), "J equal", np.array_equal(J, o["J"]))
