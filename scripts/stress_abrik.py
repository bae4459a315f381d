import numpy as np, torch, sys
sys.path.insert(0, "tests")
from _gen import poly_mat
from randlapack_amd import device as d
import oracle
ctx = d.Context(0)
rng = np.random.default_rng(0)
for (m, n, k, it, qr) in [(200, 600, 8, 6, 0), (600, 200, 8, 6, 1), (500, 500, 16, 5, 0), (1000, 300, 7, 9, 1), (300, 1000, 5, 8, 1), (64, 64, 4, 40, 0), (2000, 50, 10, 12, 0)]:
    A = poly_mat(m, n, min(m, n), rng, cond=1e6)
    try:
        o = d.drv_abrik(ctx, d.cm_from_numpy(A), m, n, k, 1e-10, max_krylov_iters=it, key=(2, 0), qr_exp=qr)
    except Exception as e:
        print(m, n, k, it, qr, "device raised", str(e)[:100]); continue
    r = oracle.abrik(A, k, 1e-10, it, key=(2, 0))
    t = o["triplets"]
    U, V, S = d.cm_to_numpy(o["U"]), d.cm_to_numpy(o["V"]), o["S"].cpu().numpy()
    sv = np.linalg.svd(A, compute_uv=False)
    lead = min(4, t)
    print(m, n, k, it, qr, "trip", t, r["triplets"], "iters", o["iters"], r["iters"], "S err vs exact", float(np.max(np.abs(S[:lead] - sv[:lead]) / sv[:lead])),
          "vs oracle", float(np.max(np.abs(S[:lead] - r["S"][:lead]) / sv[0])) if r["triplets"] >= lead else None,
          "orthU", float(np.linalg.norm(U.T @ U - np.eye(t))), "orthV", float(np.linalg.norm(V.T @ V - np.eye(t))))
