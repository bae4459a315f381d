"""Validity sweep: every pivoted-QR driver on the reference's hard test matrices (error-analysis set + adversarial)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np, torch, itertools
from randlapack_amd import device as d
import oracle, sys
ctx = d.Context(0)
DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.float64
EPS = np.finfo(float).eps
tests = [("polynomial", dict(cond_num=1e10, exponent=2.0)), ("step", dict(cond_num=1e10)), ("spiked", dict(scaling=1e10)),
         ("kahan", dict(theta=1.2, perturb=1e3)), ("adverserial", dict(scaling=1e-10)), ("exponential", dict(cond_num=1e14))]

def qr_quality(A0n, Afact, tau, J, k=None):
    n = A0n.shape[1]
    k = k or min(A0n.shape)
    Q = oracle.ungqr(Afact[:, :k], tau[:k])
    R = np.triu(Afact)[:k]
    return np.linalg.norm(A0n[:, J - 1] - Q @ R) / np.linalg.norm(A0n), np.linalg.norm(Q.T @ Q - np.eye(k)), float(np.abs(tau).max())

for (mt, kw), (m, n) in itertools.product(tests, [(512, 512), (1500, 300)]):
    if mt == "kahan" and m != n:
        continue
    A0 = d.drv_mat_gen(ctx, mt, m, n, key=(3, 0), dtype=DT, **kw)["A"]
    A0n = d.cm_to_numpy(A0).astype(np.float64)
    row = [f"{mt:12s} {m}x{n}"]
    for qw, qt, ap in [(0, 1, 1), (1, 1, 1), (0, 2, 0), (0, 0, 1)]:
        A = A0.clone()
        o = d.drv_bqrrp(ctx, A, m, n, 64, 1.0, qrcp_wide=qw, qr_tall=qt, apply_trans_q=ap)
        r, orth, tm = qr_quality(A0n, d.cm_to_numpy(A).astype(np.float64), o["tau"].cpu().numpy().astype(np.float64), o["J"].cpu().numpy())
        row.append(f"bq{qw}{qt}{ap}: {r:.1e}/{orth:.1e}")
    for qtype in (0, 1, 2):
        A = A0.clone()
        o = d.drv_hqrrp(ctx, A, m, n, nb_alg=64, qr_type=qtype)
        r, orth, tm = qr_quality(A0n, d.cm_to_numpy(A).astype(np.float64), o["tau"].cpu().numpy().astype(np.float64), o["J"].cpu().numpy())
        row.append(f"hq{qtype}: {r:.1e}/{orth:.1e}")
    if m > n:
        for qrcp in (0, 1, 2):
            A = A0.clone()
            o = d.drv_cqrrpt(ctx, A, m, n, 1.25, 4, qrcp=qrcp)
            k = o["rank"]; J = o["J"].cpu().numpy()
            if o["rc"] == 0 and k > 0:
                Q = d.cm_to_numpy(A)[:, :k].astype(np.float64); R = np.triu(d.cm_to_numpy(o["R"]).astype(np.float64))[:k]
                row.append(f"cq{qrcp}: k={k} {np.linalg.norm(A0n[:, J - 1] - Q @ R) / np.linalg.norm(A0n):.1e}/{np.linalg.norm(Q.T @ Q - np.eye(k)):.1e}")
            else:
                row.append(f"cq{qrcp}: rc={o['rc']} k={k}")
    print("  ".join(row), flush=True)
