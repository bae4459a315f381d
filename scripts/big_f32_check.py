import torch, numpy as np, time
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, b, dt) in [(16384, 1024, torch.float32), (8192, 512, torch.float64)]:
    A0 = d.drv_mat_gen(ctx, "gaussian", m, m, key=(2, 0), dtype=dt)["A"]
    A = A0.clone()
    t0 = time.perf_counter(); o = d.drv_bqrrp(ctx, A, m, m, b, 1.0, qr_tall=1, apply_trans_q=1); torch.cuda.synchronize(); dtm = time.perf_counter() - t0
    Q = A.clone()
    suf = "f32" if dt == torch.float32 else "f64"
    getattr(ctx.lib, f"rlhip_ungqr_{suf}")(ctx.h, m, m, m, Q.data_ptr(), m, o["tau"].data_ptr())
    R = torch.triu(A.T)                       # (m, m) rows = R rows
    AP = A0[(o["J"] - 1).long()]              # (n, m)
    resid = torch.linalg.norm(AP - R.T @ Q) / torch.linalg.norm(A0)
    orth = torch.linalg.norm(Q @ Q.T - torch.eye(m, dtype=dt, device=Q.device)) / np.sqrt(m)
    dg = torch.abs(torch.diagonal(R))
    print(suf, m, b, f"{dtm*1e3:.0f} ms rank {o['rank']} resid {float(resid):.2e} orth/sqrt(n) {float(orth):.2e} diag monotone-ish {float((dg[1:] <= dg[:-1] * 1.5).float().mean()):.3f}")
