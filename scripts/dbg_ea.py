import numpy as np, torch
from randlapack_amd import device as d
import oracle
ctx = d.Context(0)
m = n = 512
A0 = d.drv_mat_gen(ctx, "adverserial", m, n, key=(3, 0), scaling=1e-10)["A"]
A0n = d.cm_to_numpy(A0)
A = A0.clone()
out = d.drv_bqrrp(ctx, A, m, n, 64, 1.0, qrcp_wide=0, qr_tall=1, want_sketch=True)
t = out["tau"].cpu().numpy(); J = out["J"].cpu().numpy()
Af = d.cm_to_numpy(A)
Q = oracle.ungqr(Af, t); R = np.triu(Af)
E = A0n[:, J - 1] - Q @ R
print("per block col resid", [f"{np.linalg.norm(E[:, j:j+64]):.1e}" for j in range(0, n, 64)])
print("diag R", np.abs(np.diag(R))[::32])
o = oracle.bqrrp(A0n, 64, 1.0, qrcp_wide=0, qr_tall=1, sketch=d.cm_to_numpy(out["sketch"]))
print("oracle diag R", np.abs(np.diag(np.triu(o["A"])))[::32])
print("first J differ at", int(np.argmax(J != o["J"])))
# also the residual when Q is built only from each panel: E_k = (Q_k^T applied) -- row-block view
QtA = Q.T @ A0n[:, J - 1]
print("||tril(Q^T A P)|| per block row", [f"{np.linalg.norm(np.tril(QtA, -1)[i:i+64]):.1e}" for i in range(0, n, 64)])
print("rank", out["rank"], o["rank"])
print("row-block errors of last col block", [f"{np.linalg.norm(E[i:i+64, 448:]):.1e}" for i in range(0, n, 64)])
Eq = QtA - R
print("R vs Q^T A P, last block rows 448:, by col", [f"{np.linalg.norm(Eq[448:, j]):.1e}" for j in range(448, 512, 8)])
print("diag R last block", np.abs(np.diag(R))[448::8])
print("diag QtA last blk", np.abs(np.diag(QtA))[448::8])
QR = Q @ R
mism = []
for c in range(440, 512):
    dist = np.linalg.norm(A0n - QR[:, [c]], axis=0)
    best = int(np.argmin(dist))
    if best != J[c] - 1:
        mism.append((c, int(J[c] - 1), best, float(dist[J[c]-1]), float(dist[best])))
print("mismatches (pos, J, best, dist_J, dist_best):", mism[:12])
print("J is permutation", sorted(J.tolist()) == list(range(1, n + 1)))
print("col norms AP last blk", np.linalg.norm(A0n[:, J[448:] - 1], axis=0)[::8])
print("E col norms last blk", np.linalg.norm(E[:, 448:], axis=0)[::8])
print("max |R12| rows<448 cols>=448", np.abs(R[:448, 448:]).max(), " Eq rows<448:", np.abs(Eq[:448, 448:]).max(), "argmax row", np.unravel_index(np.abs(Eq[:, 448:]).argmax(), Eq[:, 448:].shape))
print("Eq by row block for last cols", [f"{np.linalg.norm(Eq[i:i+64, 448:]):.1e}" for i in range(0, n, 64)])
cn = np.linalg.norm(E, axis=0)
bad = np.nonzero(cn > 1e-10)[0]
print("bad columns", bad, cn[bad])
for c in bad[:3]:
    print("col", c, "Eq col nonzero rows", np.nonzero(np.abs(Eq[:, c]) > 1e-10)[0], "vals", Eq[np.abs(Eq[:, c]) > 1e-10, c][:5], "R[c,c]", R[c, c], "tau", t[c-2:c+2])
print("tau last", t[500:])
