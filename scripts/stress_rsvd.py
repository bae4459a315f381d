import numpy as np, torch, sys, itertools
sys.path.insert(0, "tests")
from _gen import poly_mat
from randlapack_amd import device as d
import oracle
ctx = d.Context(0)
rng = np.random.default_rng(1)
bad = 0
for (m, n, rank, cond) in [(600, 200, 200, 1e10), (300, 500, 120, 1e6), (1000, 64, 64, 1e14), (257, 255, 100, 1e3)]:
    A = poly_mat(m, n, rank, rng, cond=cond)
    for (k, b, p, q, s1, s2, s3, oc) in [(40, 40, 0, 1, 0, 0, 0, False), (40, 16, 2, 1, 0, 0, 0, True), (64, 64, 3, 2, 1, 1, 1, False), (50, 10, 4, 2, 2, 0, 0, True),
                                          (min(m, n), 32, 1, 1, 0, 1, 0, False), (30, 30, 5, 1, 2, 1, 0, False)]:
        k = min(k, m, n)
        r = d.drv_rsvd(ctx, d.cm_from_numpy(A), m, n, k, b, 1e-9, p, q, rs_stab=s1, rf_orth=s2, qb_orth=s3, orth_check=oc, key=(4, 0))
        o = oracle.rsvd(A, k, b, 1e-9, p, q, rs_stab=s1, rf_orth=s2, qb_orth=s3, orth_check=oc, key=(4, 0))
        U, S, V = d.cm_to_numpy(r["U"]), r["S"].cpu().numpy(), d.cm_to_numpy(r["V"])
        e_dev = np.linalg.norm(A - (U * S) @ V.T) / np.linalg.norm(A)
        e_orc = np.linalg.norm(A - (o["U"] * o["S"]) @ o["V"].T) / np.linalg.norm(A)
        kk = min(r["k"], o["k"], 10)
        ds = np.max(np.abs(S[:kk] - o["S"][:kk])) / o["S"][0] if kk else 0
        flag = "" if (r["k"] == o["k"] and r["qb_rc"] == o["qb_rc"] and abs(e_dev - e_orc) <= 1e-6 * max(e_orc, 1e-10) + 1e-12 and ds < 1e-10) else "  <-- LOOK"
        bad += bool(flag)
        print(f"{m}x{n} r{rank} c{cond:.0e} k{k} b{b} p{p} q{q} stab{s1}{s2}{s3} oc{int(oc)}: k {r['k']}/{o['k']} rc {r['qb_rc']}/{o['qb_rc']} err {e_dev:.3e}/{e_orc:.3e} dS {ds:.1e} orthU {np.linalg.norm(U.T @ U - np.eye(U.shape[1])):.1e}{flag}")
print("flagged", bad)
