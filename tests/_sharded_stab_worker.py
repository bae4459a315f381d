"""Worker for tests/test_gpu_sharded.py::test_rowsharded_stabilisers_*: the row-sharded PLUL (tournament pivoting) and HQRQ (TSQR)
stabilisers of rl_orth.hh, the RSVD object graph the reference's own tests build (PLUL inside the power scheme, p = 2), and CQRRPT with
the replicated hqrrp / bqrrp QRCP and with the orthonormal completion -- WORLD_SIZE ranks share cuda:0, the all-reduce hook exchanges
through gloo.  Rank 0 gathers, compares with the single-device run and prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist

    from randlapack_amd import device as d
    from randlapack_amd import sharded

    m, n, k, p = (int(x) for x in sys.argv[1:5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(77)
    # rank-k plus noise at 1e-9: the power scheme's result is then determined to ~1e-9 * (noise / gap)^(2p+1) whatever basis the stabiliser returns
    sig = np.linspace(10.0, 1.0, k)
    A = (np.linalg.qr(rng.standard_normal((m, k)))[0] * sig) @ np.linalg.qr(rng.standard_normal((n, k)))[0].T + 1e-9 * rng.standard_normal((m, n))
    rows = np.array_split(np.arange(m), world)[rank]
    ctx = d.Context(0)
    sharded.init_comm(ctx, dist)
    Aloc = d.cm_from_numpy(np.ascontiguousarray(A[rows]))
    res = {}
    # ---- the stabilisers themselves on a sharded tall block
    Y = rng.standard_normal((m, k)) @ (np.eye(k) + 0.3 * rng.standard_normal((k, k)))
    for name, kind in (("hqrq", 1), ("plul", 2)):
        Yl = d.cm_from_numpy(np.ascontiguousarray(Y[rows]))
        rc, _ = d.drv_stab(ctx, kind, Yl, len(rows), k)
        res[name] = (rc, d.cm_to_numpy(Yl))
    # ---- RSVD with each stabiliser inside the power scheme (rs_stab), CholQRQ as the orthogonalizer
    rs = {}
    for name, kind in (("cholqrq", 0), ("hqrq", 1), ("plul", 2)):
        r = d.drv_rsvd(ctx, Aloc, len(rows), n, k, k, 1e-12, p, 1, rs_stab=kind)
        rs[name] = (r["S"].cpu().numpy(), d.cm_to_numpy(r["U"]), d.cm_to_numpy(r["V"]), r["qb_rc"], r["k"])
    # ---- CQRRPT: replicated hqrrp / bqrrp QRCP of the sketch, and the orthonormal completion of a rank-deficient input
    ncq = min(n, 64)
    Acq = A[:, :ncq] * np.logspace(0, -3, ncq)
    cq = {}
    for name, qr in (("hqrrp", 0), ("bqrrp", 1)):
        Aq = d.cm_from_numpy(np.ascontiguousarray(Acq[rows]))
        rq = d.drv_cqrrpt(ctx, Aq, len(rows), ncq, 1.25, 4, key=(5, 0), qrcp=qr)
        cq[name] = (d.cm_to_numpy(Aq), d.cm_to_numpy(rq["R"]), rq["J"].cpu().numpy(), rq["rank"])
    Adef = Acq.copy()
    Adef[:, ncq // 2:] = Adef[:, : ncq - ncq // 2] @ rng.standard_normal((ncq - ncq // 2, ncq - ncq // 2)) * 0 + Adef[:, : ncq - ncq // 2]   # duplicate columns: rank ncq / 2
    Ao = d.cm_from_numpy(np.ascontiguousarray(Adef[rows]))
    ro = d.drv_cqrrpt(ctx, Ao, len(rows), ncq, 1.25, 4, key=(5, 0), qrcp=16 + 2)       # orthogonalization = true, geqp3
    cq["orth"] = (d.cm_to_numpy(Ao), d.cm_to_numpy(ro["R"]), ro["J"].cpu().numpy(), ro["rank"])
    gathered = [None] * world
    dist.all_gather_object(gathered, (rows, {a: v[1] for a, v in res.items()}, {a: v[1] for a, v in rs.items()}, {a: v[0] for a, v in cq.items()}))
    ctx.lib.rlhip_comm_destroy(ctx.h)
    if rank == 0:
        def assemble(pick, cols):
            M = np.zeros((m, cols))
            for rr, st, ru, qq in gathered:
                M[rr] = pick(st, ru, qq)
            return M

        ctx1 = d.Context(0)
        out = {}
        Qh = assemble(lambda st, ru, qq: st["hqrq"], k)
        Lp = assemble(lambda st, ru, qq: st["plul"], k)
        Py = Y @ np.linalg.pinv(Y)                                   # projector on span(Y)
        out["hqrq_rc"], out["plul_rc"] = res["hqrq"][0], res["plul"][0]
        out["hqrq_orth"] = float(np.linalg.norm(Qh.T @ Qh - np.eye(k)))
        out["hqrq_span"] = float(np.linalg.norm(Qh - Py @ Qh) + np.linalg.norm(Y - Qh @ (Qh.T @ Y)) / np.linalg.norm(Y))
        out["plul_span"] = float(np.linalg.norm(Lp - Py @ Lp) / np.linalg.norm(Lp) + np.linalg.norm(Y - Lp @ np.linalg.lstsq(Lp, Y, rcond=None)[0]) / np.linalg.norm(Y))
        out["plul_max"] = float(np.abs(Lp).max())
        out["plul_cond"] = float(np.linalg.cond(Lp))
        # k rows of L form a unit lower triangular matrix in the tournament's pivot order: exactly k rows hold a 1 that is their last non-zero
        last = np.array([np.max(np.nonzero(np.abs(row) > 1e-13)[0]) if np.any(np.abs(row) > 1e-13) else -1 for row in Lp])
        out["plul_unit_rows"] = int(sum(1 for c in range(k) if np.any((last == c) & (np.abs(Lp[:, c] - 1.0) < 1e-12))))
        sv = np.linalg.svd(A, compute_uv=False)[:k]
        for name, kind in (("cholqrq", 0), ("hqrq", 1), ("plul", 2)):
            S = rs[name][0]
            U = assemble(lambda st, ru, qq, nm=name: ru[nm], rs[name][4])
            V = rs[name][2]
            r1 = d.drv_rsvd(ctx1, d.cm_from_numpy(A), m, n, k, k, 1e-12, p, 1, rs_stab=kind)
            S1 = r1["S"].cpu().numpy()
            out[f"rsvd_{name}"] = dict(k=rs[name][4], S_vs_single=float(np.max(np.abs(S - S1)) / S1[0]), S_vs_exact=float(np.max(np.abs(S - sv)) / sv[0]),
                                       orthU=float(np.linalg.norm(U.T @ U - np.eye(len(S)))), recon=float(np.linalg.norm(A - (U * S) @ V.T) / np.linalg.norm(A)))
        for name, qr in (("hqrrp", 0), ("bqrrp", 1)):
            Qc = assemble(lambda st, ru, qq, nm=name: qq[nm], ncq)
            _, Rq, Jq, kq = cq[name]
            A1 = d.cm_from_numpy(Acq)
            r1 = d.drv_cqrrpt(ctx1, A1, m, ncq, 1.25, 4, key=(5, 0), qrcp=qr)
            out[f"cq_{name}"] = dict(rank=[kq, r1["rank"]], J_equal=bool(np.array_equal(Jq, r1["J"].cpu().numpy())),
                                     R=float(np.linalg.norm(Rq[:kq] - d.cm_to_numpy(r1["R"])[:kq]) / np.linalg.norm(Rq[:kq])),
                                     resid=float(np.linalg.norm(Acq[:, Jq - 1] - Qc[:, :kq] @ Rq[:kq]) / np.linalg.norm(Acq)),
                                     orth=float(np.linalg.norm(Qc[:, :kq].T @ Qc[:, :kq] - np.eye(kq))))
        Qo = assemble(lambda st, ru, qq: qq["orth"], ncq)
        _, Ro, Jo, ko = cq["orth"]
        out["cq_orth"] = dict(rank=ko, ncols=ncq, orth_all=float(np.linalg.norm(Qo.T @ Qo - np.eye(ncq))),
                              resid=float(np.linalg.norm(Adef - Qo[:, :ko] @ (Qo[:, :ko].T @ Adef)) / np.linalg.norm(Adef)))   # the first `rank` columns span A
        print("STAB_RESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
