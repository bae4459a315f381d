"""Worker for tests/test_gpu_sharded.py: WORLD_SIZE ranks share cuda:0 and run the row-sharded C++ drivers; the
all-reduce hook exchanges through gloo on the host, so every reduction point of the sharded drivers is exercised
on a 1-GPU box.  Rank 0 compares with the unsharded device run and the oracle and prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    from randlapack_amd import device as d
    from randlapack_amd import sharded

    m, n, k, p = (int(x) for x in sys.argv[1:5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(2024)
    A = rng.standard_normal((m, n)) @ np.diag(np.linspace(1.0, 0.05, n)) @ np.linalg.qr(rng.standard_normal((n, n)))[0]
    rows = np.array_split(np.arange(m), world)[rank]
    ctx = d.Context(0)
    transport = sharded.init_comm(ctx, dist)
    assert transport == "torch.distributed"
    assert ctx.lib.rlhip_comm_size(ctx.h) == world and ctx.lib.rlhip_comm_rank(ctx.h) == rank
    Aloc = d.cm_from_numpy(np.ascontiguousarray(A[rows]))
    r = d.drv_rsvd(ctx, Aloc, len(rows), n, k, k, 1e-12, p, 1)
    Uloc = d.cm_to_numpy(r["U"])
    # multi-block QB on the shards (b_sz < k exercises the deflation / re-orthogonalisation all-reduces)
    r2 = d.drv_rsvd(ctx, Aloc, len(rows), n, k, max(k // 4, 1), 1e-12, p, 1)
    U2loc = d.cm_to_numpy(r2["U"])
    # CQRRPT on the shards: sketch partial sums and the Gram matrix are the two exchanges
    ncq = min(n, 96)
    Acq = A[:, :ncq] * np.logspace(0, -3, ncq)
    Aq = d.cm_from_numpy(np.ascontiguousarray(Acq[rows]))
    rq = d.drv_cqrrpt(ctx, Aq, len(rows), ncq, 1.25, 4, key=(5, 0))
    Qloc, Rq, Jq = d.cm_to_numpy(Aq), d.cm_to_numpy(rq["R"]), rq["J"].cpu().numpy()
    # BQRRP on the shards (config 4's layout): Gram / top-block / W / R12 exchanges
    nbq, bb = min(n, 120), 32
    Abq = A[:, :nbq] * np.logspace(0, -2, nbq)
    Ab = d.cm_from_numpy(np.ascontiguousarray(Abq[rows]))
    rb = d.drv_bqrrp(ctx, Ab, len(rows), nbq, bb, 1.0, key=(8, 0), m_global=m)          # the reference's default triple {luqr, geqrf, ormqr}: TSQR panels
    Ab_loc, tau_b, J_b = d.cm_to_numpy(Ab), rb["tau"].cpu().numpy(), rb["J"].cpu().numpy()
    # ... and the Cholesky-QR panels of the fast triple {luqr, cholqr, gemqrt}
    Af = d.cm_from_numpy(np.ascontiguousarray(Abq[rows]))
    rf = d.drv_bqrrp(ctx, Af, len(rows), nbq, bb, 1.0, key=(8, 0), m_global=m, qrcp_wide=0, qr_tall=1, apply_trans_q=1)
    Af_loc, tau_f, J_f = d.cm_to_numpy(Af), rf["tau"].cpu().numpy(), rf["J"].cpu().numpy()
    # ... and the same call with the look-ahead of call_sharded FORCED (side queue beside the tail of the apply; collectives through this
    #     process group's hook): one process per rank, as in production
    Al = d.cm_from_numpy(np.ascontiguousarray(Abq[rows]))
    la0 = ctx.path_count(12)
    with ctx.options(bqrrp_lookahead_min_elems=0):
        rl = d.drv_bqrrp(ctx, Al, len(rows), nbq, bb, 1.0, key=(8, 0), m_global=m, qrcp_wide=0, qr_tall=1, apply_trans_q=1)
    la_taken = ctx.path_count(12) - la0
    Al_loc, tau_l, J_l = d.cm_to_numpy(Al), rl["tau"].cpu().numpy(), rl["J"].cpu().numpy()
    # the same factorization with the rows dealt block-cyclically (blocks of bb rows, block g on rank g % world)
    crows = np.concatenate([np.arange(g * bb, min((g + 1) * bb, m)) for g in range(rank, (m + bb - 1) // bb, world)] or [np.zeros(0, dtype=np.int64)]).astype(np.int64)
    Ac = d.cm_from_numpy(np.ascontiguousarray(Abq[crows]))
    rc_ = d.drv_bqrrp(ctx, Ac, len(crows), nbq, bb, 1.0, key=(8, 0), m_global=m, block_cyclic=True)
    Ac_loc, tau_c, J_c = d.cm_to_numpy(Ac), rc_["tau"].cpu().numpy(), rc_["J"].cpu().numpy()
    # standalone hqrrp on the shards (TSQR panels, pivots of a pivoted panel from the QRCP of its R factor): three panel types
    nhq, nbh = min(n, 112), 32
    Ahq = A[:, :nhq] * np.logspace(0, -3, nhq)[np.random.default_rng(4).permutation(nhq)]
    hq = {}
    for tag, (pv, qt) in (("piv", (1, 0)), ("qr", (0, 0)), ("chol", (0, 2))):
        Ah = d.cm_from_numpy(np.ascontiguousarray(Ahq[rows]))
        rh = d.drv_hqrrp(ctx, Ah, len(rows), nhq, nb_alg=nbh, pp=8, panel_pivoting=pv, qr_type=qt, key=(13, 0), m_global=m)
        hq[tag] = (d.cm_to_numpy(Ah), rh["tau"].cpu().numpy(), rh["J"].cpu().numpy(), rh["rc"])
    # ABRIK on the row-sharded operator (CQRRT panels)
    ka, ita = 8, 8
    ra = d.drv_abrik(ctx, Aloc, len(rows), n, ka, 1e-12, ita, key=(6, 0), qr_exp=1)
    Ua_loc = d.cm_to_numpy(ra["U"])
    # ... and with the reference's DEFAULT panels (qr_exp = geqrf_ungqr): sharded Cholesky-QR panel + the Householder sign vector of its top block
    rha = d.drv_abrik(ctx, Aloc, len(rows), n, ka, 1e-12, ita, key=(6, 0), qr_exp=0)
    Uh_loc = d.cm_to_numpy(rha["U"])
    # linop QR drivers and ABRIK on a ROW-SHARDED sparse operator (each rank holds a row block of the CSR matrix)
    import scipy.sparse as sp
    nsp = min(n, 60)
    Ssp = (sp.random(m, nsp, 0.15, random_state=np.random.default_rng(11), format="csr", data_rvs=np.random.default_rng(12).standard_normal)
           @ sp.diags(0.9 ** np.arange(nsp))).tocsr()
    op_loc = d.CsrOperator.from_scipy(Ssp[rows], device="cuda:0")
    lin = {alg: d.cm_to_numpy(d.drv_qr_linops(ctx, alg, op_loc, d_factor=2.0, nnz=2, key=(9, 0))["R"]) for alg in ("cqrrt", "cholqr", "scholqr3")}
    rsa = d.drv_abrik_linop(ctx, op_loc, 6, 1e-12, max_krylov_iters=6, key=(6, 0), qr_exp=1)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rows, Uloc, U2loc, Qloc, Ua_loc, Ab_loc, crows, Ac_loc, Af_loc, {t: v[0] for t, v in hq.items()}, Al_loc, la_taken, Uh_loc))
    ctx.lib.rlhip_comm_destroy(ctx.h)
    if rank == 0:
        import oracle

        U = np.zeros((m, r["k"])); U2 = np.zeros((m, r2["k"]))
        Qc = np.zeros((m, ncq))
        Ua = np.zeros((m, ra["triplets"]))
        Abq_out = np.zeros((m, nbq))
        Acq_out = np.zeros((m, nbq))
        Afq_out = np.zeros((m, nbq))
        Hq_out = {t: np.zeros((m, nhq)) for t in hq}
        Alq_out = np.zeros((m, nbq))
        la_counts = []
        Uh = np.zeros((m, rha["triplets"]))
        for rr, u, u2, qq, ua, ab, cr, ac, af, hh, al, lat, uh in gathered:
            U[rr] = u; U2[rr] = u2; Qc[rr] = qq; Ua[rr] = ua; Abq_out[rr] = ab; Acq_out[cr] = ac; Afq_out[rr] = af; Alq_out[rr] = al; Uh[rr] = uh
            la_counts.append(int(lat))
            for t in hh:
                Hq_out[t][rr] = hh[t]
        S, V = r["S"].cpu().numpy(), d.cm_to_numpy(r["V"])
        S2, V2 = r2["S"].cpu().numpy(), d.cm_to_numpy(r2["V"])
        ctx1 = d.Context(0)
        r1 = d.drv_rsvd(ctx1, d.cm_from_numpy(A), m, n, k, k, 1e-12, p, 1)
        S1 = r1["S"].cpu().numpy()
        ref = oracle.rsvd(A, k, k, 1e-12, p, 1)          # same Philox stream on both sides (oracle/oracle.cpp fill_dense)
        Ab1 = d.cm_from_numpy(Abq)
        rb1 = d.drv_bqrrp(ctx1, Ab1, m, nbq, bb, 1.0, key=(8, 0))
        Ab1n = d.cm_to_numpy(Ab1)
        Qb = oracle.ungqr(Abq_out, tau_b)
        Rb = np.triu(Abq_out)[:nbq]
        ra1 = d.drv_abrik(ctx1, d.cm_from_numpy(A), m, n, ka, 1e-12, ita, key=(6, 0), qr_exp=1)
        Sa, Sa1, Va = ra["S"].cpu().numpy(), ra1["S"].cpu().numpy(), d.cm_to_numpy(ra["V"])
        rha1 = d.drv_abrik(ctx1, d.cm_from_numpy(A), m, n, ka, 1e-12, ita, key=(6, 0), qr_exp=0)
        Sh, Sh1, Vh = rha["S"].cpu().numpy(), rha1["S"].cpu().numpy(), d.cm_to_numpy(rha["V"])
        sv = np.linalg.svd(A, compute_uv=False)
        Aq1 = d.cm_from_numpy(Acq)
        rq1 = d.drv_cqrrpt(ctx1, Aq1, m, ncq, 1.25, 4, key=(5, 0))
        kq = rq["rank"]
        nA = np.linalg.norm(A)
        op1 = d.CsrOperator.from_scipy(Ssp, device="cuda:0")
        lin1 = {alg: d.cm_to_numpy(d.drv_qr_linops(ctx1, alg, op1, d_factor=2.0, nnz=2, key=(9, 0))["R"]) for alg in ("cqrrt", "cholqr", "scholqr3")}
        rsa1 = d.drv_abrik_linop(ctx1, op1, 6, 1e-12, max_krylov_iters=6, key=(6, 0), qr_exp=1)
        Ssa, Ssa1 = rsa["S"].cpu().numpy(), rsa1["S"].cpu().numpy()
        hq_cmp = {}
        for tag, (pv, qt) in (("piv", (1, 0)), ("qr", (0, 0)), ("chol", (0, 2))):
            Ah1 = d.cm_from_numpy(Ahq)
            rh1 = d.drv_hqrrp(ctx1, Ah1, m, nhq, nb_alg=nbh, pp=8, panel_pivoting=pv, qr_type=qt, key=(13, 0))
            F1, Fs = d.cm_to_numpy(Ah1), Hq_out[tag]
            Jh = hq[tag][2]
            Qh = oracle.ungqr(Fs, hq[tag][1])
            hq_cmp[tag] = dict(rc=[hq[tag][3], rh1["rc"]], J_equal=bool(np.array_equal(Jh, rh1["J"].cpu().numpy())),
                               A=float(np.linalg.norm(Fs - F1) / np.linalg.norm(F1)), tau=float(np.max(np.abs(hq[tag][1] - rh1["tau"].cpu().numpy()))),
                               resid=float(np.linalg.norm(Ahq[:, Jh - 1] - Qh @ np.triu(Fs)[:nhq]) / np.linalg.norm(Ahq)),
                               orth=float(np.linalg.norm(Qh.T @ Qh - np.eye(min(m, nhq)))))
        out = dict(
            hqrrp=hq_cmp,
            lin_R={alg: float(np.linalg.norm(np.triu(lin[alg]) - np.triu(lin1[alg])) / np.linalg.norm(np.triu(lin1[alg]))) for alg in lin},
            sp_abrik_trip=[rsa["triplets"], rsa1["triplets"]], sp_abrik_S=float(np.max(np.abs(Ssa[:6] - Ssa1[:6]) / Ssa1[:6])),
            bq_rank=rb["rank"], bq_rank1=rb1["rank"], bq_J_equal=bool(np.array_equal(J_b, rb1["J"].cpu().numpy())),
            bq_A=float(np.linalg.norm(Abq_out - Ab1n) / np.linalg.norm(Ab1n)), bq_tau=float(np.max(np.abs(tau_b - rb1["tau"].cpu().numpy()))),
            bqf_rank=rf["rank"], bqf_J_equal=bool(np.array_equal(J_f, rb1["J"].cpu().numpy())),
            bqf_A=float(np.linalg.norm(Afq_out - Ab1n) / np.linalg.norm(Ab1n)), bqf_tau=float(np.max(np.abs(tau_f - rb1["tau"].cpu().numpy()))),
            bql_rank=rl["rank"], bql_J_equal=bool(np.array_equal(J_l, rb1["J"].cpu().numpy())), bql_lookaheads=la_counts,
            bql_A=float(np.linalg.norm(Alq_out - Ab1n) / np.linalg.norm(Ab1n)), bql_tau=float(np.max(np.abs(tau_l - rb1["tau"].cpu().numpy()))),
            bqc_rank=rc_["rank"], bqc_J_equal=bool(np.array_equal(J_c, rb1["J"].cpu().numpy())),
            bqc_A=float(np.linalg.norm(Acq_out - Ab1n) / np.linalg.norm(Ab1n)), bqc_tau=float(np.max(np.abs(tau_c - rb1["tau"].cpu().numpy()))),
            bq_resid=float(np.linalg.norm(Abq[:, J_b - 1] - Qb @ Rb) / np.linalg.norm(Abq)), bq_orth=float(np.linalg.norm(Qb.T @ Qb - np.eye(nbq))),
            abh_iters=[rha["iters"], rha1["iters"]], abh_trip=[rha["triplets"], rha1["triplets"]], abh_next_ctr=[list(rha["next_ctr"]), list(rha1["next_ctr"])],
            abh_S_vs_single=float(np.max(np.abs(Sh[:ka] - Sh1[:ka]) / Sh1[:ka])), abh_orthU=float(np.linalg.norm(Uh.T @ Uh - np.eye(rha["triplets"]))),
            abh_res=float(min(np.linalg.norm(A.T @ Uh - Vh * Sh), np.linalg.norm(A @ Vh - Uh * Sh))),
            ab_iters=ra["iters"], ab_iters1=ra1["iters"], ab_trip=ra["triplets"], ab_trip1=ra1["triplets"],
            ab_S_vs_single=float(np.max(np.abs(Sa[:ka] - Sa1[:ka]) / Sa1[:ka])),
            ab_S_vs_exact=float(np.max(np.abs(Sa[:ka] - sv[:ka]) / sv[:ka])),
            ab_orthU=float(np.linalg.norm(Ua.T @ Ua - np.eye(ra["triplets"]))),
            ab_res=float(min(np.linalg.norm(A.T @ Ua - Va * Sa), np.linalg.norm(A @ Va - Ua * Sa))),
            cq_rank=kq, cq_rank1=rq1["rank"], cq_J_equal=bool(np.array_equal(Jq, rq1["J"].cpu().numpy())),
            cq_R=float(np.linalg.norm(Rq[:kq] - d.cm_to_numpy(rq1["R"])[:kq]) / np.linalg.norm(Rq[:kq])),
            cq_resid=float(np.linalg.norm(Acq[:, Jq - 1] - Qc[:, :kq] @ Rq[:kq]) / np.linalg.norm(Acq)),
            cq_orth=float(np.linalg.norm(Qc[:, :kq].T @ Qc[:, :kq] - np.eye(kq))),
            k=r["k"], k2=r2["k"], qb_rc=r["qb_rc"], qb_rc2=r2["qb_rc"],
            S_vs_single=float(np.max(np.abs(S - S1)) / S1[0]),
            S_vs_oracle=float(np.max(np.abs(S - ref["S"])) / ref["S"][0]),
            recon=float(np.linalg.norm(A - (U * S) @ V.T) / nA),
            recon_ref=float(np.linalg.norm(A - (ref["U"] * ref["S"]) @ ref["V"].T) / nA),
            orthU=float(np.linalg.norm(U.T @ U - np.eye(r["k"]))),
            orthV=float(np.linalg.norm(V.T @ V - np.eye(r["k"]))),
            recon2=float(np.linalg.norm(A - (U2 * S2) @ V2.T) / nA),
            orthU2=float(np.linalg.norm(U2.T @ U2 - np.eye(r2["k"]))),
        )
        if os.environ.get("RLHIP_TEST_DEBUG"):
            print("abrik default panels: sharded S", Sh[:5], "single S", Sh1[:5], "cqrrt sharded S", Sa[:5], file=sys.stderr)
            print("S", S[:5], "S1", S1[:5], "ref", ref["S"][:5], ref["k"], ref["qb_rc"], file=sys.stderr)
        print("SHARDED_RESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
