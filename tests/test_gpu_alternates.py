"""-m gpu: the library's kernel selection is decided from the shapes alone (no environment switches); what CAN be switched is the short list
of per-context options in include/rlhip.h (enum rlhip_option).  The pairs of routes behind the first four options are compared bitwise or to
rounding by their own tests (test_cholqrq_one_stream_equals_three_calls, test_gesdd_gram_route,
test_gesdd_persistent_jacobi_equals_per_launch_sweeps, test_trsm_fused_asm_and_plain_loads_agree_bitwise); this file covers the rest of the
list and the kernels that used to be reachable only through a switch and are now reached by SHAPE: each block below names the fallback
kernel its shape lands on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _d():
    from randlapack_amd import device as d

    return d


def test_option_api_round_trip(ctx):
    for name in ctx.OPT:
        assert ctx.get_option(name) == -1, f"{name} is not at its default at the start of a test"
    with ctx.options(gesdd_gram=0, bqrrp_lookahead_min_elems=0):
        assert ctx.get_option("gesdd_gram") == 0 and ctx.get_option("bqrrp_lookahead_min_elems") == 0
    assert ctx.get_option("gesdd_gram") == -1
    assert ctx.lib.rlhip_set_option(ctx.h, 99, 1) == -1 and ctx.lib.rlhip_get_option(ctx.h, -1) == -(2**63)


@pytest.mark.parametrize("tall", [1, 0])
def test_hqrrp_tall_panel_option_both_orders_factor(ctx, tall):
    """hqrrp with pivoted panels: pivots of a tall panel from the QRCP of its R factor (default) or one pivoted sweep (the reference's order,
    rl_hqrrp.hh:557-805) -- both a valid GEQP3-format factorization with the same pivot quality."""
    import scipy.linalg.lapack as ll

    d = _d()
    rng = np.random.default_rng(8)
    mm, nn = 4096, 512
    H0 = rng.standard_normal((mm, nn)) * np.logspace(0, -4, nn)[rng.permutation(nn)]
    Hd = d.cm_from_numpy(H0)
    with ctx.options(hqrrp_tall_panel=tall):
        r = d.drv_hqrrp(ctx, Hd, mm, nn, nb_alg=64, pp=10, panel_pivoting=1, qr_type=0, key=(2, 0))
    Jh = r["J"].cpu().numpy(); Ho = d.cm_to_numpy(Hd); th = r["tau"].cpu().numpy()
    assert sorted(Jh.tolist()) == list(range(1, nn + 1))
    qfull, _, info = ll.dorgqr(np.asfortranarray(Ho[:, :nn].copy()), th)
    assert np.linalg.norm(H0[:, Jh - 1] - qfull @ np.triu(Ho)[:nn]) <= EPS**0.75 * np.linalg.norm(H0)
    dg = np.abs(np.diag(Ho)[:nn])
    sv = np.linalg.svd(H0, compute_uv=False)
    assert np.all(dg / sv > 0.05) and np.all(dg / sv < 20)


def test_bqrrp_cholqr_fallback_option_off_is_the_reference_behaviour(ctx):
    """BQRRP with Cholesky-QR panels on a matrix whose second panel is numerically rank deficient after preconditioning: with the fallback (default)
    that panel is refactored by Householder reflectors and Q stays orthonormal; with the option off the reference's statement order runs on
    (rl_bqrrp.hh:461).  Pivots and rank are the same either way."""
    d = _d()
    rng = np.random.default_rng(12)
    m, n, b = 2048, 512, 128
    s = np.concatenate([np.logspace(0, -2, 200), np.full(n - 200, 1e-15)])
    A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    out = {}
    for fb in (1, 0):
        Ad = d.cm_from_numpy(A)
        with ctx.options(bqrrp_cholqr_fallback=fb):
            r = d.drv_bqrrp(ctx, Ad, m, n, b, qrcp_wide=0, qr_tall=1, apply_trans_q=1, key=(4, 0), want_sketch=True)
        out[fb] = (r["J"].cpu().numpy(), r["rank"], d.cm_to_numpy(Ad), r["tau"].cpu().numpy())
    assert np.array_equal(out[1][0][:128], out[0][0][:128])                        # the first block never takes the branch
    import scipy.linalg.lapack as ll
    J, rank, F, tau = out[1]
    q, _, info = ll.dorgqr(np.asfortranarray(F[:, :n].copy()), tau)
    assert np.linalg.norm(q.T @ q - np.eye(n)) <= 1e-10                            # default: orthonormal whatever the panels looked like
    assert np.linalg.norm(A[:, J - 1] - q @ np.triu(F)[:n]) <= 1e-12 * np.linalg.norm(A)


def test_shapes_that_land_on_the_fallback_kernels(ctx):
    """Every kernel that used to sit behind an environment switch is the natural route of some shape; LAPACK / numpy decide."""
    import scipy.linalg.lapack as ll
    import torch

    d = _d()
    rng = np.random.default_rng(5)
    # pivoted QR whose columns do not fit LDS: the rendezvous kernel (qrcp_kernel), pivots identical to dgeqp3
    m, n = 9600, 96
    A = rng.standard_normal((m, n)) * np.logspace(0, -3, n)[rng.permutation(n)]
    Ad = d.cm_from_numpy(A); J = torch.zeros(n, dtype=torch.int64, device="cuda"); tau = torch.zeros(n, dtype=torch.float64, device="cuda")
    assert ctx.lib.rlhip_geqp3_f64(ctx.h, m, n, Ad.data_ptr(), m, J.data_ptr(), tau.data_ptr()) == 0
    qr_ref, jp_ref, _, _, _ = ll.dgeqp3(A)
    assert np.array_equal(J.cpu().numpy(), jp_ref)
    assert np.abs(np.abs(np.triu(d.cm_to_numpy(Ad))[:n]) - np.abs(np.triu(qr_ref)[:n])).max() <= 1e-11 * np.abs(qr_ref).max()
    # row-pivoted LU: < 1024 rows below the diagonal (LDS panel kernel), 1024 .. 65536 (fast register steps), beyond (general register step)
    for (mm, nn, dt, fn, ref) in ((900, 200, np.float64, ctx.lib.rlhip_getrf_f64, ll.dgetrf), (3000, 200, np.float64, ctx.lib.rlhip_getrf_f64, ll.dgetrf),
                                  (40000, 96, np.float64, ctx.lib.rlhip_getrf_f64, ll.dgetrf), (70000, 64, np.float32, ctx.lib.rlhip_getrf_f32, ll.sgetrf)):
        B = (rng.standard_normal((mm, nn)) * np.logspace(0, -2, nn)).astype(dt)
        Bd = d.cm_from_numpy(B); ip = torch.zeros(nn, dtype=torch.int64, device="cuda")
        assert fn(ctx.h, mm, nn, Bd.data_ptr(), mm, ip.data_ptr()) == 0
        lu_ref, piv_ref, _ = ref(B)
        assert np.array_equal(ip.cpu().numpy() - 1, piv_ref), ("getrf pivots", mm, nn, dt)
        assert np.abs(d.cm_to_numpy(Bd) - lu_ref).max() <= (1e-12 if dt is np.float64 else 2e-4) * np.abs(lu_ref).max()
    # right-upper solve below the fused kernel's row threshold: per-block MFMA kernel (trsm_blk_kernel), ragged last block
    for (mm, nn) in ((3000, 300), (15000, 512)):
        U = np.triu(rng.standard_normal((nn, nn))) + 25 * np.eye(nn)
        Bm = rng.standard_normal((mm, nn))
        Bd = d.cm_from_numpy(Bm)
        before = ctx.path_count(2)
        ctx.trsm(mm, nn, 1.0, d.cm_from_numpy(U), nn, Bd, mm)
        assert ctx.path_count(2) == before, "the fused kernel took a short input"
        X = d.cm_to_numpy(Bd)
        assert np.linalg.norm(X @ U - Bm) <= 1e-13 * np.linalg.norm(Bm) * np.sqrt(nn), ("trsm", mm, nn)
    # products the persistent kernel declines (N not a multiple of 256; K short): the tiled MFMA kernel, all four layouts
    for dt, tol in ((torch.float64, 1e-13), (torch.float32, 3e-5)):
        for (ta, mm, nn, kk) in (("N", 4096, 200, 2048), ("T", 2048, 256, 1000)):
            Am = torch.randn((kk, mm) if ta == "N" else (mm, kk), dtype=dt, device="cuda")
            Bm = torch.randn((nn, kk), dtype=dt, device="cuda")
            Cm = torch.zeros((nn, mm), dtype=dt, device="cuda")
            before = ctx.path_count(0) + ctx.path_count(1)
            ctx.gemm(ta, "N", mm, nn, kk, 1.0, Am, mm if ta == "N" else kk, Bm, kk, 0.0, Cm, mm)
            assert ctx.path_count(0) + ctx.path_count(1) == before
            An = Am.T.double().cpu().numpy() if ta == "N" else Am.double().cpu().numpy()
            ref = An @ Bm.T.double().cpu().numpy()
            assert np.abs(Cm.T.double().cpu().numpy() - ref).max() <= tol * np.sqrt(kk) * np.abs(ref).max(), ("gemm", dt, ta)
    # unpivoted Householder QR: wider than the block-pipelined kernel's 2048 rows (flag-pipelined kernel), tall-skinny ill-conditioned
    # (sketch-preconditioned Cholesky-QR retry)
    for (mm, nn, cond) in ((2600, 300, 10.0), (20000, 64, 1e10)):
        Q0 = (np.linalg.qr(rng.standard_normal((mm, nn)))[0] * np.logspace(0, -np.log10(cond), nn)) @ np.linalg.qr(rng.standard_normal((nn, nn)))[0].T
        Qd = d.cm_from_numpy(Q0); tq = torch.zeros(nn, dtype=torch.float64, device="cuda")
        assert ctx.lib.rlhip_geqrf_f64(ctx.h, mm, nn, Qd.data_ptr(), mm, tq.data_ptr()) == 0
        qr_ref, _, _, _ = ll.dgeqrf(Q0)
        got = d.cm_to_numpy(Qd)
        assert np.abs(np.abs(np.triu(got)[:nn]) - np.abs(np.triu(qr_ref)[:nn])).max() <= 1e-9 * np.abs(qr_ref).max(), ("geqrf R", mm, nn, cond)
    # sparse products with more than 32 right-hand sides: a wavefront per row (csr_spmm_rm_kernel) + the transpose passes
    import scipy.sparse as sp
    Ssp = sp.random(3000, 1700, 0.01, random_state=np.random.default_rng(3), format="csr", data_rvs=np.random.default_rng(4).standard_normal).tocsr()
    op = d.CsrOperator.from_scipy(Ssp)
    for nb_ in (7, 100):
        X = rng.standard_normal((1700, nb_)); Z = rng.standard_normal((3000, nb_))
        Y = d.cm_to_numpy(d.linop_apply(ctx, op, "L", "N", d.cm_from_numpy(X), 3000, nb_, 1700))
        assert np.abs(Y - Ssp @ X).max() <= 1e-13 * np.abs(Ssp @ X).max() * 10, ("spmm", nb_)
        Yt = d.cm_to_numpy(d.linop_apply(ctx, op, "L", "T", d.cm_from_numpy(Z), 1700, nb_, 3000))
        assert np.abs(Yt - Ssp.T @ Z).max() <= 1e-13 * np.abs(Ssp.T @ Z).max() * 10, ("spmm^T", nb_)
