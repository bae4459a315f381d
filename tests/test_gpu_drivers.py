"""-m gpu: the driver/comp objects (C++ layer over the C ABI) against the CPU oracle -- the parity tests
proper.  Same seeds on both sides; where stated the device-generated sketch is injected into the oracle so
the comparison starts from bit-identical Omega (the reference's own GPU-vs-CPU precedent,
test/drivers/test_bqrrp_gpu.cu:91-110,231-249)."""
import numpy as np
import pytest

from _gen import poly_mat, poly_singvals, with_singvals

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _d():
    from randlapack_amd import device

    return device


def subspace_gap(Q1, Q2):
    return np.linalg.norm(Q1 - Q2 @ (Q2.T @ Q1))


def test_cholqrq_vs_oracle(ctx, orc):
    d = _d()
    rng = np.random.default_rng(0)
    Y = poly_mat(1000, 300, 300, rng, cond=100.0) @ rng.standard_normal((300, 200))
    Yd = d.cm_from_numpy(Y)
    rc, fail = d.drv_stab(ctx, 0, Yd, 1000, 200)
    rc_o, Qo = orc.stab(0, Y)
    assert rc == rc_o == 0 and not fail
    Q = d.cm_to_numpy(Yd)
    # CholQR's Q is unique (R has positive diagonal): entrywise agreement up to cond^2*eps
    np.testing.assert_allclose(Q, Qo, atol=1e-9)
    # second pass restores orthonormality to eps^0.625 (test_orth.cc:135-153)
    d.drv_stab(ctx, 0, Yd, 1000, 200)
    Q2 = d.cm_to_numpy(Yd)
    assert np.linalg.norm(Q2.T @ Q2 - np.eye(200)) <= EPS**0.625


@pytest.mark.parametrize("m,k,dtype", [(40000, 256, "f64"), (20000, 256, "f32"), (17000, 256, "f64")])
def test_cholqrq_one_stream_equals_three_calls(ctx, orc, monkeypatch, m, k, dtype):
    """CholQRQ on a tall 256-column input runs as ONE stream of kernels with one host read (rlhip_cholqrq: syrk, device-side potrf info,
    conditioning guard and gated fused solve; path counter 11) -- the same kernels as the three separate calls, so Q is BITWISE the same;
    checked against the oracle too.  A rank-deficient input reports the reference's failure (rl_orth.hh:81-85) and leaves A untouched."""
    import torch

    d = _d()
    rng = np.random.default_rng(m + k)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    Y = rng.standard_normal((m, k)) @ (np.eye(k) + 0.01 * rng.standard_normal((k, k)))
    out = {}
    for fused in ("1", "0"):
        with ctx.options(cholqrq_one_stream=int(fused)):
            Yd = d.cm_from_numpy(Y).to(tdt)
            before = ctx.path_count(11)
            rc, fail = d.drv_stab(ctx, 0, Yd, m, k)
        assert rc == 0 and not fail
        assert ctx.path_count(11) - before == (1 if fused == "1" else 0)
        out[fused] = d.cm_to_numpy(Yd)
    assert np.array_equal(out["1"], out["0"])
    Q = out["1"].astype(np.float64)
    eps = np.finfo(np.float64 if dtype == "f64" else np.float32).eps
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= 50 * eps * k
    if dtype == "f64":
        rc_o, Qo = orc.stab(0, Y)
        assert rc_o == 0
        np.testing.assert_allclose(Q, Qo, atol=1e-11)
    # failure: a zero column -> a zero pivot; the reference returns 1 before its trsm, A keeps its values
    Yb = Y.copy()
    Yb[:, 17] = 0.0
    Ybd = d.cm_from_numpy(Yb).to(tdt)
    keep = Ybd.clone()
    rc, fail = d.drv_stab(ctx, 0, Ybd, m, k)
    assert rc == 1 and fail
    assert torch.equal(Ybd, keep)


def test_cholqrq_failure_code(ctx, orc):
    d = _d()
    A = np.ones((50, 4))
    rc, fail = d.drv_stab(ctx, 0, d.cm_from_numpy(A), 50, 4)
    assert rc == 1 and fail                                  # rl_orth.hh:81-85, same as the oracle
    assert orc.stab(0, A)[0] == 1


@pytest.mark.parametrize("p,q", [(0, 1), (1, 1), (2, 1), (3, 2), (4, 2)])
def test_rs_vs_oracle(ctx, orc, p, q):
    d = _d()
    rng = np.random.default_rng(p * 10 + q)
    m, n, k = 400, 150, 20
    A = poly_mat(m, n, n, rng, cond=1e3)
    rc, Om, nxt = d.drv_rs(ctx, d.cm_from_numpy(A), m, n, k, p, q, key=(5, 0))
    rc_o, Om_o, nxt_o = orc.rs(A, k, p, q, key=(5, 0))
    assert rc == rc_o == 0 and nxt == nxt_o                   # RNG state advanced identically (A.1, A.2)
    np.testing.assert_allclose(d.cm_to_numpy(Om), Om_o, rtol=0, atol=1e-9 * np.abs(Om_o).max())


def test_rf_vs_oracle_with_injected_sketch(ctx, orc):
    d = _d()
    rng = np.random.default_rng(7)
    m, n, k = 600, 200, 32
    A = poly_mat(m, n, n, rng, cond=1e4)
    Ad = d.cm_from_numpy(A)
    # device sketch, extracted first, then replayed into the oracle: parity "from Omega onward"
    Om = d.cm_empty(n, k)
    ctx.fill_dense(Om, n, k, key=(3, 0))
    rc, Q, _ = d.drv_rf(ctx, Ad, m, n, k, 0, 1, key=(3, 0))
    with orc.inject_sketch(d.cm_to_numpy(Om).T.ravel()):
        rc_o, Qo, _ = orc.rf(A, k, 0, 1, key=(3, 0))
    assert rc == rc_o == 0
    Qg = d.cm_to_numpy(Q)
    assert np.linalg.norm(Qg.T @ Qg - np.eye(k)) <= EPS**0.625
    np.testing.assert_allclose(Qg, Qo, atol=1e-8)            # same unique CholQR factor (cond^2 * eps)
    assert subspace_gap(Qg, Qo) <= 1e-10


@pytest.mark.parametrize("b_sz,p", [(2, 5), (10, 2), (10, 5), (50, 2), (7, 2)])
def test_qb_vs_oracle(ctx, orc, b_sz, p):
    # test/comps/test_qb.cc:236-363 sizes: 100x100, k=50
    d = _d()
    rng = np.random.default_rng(11)
    m = n = 100
    k = 50
    A = with_singvals(m, n, poly_singvals(k, 0.1, 2025.0, 2.0), rng)
    tol = EPS**0.75
    rc, kf, Q, BT, nxt = d.drv_qb(ctx, d.cm_from_numpy(A), m, n, k, b_sz, tol, p, 1)
    rc_o, kf_o, Qo, BTo, nxt_o = orc.qb(A, k, b_sz, tol, p, 1)
    assert (kf, nxt) == (kf_o, nxt_o)                         # final k and RNG state: exact
    # the matrix has exact rank k, so after the last block |norm_A - norm_B| is pure rounding noise and the
    # "tolerance reached (0)" vs "rank reached (3)" verdict may legitimately differ in that one corner
    assert rc == rc_o or {rc, rc_o} <= {0, 3}
    Qg, Bg = d.cm_to_numpy(Q), d.cm_to_numpy(BT)
    assert np.linalg.norm(Qg.T @ Qg - np.eye(kf)) <= EPS**0.625                    # test_qb.cc:162-174
    assert np.linalg.norm(A - Qg @ Bg.T) <= EPS**0.625 * np.linalg.norm(A)
    assert abs(np.linalg.norm(A - Qg @ Bg.T) - np.linalg.norm(A - Qo @ BTo.T)) <= 1e-10 * np.linalg.norm(A)


def test_qb_early_exit_codes_match(ctx, orc):
    # rank-20 matrix, ask for 50 with loose tol -> both sides must stop at the same block with code 0
    d = _d()
    rng = np.random.default_rng(12)
    A = with_singvals(200, 120, poly_singvals(20, 0.1, 50.0, 2.0), rng)
    # tol well above the sqrt(eps)-level noise floor of the error estimate (rl_qb.hh:225), so the verdict is
    # not decided by the last bit
    rc, kf, Q, BT, _ = d.drv_qb(ctx, d.cm_from_numpy(A), 200, 120, 50, 10, 1e-6, 2, 1)
    rc_o, kf_o, _, _, _ = orc.qb(A, 50, 10, 1e-6, 2, 1)
    assert (rc, kf) == (rc_o, kf_o) and rc in (0, 2)
    assert np.linalg.norm(A - d.cm_to_numpy(Q) @ d.cm_to_numpy(BT).T) <= 1e-6 * np.linalg.norm(A)


@pytest.mark.parametrize("m,n,k,b,p,q", [(4096, 512, 64, 64, 2, 1), (1000, 300, 40, 10, 2, 1), (500, 200, 50, 50, 0, 1),
                                         (2000, 400, 32, 16, 1, 1), (300, 300, 1, 1, 2, 1)])
def test_rsvd_vs_oracle(ctx, orc, m, n, k, b, p, q):
    """First case = BASELINE.json configs[0] (4096x512 fp64, rank 64)."""
    d = _d()
    rng = np.random.default_rng(m + k)
    A = poly_mat(m, n, n, rng)
    tol = EPS**0.5625
    r = d.drv_rsvd(ctx, d.cm_from_numpy(A), m, n, k, b, tol, p, q)
    o = orc.rsvd(A, k, b, tol, p, q)
    assert (r["rc"], r["qb_rc"], r["k"], r["next_ctr"]) == (o["rc"], o["qb_rc"], o["k"], o["next_ctr"])
    U, S, V = d.cm_to_numpy(r["U"]), r["S"].cpu().numpy(), d.cm_to_numpy(r["V"])
    kk = r["k"]
    # singular values: relative 1e-12*sqrt(k)-class agreement on the leading part of the spectrum; absolute
    # 1e-10*sigma_1 everywhere (with p = 0 the CholQR range finder itself loses cond(A*Omega)^2 * eps on BOTH
    # sides, so the trailing values carry rounding-dependent noise of that size)
    big = o["S"] > 0.1 * o["S"][0]
    assert np.max(np.abs(S[big] - o["S"][big]) / o["S"][big]) <= 1e-12 * np.sqrt(kk) * 100
    assert np.max(np.abs(S - o["S"])) <= 1e-10 * o["S"][0]
    errg = np.linalg.norm(A - (U * S) @ V.T)
    erro = np.linalg.norm(A - (o["U"] * o["S"]) @ o["V"].T)
    assert errg <= erro * (1 + 1e-8) + 1e-12 * np.linalg.norm(A)       # reconstruction as good as the reference's
    assert np.linalg.norm(V.T @ V - np.eye(kk)) <= EPS**0.75 * np.sqrt(n) * 10
    assert subspace_gap(V[:, :max(1, kk // 2)], o["V"]) <= 1e-6         # leading right subspace agrees


def test_rsvd_bad_arguments_raise(ctx):
    d = _d()
    from randlapack_amd._lib import RlhipError

    A = d.cm_zeros(4, 4)
    with pytest.raises(RlhipError):
        d.drv_rsvd(ctx, A, 4, 4, 0, 2, 1e-3, 0, 1)          # k <= 0   (rl_rsvd.hh:130)
    with pytest.raises(RlhipError):
        d.drv_rsvd(ctx, A, 4, 4, 2, 2, -1.0, 0, 1)          # tol < 0  (rl_rsvd.hh:131)


def test_rsvd_full_size_properties(ctx):
    """BASELINE.json configs[1]: 200000 x 20000 fp64, rank 256, one QB block, p = 0.  The oracle cannot run at
    this size in seconds, so parity is carried by size-independent properties: orthonormal factors, the
    norm identity ||A Omega|| consistency via sigma, and B = Q^T A (||U S V^T||_F^2 = sum sigma^2 <= ||A||_F^2)."""
    import torch

    d = _d()
    m, n, k = 200000, 20000, 256
    A = d.cm_empty(m, n)
    ctx.fill_dense(A, m, n, key=(7, 0))
    r = d.drv_rsvd(ctx, A, m, n, k, k, 1e-12, 0, 1)
    assert r["rc"] == 0 and r["qb_rc"] == 3 and r["k"] == k and r["next_ctr"] == (n * k // 4, 0, 0, 0)
    U, S, V = r["U"], r["S"], r["V"]
    I = torch.eye(k, device="cuda", dtype=torch.float64)
    assert float(torch.linalg.norm(U @ U.T - I)) <= EPS**0.75 * np.sqrt(n)       # U is stored (k, m)
    assert float(torch.linalg.norm(V @ V.T - I)) <= EPS**0.75 * np.sqrt(n)
    assert bool((S[:-1] >= S[1:]).all()) and float(S[-1]) > 0
    # Gaussian A: the captured singular values lie inside the Marchenko-Pastur bulk edge
    assert float(S[0]) <= np.sqrt(m) + np.sqrt(n) + 5 and float(S[-1]) >= np.sqrt(m) - np.sqrt(n) - 5
    # A^T U = V S  on a column sample of A (checks the triple against the data itself)
    cols = torch.arange(0, n, 97, device="cuda")
    lhs = A[cols] @ U.T                                  # (len(cols), m) @ (m, k)
    rhs = V[:, cols].T * S
    assert float(torch.linalg.norm(lhs - rhs)) <= 1e-9 * float(torch.linalg.norm(rhs))


# ---------------------------------------------------------------------------------------------------
# CQRRPT (drivers/rl_cqrrpt.hh) vs the oracle, sharing ONE sketch (test_bqrrp_gpu.cu:91-110 precedent)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,rank,cond", [(10000, 200, 200, 1.0), (4000, 200, 100, 1e6), (2000, 50, 50, 1e10),
                                           (10, 5, 5, 1.0), (5000, 300, 300, 1e8)])
def test_cqrrpt_vs_oracle_shared_sketch(ctx, orc, m, n, rank, cond):
    d = _d()
    rng = np.random.default_rng(m + n + rank)
    A = rng.standard_normal((m, n)) if cond == 1.0 else poly_mat(m, n, rank, rng, cond=cond)
    eps_user = EPS**0.85
    Ad = d.cm_from_numpy(A)
    r = d.drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, eps=eps_user, want_sketch=True, key=(11, 0))
    o = orc.cqrrpt(A, d.cm_to_numpy(r["sketch"]), eps_user)
    assert r["rc"] == o["rc"] == 0
    assert r["rank"] == o["rank"]                                            # rank: exact (same thresholds, a19)
    k = r["rank"]
    J, Jo = r["J"].cpu().numpy(), o["J"]
    np.testing.assert_array_equal(J[:k], Jo[:k])                             # pivot order: bit-exact
    assert sorted(J.tolist()) == list(range(1, n + 1))
    Q, R = d.cm_to_numpy(Ad)[:, :k], d.cm_to_numpy(r["R"])[:k]
    if np.array_equal(J, Jo):
        assert np.linalg.norm(R - o["R"][:k]) <= EPS**0.6 * np.linalg.norm(o["R"])       # ||dR||_F <= eps^0.6
    atol = EPS**0.75
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= atol * np.linalg.norm(A)                # test_cqrrpt.cc:102-104
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= atol * np.sqrt(n)
    assert abs(k - rank) <= 5                                                              # :178-179


@pytest.mark.parametrize("n,cond", [(512, 1.0), (256, 1e8), (300, 1.0)])
def test_cqrrpt_folded_pivoting_equals_reference_statement_order(ctx, orc, n, cond, monkeypatch):
    """m >= 16384 with a full-rank sketch takes the folded flow (pivoting read inside the first solve, out-of-place solves through
    rlhip_trsm_gather; n = 300: its gather-copy route): same rank, pivots, R and Q as the reference's col_swap -> trsm -> syrk -> trsm
    order (CQRRPT::fold_pivoting = false), and both against the oracle on the shared sketch."""
    d = _d()
    m = 20000
    rng = np.random.default_rng(n)
    A = rng.standard_normal((m, n)) if cond == 1.0 else poly_mat(m, n, n, rng, cond=cond)
    eps_user = EPS**0.85
    res = {}
    for fold in ("1", "0"):
        Ad = d.cm_from_numpy(A)
        before = ctx.path_count(4)
        with ctx.options(cqrrpt_fold_pivoting=int(fold)):
            r = d.drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, eps=eps_user, want_sketch=True, key=(3, 0))
        took_fused_gather = ctx.path_count(4) - before
        assert r["rc"] == 0
        res[fold] = (r["rank"], r["J"].cpu().numpy(), d.cm_to_numpy(r["R"]), d.cm_to_numpy(Ad), d.cm_to_numpy(r["sketch"]), took_fused_gather)
    (k1, J1, R1, Q1, S1, g1), (k0, J0, R0, Q0, S0, g0) = res["1"], res["0"]
    # two fused out-of-place solves for a well-conditioned R_sk with whole 256-blocks; a graded R_sk (cond 1e8) fails the conditioning
    # guard of the first solve and a ragged n every shape test: those take the gather-copy route
    assert g0 == 0 and (g1 == 2 if (n % 256 == 0 and cond == 1.0) else g1 in (0, 1))
    assert k1 == k0 == n and np.array_equal(J1, J0) and np.array_equal(S1, S0)
    assert np.linalg.norm(R1 - R0) <= 1e-12 * np.linalg.norm(R0)
    assert np.linalg.norm(Q1 - Q0) <= 1e-11 * np.sqrt(n)
    o = orc.cqrrpt(A, S1, eps_user)
    assert o["rank"] == n and np.array_equal(o["J"], J1)
    assert np.linalg.norm(R1 - o["R"]) <= EPS**0.6 * np.linalg.norm(o["R"])
    assert np.linalg.norm(A[:, J1 - 1] - Q1 @ R1) <= EPS**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q1.T @ Q1 - np.eye(n)) <= EPS**0.75 * np.sqrt(n)


def test_cqrrpt_own_sketch_properties_and_state(ctx):
    d = _d()
    rng = np.random.default_rng(5)
    m, n = 20000, 256
    A = poly_mat(m, n, n, rng, cond=1e8)
    Ad = d.cm_from_numpy(A)
    r = d.drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, key=(3, 0))
    dsk = int(1.25 * n)
    assert dsk == int(1.25 * n)
    assert r["rc"] == 0 and r["next_ctr"] == (m * ((4 + 1) // 2), 0, 0, 0)      # S.next_state of the independent-column SASO: m * ceil(nnz / 2) Philox blocks
    k = r["rank"]
    J = r["J"].cpu().numpy()
    Q, R = d.cm_to_numpy(Ad)[:, :k], d.cm_to_numpy(r["R"])[:k]
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= EPS**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= EPS**0.75 * np.sqrt(n)
    # rank-revealing quality: |R_ii| tracks the singular values within a modest factor
    s = np.linalg.svd(A, compute_uv=False)[:k]
    ratio = np.abs(np.diag(R)) / s
    assert ratio.max() < 50 and ratio.min() > 1 / 50


def test_cqrrpt_zero_matrix_and_bad_args(ctx):
    from randlapack_amd._lib import RlhipError

    d = _d()
    A = d.cm_zeros(50, 5)
    r = d.drv_cqrrpt(ctx, A, 50, 5)
    assert r["rc"] == 0 and r["rank"] == 0 and float(A.abs().max()) == 0.0        # rl_cqrrpt.hh:256-261
    with pytest.raises(RlhipError):
        d.drv_cqrrpt(ctx, A, 50, 5, d_factor=0.5)                                  # d_factor < 1 (:165)


def test_cqrrpt_full_size_properties(ctx):
    """BASELINE.json configs[2]: 1048576 x 1024 fp64, d = 1280, nnz = 4."""
    import torch

    d = _d()
    m, n = 1048576, 1024
    A = d.cm_empty(m, n)
    ctx.fill_dense(A, m, n, key=(3, 0))
    colsum_before = A.sum(dim=1)                       # per-column sums survive a pure column permutation
    A0_sample = A[:, :4096].clone()
    r = d.drv_cqrrpt(ctx, A, m, n, 1.25, 4)
    assert r["rc"] == 0 and r["rank"] == n
    J = r["J"].cpu().numpy()
    assert sorted(J.tolist()) == list(range(1, n + 1))
    Q, R = A, r["R"]                                   # Q stored (n, m); R stored (n, n) column-major -> R[j, i] = R_ij
    I = torch.eye(n, device="cuda", dtype=torch.float64)
    assert float(torch.linalg.norm(Q @ Q.T - I)) <= EPS**0.75 * np.sqrt(n)
    Rm = R.T                                           # (n, n) with Rm[i, j] = R_ij
    assert float(torch.linalg.norm(torch.tril(Rm, -1))) == 0.0
    # A P = Q R on a 4096-row sample
    AP = A0_sample[torch.from_numpy(J - 1).cuda()]     # (n, 4096): rows = permuted columns of A
    QR = (Q[:, :4096].T @ Rm).T
    assert float(torch.linalg.norm(AP - QR)) <= EPS**0.75 * float(torch.linalg.norm(AP))
    del colsum_before


# ---------------------------------------------------------------------------------------------------
# HQRQ / PLUL stabilisers (comps/rl_orth.hh:100-230)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,k", [(1000, 200), (4096, 256), (50, 7)])
def test_hqrq_plul_vs_oracle(ctx, orc, m, k):
    d = _d()
    rng = np.random.default_rng(m + k)
    Y = rng.standard_normal((m, k))
    Yd = d.cm_from_numpy(Y)
    rc, _ = d.drv_stab(ctx, 1, Yd, m, k)
    Q = d.cm_to_numpy(Yd)
    rco, Qo = orc.stab(1, Y)
    assert rc == rco == 0
    np.testing.assert_allclose(Q, Qo, atol=1e-12, rtol=0)                             # same reflectors -> same Q
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= EPS**0.625 * np.sqrt(k)             # test_orth.cc tolerance
    Yd = d.cm_from_numpy(Y)
    rc, _ = d.drv_stab(ctx, 2, Yd, m, k)
    L = d.cm_to_numpy(Yd)
    rco, Lo = orc.stab(2, Y)
    assert rc == rco == 0
    np.testing.assert_allclose(L, Lo, atol=1e-11, rtol=0)
    assert np.abs(L).max() <= 1.0 + 1e-14                                             # partial pivoting: |l_ij| <= 1


def test_plul_singular_input_device(ctx, orc):
    # test_orth.cc:109-133: rank-deficient input returns 0, entries finite and <= 1
    d = _d()
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((200, 10)) @ rng.standard_normal((10, 40))
    Y[:, 5] = 0
    Yd = d.cm_from_numpy(Y)
    rc, _ = d.drv_stab(ctx, 2, Yd, 200, 40)
    L = d.cm_to_numpy(Yd)
    assert rc == 0 and np.all(np.isfinite(L)) and np.abs(L).max() <= 1.0 + 1e-12


def test_rf_with_hqrq_device(ctx, orc):
    # test/comps/test_rf.cc:143-201: 100 x 100, k in {100, 50}, p = 5, HQRQ for both the stabiliser and the orthogonaliser
    d = _d()
    rng = np.random.default_rng(12)
    for k in (100, 50):
        A = poly_mat(100, 100, k, rng, cond=1e3) if k < 100 else rng.standard_normal((100, 100))
        rc, Q, _ = d.drv_rf(ctx, d.cm_from_numpy(A), 100, 100, k, 5, 1, rs_stab=1, orth_kind=1, key=(9, 0))
        Qn = d.cm_to_numpy(Q)
        assert rc == 0
        assert np.linalg.norm(Qn.T @ Qn - np.eye(k)) <= EPS**0.625 * np.sqrt(k)
        if k < 100:
            assert np.linalg.norm(A - Qn @ (Qn.T @ A)) <= 1e-9 * np.linalg.norm(A)


# ---------------------------------------------------------------------------------------------------
# BQRRP (drivers/rl_bqrrp.hh) vs the oracle with a shared sketch; every subroutine family
# ---------------------------------------------------------------------------------------------------
def _bqrrp_verify(orc, A, Aout, tau, J, atol=EPS**0.75):
    m, n = A.shape
    mn = min(m, n)
    Q = orc.ungqr(Aout, tau)
    R = np.triu(Aout)[:mn]
    assert sorted(J.tolist()) == list(range(1, n + 1))
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= atol * np.linalg.norm(A)             # test_bqrrp.cc:105-107
    assert np.linalg.norm(Q.T @ Q - np.eye(mn)) <= atol * np.sqrt(mn)


BQRRP_OPTS = [(0, 1, 1), (1, 1, 1), (0, 2, 0), (0, 0, 1), (1, 0, 0), (0, 1, 0)]


@pytest.mark.parametrize("opts", BQRRP_OPTS)
@pytest.mark.parametrize("m,n,b,kind", [(1000, 400, 100, "gauss"), (1000, 400, 140, "gauss"), (500, 200, 50, "poly"),
                                        (1024, 1024, 128, "step")])
def test_bqrrp_vs_oracle_shared_sketch(ctx, orc, opts, m, n, b, kind):
    d = _d()
    rng = np.random.default_rng(m + n + b)
    if kind == "gauss":
        A = rng.standard_normal((m, n))
    elif kind == "poly":
        A = poly_mat(m, n, min(m, n), rng, cond=1e4)
    else:
        s = np.ones(n); s[n // 2:] = 1e-10
        A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    Ad = d.cm_from_numpy(A)
    r = d.drv_bqrrp(ctx, Ad, m, n, b, 1.0, want_sketch=True, key=(21, 0), qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2])
    o = orc.bqrrp(A, b, 1.0, qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2], sketch=d.cm_to_numpy(r["sketch"]))
    assert r["rc"] == o["rc"] == 0
    assert r["rank"] == o["rank"]
    Aout, tau, J = d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy()
    _bqrrp_verify(orc, A, Aout, tau, J)
    if kind != "step":
        # well-separated pivots: the permutation is bit-identical to the reference path's, R and tau agree to rounding
        np.testing.assert_array_equal(J, o["J"])
        mn = min(m, n)
        Rd, Ro = np.triu(Aout)[:mn], np.triu(o["A"])[:mn]
        assert np.linalg.norm(Rd - Ro) <= EPS**0.6 * np.linalg.norm(Ro)
        np.testing.assert_allclose(tau, o["tau"], atol=1e-9, rtol=0)
    else:
        # cond 1e10 step: pivots among the 1e-10 half are decided by rounding noise; the large half must agree as a set
        assert set(J[:n // 2].tolist()) == set(o["J"][:n // 2].tolist())
        dR = np.abs(np.diag(Aout))
        assert dR[n // 2 - 1] > 1e6 * dR[n // 2]


@pytest.mark.parametrize("opts", [(0, 1, 1), (1, 1, 1), (0, 2, 0)])
def test_bqrrp_lowrank_wide_and_zero(ctx, orc, opts):
    d = _d()
    rng = np.random.default_rng(31)
    kw = dict(qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2])
    # low rank (test_bqrrp.cc:188-207): rank is the block-rounded upper bound, identical on both sides
    A = poly_mat(400, 150, 60, rng, cond=1e3)
    Ad = d.cm_from_numpy(A)
    r = d.drv_bqrrp(ctx, Ad, 400, 150, 40, 1.0, want_sketch=True, key=(1, 0), **kw)
    o = orc.bqrrp(A, 40, 1.0, sketch=d.cm_to_numpy(r["sketch"]), **kw)
    assert r["rank"] == o["rank"] == 80
    J = r["J"].cpu().numpy()
    np.testing.assert_array_equal(J[:40], o["J"][:40])          # first block: pivots well separated -> exact
    _bqrrp_verify(orc, A, d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), J)
    # wide (test_bqrrp.cc:414-438)
    A = rng.standard_normal((300, 500))
    Ad = d.cm_from_numpy(A)
    r = d.drv_bqrrp(ctx, Ad, 300, 500, 64, 1.0, want_sketch=True, key=(2, 0), **kw)
    o = orc.bqrrp(A, 64, 1.0, sketch=d.cm_to_numpy(r["sketch"]), **kw)
    assert r["rank"] == o["rank"] == 300
    J = r["J"].cpu().numpy()
    np.testing.assert_array_equal(J[:300], o["J"][:300])
    _bqrrp_verify(orc, A, d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), J)
    # all-zero and below-eps inputs (test_bqrrp.cc:265-352; rl_bqrrp.hh:373-399)
    for scale in (0.0, 1e-20):
        A = scale * rng.standard_normal((100, 40))
        Ad = d.cm_from_numpy(A)
        r = d.drv_bqrrp(ctx, Ad, 100, 40, 10, 1.0, key=(3, 0), **kw)
        assert r["rc"] == 0 and r["rank"] == 0
        assert float(Ad.abs().max()) <= EPS**0.75


def test_bqrrp_state_and_internal_nb(ctx, orc):
    d = _d()
    rng = np.random.default_rng(41)
    A = rng.standard_normal((500, 280))
    Ad = d.cm_from_numpy(A)
    r = d.drv_bqrrp(ctx, Ad, 500, 280, 90, 1.0, internal_nb=30, key=(5, 0))
    assert r["rc"] == 0 and r["rank"] == 280
    assert r["next_ctr"][0] == (90 * 500 + 3) // 4               # one d x m fill_dense (rl_bqrrp.hh:310-311)
    _bqrrp_verify(orc, A, d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy())


def test_bqrrp_midsize_properties(ctx, orc):
    # 4096 x 4096, b = 256: residual through the implicit Q (apply Q^T to A[:, J] with the device gemqrt-equivalent)
    import torch

    d = _d()
    m = n = 4096
    b = 256
    A = d.cm_empty(m, n)
    ctx.fill_dense(A, m, n, key=(4, 0))
    A0 = A.clone()
    r = d.drv_bqrrp(ctx, A, m, n, b, 1.0, key=(6, 0))
    assert r["rc"] == 0 and r["rank"] == n
    J = r["J"]
    assert torch.equal(torch.sort(J).values, torch.arange(1, n + 1, device="cuda"))
    # Q from (V, tau) on the device: ungqr, then || A0[:, J] - Q R ||
    R = torch.triu(A.T).contiguous()                      # (n, m) tensor holds column-major A: .T is the m x n view
    Qd = A.clone()
    assert ctx.lib.rlhip_ungqr_f64(ctx.h, m, n, n, Qd.data_ptr(), m, r["tau"].data_ptr()) == 0
    ctx.sync()
    Q = Qd.T                                              # m x n view
    AP = A0.T[:, (J - 1)]
    res = torch.linalg.norm(AP - Q @ R) / torch.linalg.norm(AP)
    orth = torch.linalg.norm(Q.T @ Q - torch.eye(n, device="cuda", dtype=torch.float64))
    assert float(res) <= EPS**0.75 and float(orth) <= EPS**0.75 * np.sqrt(n)
    dR = torch.abs(torch.diagonal(R))
    assert float((dR[1:] <= dR[:-1] * 1.5).double().mean()) > 0.95    # |r_ii| essentially non-increasing (QRCP quality)


# ---------------------------------------------------------------------------------------------------
# RCCL binding at world size 1 (the N > 1 exchange pattern itself is covered by the gloo test on CPU)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("force_hook", [False, True])
def test_comm_world1_allreduce_is_identity(ctx, force_hook):
    import os

    import torch
    import torch.distributed as dist

    from randlapack_amd import sharded

    d = _d()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    os.environ["RLHIP_COMM_SINGLE_RANK_NCCL"] = "1"       # build a REAL one-rank RCCL communicator: ncclCommInitRank + ncclAllReduce run
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        transport = sharded.init_comm(ctx, dist, force_hook=force_hook)
        assert transport == ("torch.distributed" if force_hook else "rccl")
        assert ctx.lib.rlhip_comm_size(ctx.h) == 1 and ctx.lib.rlhip_comm_rank(ctx.h) == 0
        x = torch.arange(1000, dtype=torch.float64, device="cuda")
        assert ctx.lib.rlhip_allreduce_sum_f64(ctx.h, x.data_ptr(), 1000) == 0
        y = torch.arange(777, dtype=torch.float32, device="cuda")
        assert ctx.lib.rlhip_allreduce_sum_f32(ctx.h, y.data_ptr(), 777) == 0
        import ctypes as C

        h = (C.c_double * 3)(1.5, -2.0, 4.0)
        assert ctx.lib.rlhip_allreduce_sum_host_f64(ctx.h, h, 3) == 0 and list(h) == [1.5, -2.0, 4.0]
        ctx.sync()
        assert torch.equal(x, torch.arange(1000, dtype=torch.float64, device="cuda"))
        assert torch.equal(y, torch.arange(777, dtype=torch.float32, device="cuda"))
        # a driver call with a communicator attached (size 1: the reductions are skipped)
        A = d.cm_empty(2048, 256)
        ctx.fill_dense(A, 2048, 256, key=(1, 0))
        r = d.drv_rsvd(ctx, A, 2048, 256, 32, 32, 1e-12, 0, 1, key=(0, 0))
        assert r["rc"] == 0
    finally:
        ctx.lib.rlhip_comm_destroy(ctx.h)
        os.environ.pop("RLHIP_COMM_SINGLE_RANK_NCCL", None)
        if created:
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------
# HQRRP (drivers/rl_hqrrp.hh) vs the oracle sharing the Uniform(-1,1) sketching matrix G
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("qr_type,panel_pivoting", [(0, 1), (0, 0), (1, 0), (2, 0)])
@pytest.mark.parametrize("m,n,nb,pp", [(300, 120, 32, 5), (200, 200, 64, 10), (150, 260, 32, 8), (500, 70, 16, 4),
                                       (1280, 1024, 64, 10)])
def test_hqrrp_vs_oracle_shared_sketch(ctx, orc, m, n, nb, pp, qr_type, panel_pivoting):
    d = _d()
    rng = np.random.default_rng(m + n + nb)
    A = poly_mat(m, n, min(m, n), rng, cond=1e4)
    Ad = d.cm_from_numpy(A)
    r = d.drv_hqrrp(ctx, Ad, m, n, nb, pp, panel_pivoting, qr_type, key=(7, 0), want_G=True)
    o = orc.hqrrp(A, nb, pp, panel_pivoting, qr_type, key=(7, 0), G=d.cm_to_numpy(r["G"]))
    assert r["rc"] == o["rc"] == 0
    assert r["next_ctr"] == o["next_ctr"]                                    # one (nb+pp) x m fill (rl_hqrrp.hh:929-930)
    Aout, tau, J = d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy()
    np.testing.assert_array_equal(J, o["J"])                                 # pivot order: bit-exact
    mn = min(m, n)
    Rd, Ro = np.triu(Aout)[:mn], np.triu(o["A"])[:mn]
    assert np.linalg.norm(Rd - Ro) <= (1e-9 if qr_type == 2 else EPS**0.6) * np.linalg.norm(Ro)
    _bqrrp_verify(orc, A, Aout, tau, J, atol=(1e-8 if qr_type == 2 else EPS**0.75))


def test_hqrrp_own_sketch_and_bad_args(ctx, orc):
    d = _d()
    rng = np.random.default_rng(5)
    A = rng.standard_normal((400, 150))
    Ad = d.cm_from_numpy(A)
    r = d.drv_hqrrp(ctx, Ad, 400, 150, 32, 6, key=(9, 0))
    o = orc.hqrrp(A, 32, 6, key=(9, 0))                                      # both sides generate G from the same stream
    np.testing.assert_array_equal(r["J"].cpu().numpy(), o["J"])
    _bqrrp_verify(orc, A, d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy())
    from randlapack_amd import _lib

    with pytest.raises(_lib.RlhipError):
        d.drv_hqrrp(ctx, Ad, 400, 150, 0, 6)                                 # nb_alg must be positive


@pytest.mark.parametrize("qrcp", [0, 1, 2])
def test_cqrrpt_qrcp_choices_vs_oracle(ctx, orc, qrcp):
    """CQRRPT with qrcp = hqrrp / bqrrp / geqp3 (rl_cqrrpt.hh:230-247).  The inner randomized QRCP continues from the state the
    SASO construction leaves behind, so the oracle is started from that state."""
    import ctypes as C

    d = _d()
    rng = np.random.default_rng(77)
    m, n = 6000, 200
    A = poly_mat(m, n, n, rng, cond=1e5)
    dd = int(1.25 * n)
    nxt = (C.c_uint32 * 4)()
    S = C.c_void_p()
    assert ctx.lib.rlhip_saso_create(ctx.h, dd, m, 4, (C.c_uint32 * 4)(0, 0, 0, 0), (C.c_uint32 * 2)(11, 0), nxt, C.byref(S)) == 0
    ctx.lib.rlhip_saso_destroy(ctx.h, S)
    Ad = d.cm_from_numpy(A)
    r = d.drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, want_sketch=True, key=(11, 0), qrcp=qrcp)
    o = orc.cqrrpt(A, d.cm_to_numpy(r["sketch"]), EPS**0.85, qrcp=qrcp, ctr=tuple(nxt), key=(11, 0))
    assert r["rc"] == o["rc"] == 0 and r["rank"] == o["rank"] == n
    J = r["J"].cpu().numpy()
    if qrcp != 1:       # bqrrp's inner Gaussian sketch goes through different libm implementations (device vs host): pivots
        np.testing.assert_array_equal(J, o["J"])            # may differ on near-ties there; hqrrp's uniform sketch is exact
    k = r["rank"]
    Q, R = d.cm_to_numpy(Ad)[:, :k], d.cm_to_numpy(r["R"])[:k]
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= EPS**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= EPS**0.75 * np.sqrt(n)


# ---------------------------------------------------------------------------------------------------
# fp32 instantiations of the drivers (BASELINE config 4 is fp32).  Tolerances: the reference's eps^0.75 rule with
# float eps (test_bqrrp.cc is typed on T), checked in float64 arithmetic on the host.
# ---------------------------------------------------------------------------------------------------
def _cm32(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a.T.astype(np.float32))).cuda()


EPS32 = float(np.finfo(np.float32).eps)


@pytest.mark.parametrize("opts", [(0, 1, 1), (1, 1, 1), (0, 2, 0)])
def test_bqrrp_f32(ctx, orc, opts):
    d = _d()
    rng = np.random.default_rng(8)
    m, n, b = 1000, 400, 100
    A = rng.standard_normal((m, n)).astype(np.float32).astype(np.float64)
    Ad = _cm32(A)
    r = d.drv_bqrrp(ctx, Ad, m, n, b, 1.0, key=(3, 0), qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2])
    assert r["rc"] == 0 and r["rank"] == n
    Aout = d.cm_to_numpy(Ad).astype(np.float64)
    _bqrrp_verify(orc, A, Aout, r["tau"].cpu().numpy().astype(np.float64), r["J"].cpu().numpy(), atol=EPS32**0.75 * 4)


def test_cqrrpt_hqrrp_rsvd_f32(ctx, orc):
    d = _d()
    rng = np.random.default_rng(9)
    m, n = 20000, 256
    A = poly_mat(m, n, n, rng, cond=1e3).astype(np.float32).astype(np.float64)
    Ad = _cm32(A)
    r = d.drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, key=(1, 0))
    k = r["rank"]
    assert r["rc"] == 0 and k == n
    Q, R, J = d.cm_to_numpy(Ad)[:, :k].astype(np.float64), d.cm_to_numpy(r["R"])[:k].astype(np.float64), r["J"].cpu().numpy()
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= EPS32**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= EPS32**0.75 * np.sqrt(n)
    Ad = _cm32(A[:2000])
    r = d.drv_hqrrp(ctx, Ad, 2000, n, 32, 8)
    _bqrrp_verify(orc, A[:2000], d.cm_to_numpy(Ad).astype(np.float64), r["tau"].cpu().numpy().astype(np.float64), r["J"].cpu().numpy(),
                  atol=EPS32**0.75)
    Ad = _cm32(A)
    r = d.drv_rsvd(ctx, Ad, m, n, 32, 32, 1e-5, 2, 1)
    U, S, V = (d.cm_to_numpy(r["U"]).astype(np.float64), r["S"].cpu().numpy().astype(np.float64), d.cm_to_numpy(r["V"]).astype(np.float64))
    assert r["k"] == 32
    assert np.linalg.norm(U.T @ U - np.eye(32)) <= EPS32**0.75 * 10
    o = orc.rsvd(A, 32, 32, 1e-5, 2, 1)
    # same sketch stream in both precisions (the Gaussian is generated in fp64 and rounded): approximation quality equal to fp64's
    assert np.linalg.norm(A - (U * S) @ V.T) <= np.linalg.norm(A - (o["U"] * o["S"]) @ o["V"].T) * (1 + 1e-3)


# ---------------------------------------------------------------------------------------------------
# ABRIK (drivers/rl_abrik.hh) on a dense device operator vs the oracle (same Philox stream on both sides)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k,iters", [(400, 300, 8, 2), (400, 300, 8, 5), (400, 300, 8, 12), (300, 500, 4, 9), (2000, 2000, 32, 8)])
def test_abrik_vs_oracle(ctx, orc, m, n, k, iters):
    d = _d()
    rng = np.random.default_rng(m + k + iters)
    s = np.logspace(0, -6, min(m, n))
    A = (np.linalg.qr(rng.standard_normal((m, len(s))))[0] * s) @ np.linalg.qr(rng.standard_normal((n, len(s))))[0].T
    r = d.drv_abrik(ctx, d.cm_from_numpy(A), m, n, k, 1e-12, iters, key=(1, 0))
    o = orc.abrik(A, k, 1e-12, iters, key=(1, 0))
    assert r["rc"] == o["rc"] == 0
    assert (r["iters"], r["triplets"], r["next_ctr"]) == (o["iters"], o["triplets"], o["next_ctr"])
    U, S, V = d.cm_to_numpy(r["U"]), r["S"].cpu().numpy(), d.cm_to_numpy(r["V"])
    t = r["triplets"]
    kk = min(k, t)
    assert np.max(np.abs(S[:kk] - o["S"][:kk]) / o["S"][:kk]) <= 1e-9     # leading block: same numbers as the reference path
    assert np.max(np.abs(S - o["S"])) <= 1e-6 * o["S"][0]                  # whole Ritz spectrum (the tail is not converged)
    assert abs(r["norm_R_end"] - o["norm_R_end"]) <= 1e-8 * o["norm_R_end"]
    assert np.linalg.norm(U.T @ U - np.eye(t)) <= 1e-10 and np.linalg.norm(V.T @ V - np.eye(t)) <= 1e-10
    assert min(np.linalg.norm(A.T @ U - V * S), np.linalg.norm(A @ V - U * S)) <= 1e-10


def test_abrik_early_termination_and_bad_args(ctx, orc):
    d = _d()
    rng = np.random.default_rng(3)
    A = rng.standard_normal((300, 20)) @ rng.standard_normal((20, 200))
    r = d.drv_abrik(ctx, d.cm_from_numpy(A), 300, 200, 8, 1e-12, 50, key=(2, 0))
    o = orc.abrik(A, 8, 1e-12, 50, key=(2, 0))
    assert (r["iters"], r["triplets"]) == (o["iters"], o["triplets"])
    assert r["iters"] <= 8
    np.testing.assert_allclose(r["S"].cpu().numpy(), o["S"], rtol=1e-8, atol=1e-10 * o["S"][0])
    from randlapack_amd import _lib

    with pytest.raises(_lib.RlhipError):
        d.drv_abrik(ctx, d.cm_from_numpy(A), 300, 200, 0, 1e-12, 5)         # k must be > 0 (rl_abrik.hh:176)


# ---------------------------------------------------------------------------------------------------
# CQRRT (drivers/rl_cqrrt.hh) and ABRIK with CQRRT panels (qr_exp = cqrrt)
# ---------------------------------------------------------------------------------------------------
# (square input: d_factor = 2 as in test_cqrrt.cc:153-173; a 2-nonzeros-per-column operator with d = 1.25 n is likely to hold two parallel
#  columns when m <= d, under RandBLAS's independent-column distribution as well)
@pytest.mark.parametrize("m,n,cond,d_factor,nnz", [(5000, 200, 1e2, 1.25, 2), (2000, 64, 1e8, 1.25, 2), (300, 300, 1e3, 2.0, 4)])
def test_cqrrt_vs_oracle_shared_sketch(ctx, orc, m, n, cond, d_factor, nnz):
    d = _d()
    rng = np.random.default_rng(m + n)
    A = poly_mat(m, n, n, rng, cond=cond)
    Ad = d.cm_from_numpy(A)
    r = d.drv_cqrrt(ctx, Ad, m, n, d_factor, nnz, want_sketch=True, key=(4, 0))
    o = orc.cqrrt(A, d.cm_to_numpy(r["sketch"]))
    assert r["rc"] == o["rc"] == 0
    Q, R = d.cm_to_numpy(Ad), np.triu(d.cm_to_numpy(r["R"]))
    assert np.linalg.norm(R - o["R"]) <= EPS**0.6 * np.linalg.norm(o["R"])
    assert np.linalg.norm(A - Q @ R) <= EPS**0.75 * np.linalg.norm(A)                  # test_cqrrt.cc error check
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= EPS**0.75 * np.sqrt(n)


def test_abrik_cqrrt_panels_match_householder_panels(ctx, orc):
    d = _d()
    rng = np.random.default_rng(21)
    m, n, k, iters = 1500, 1200, 16, 10
    s = np.logspace(0, -5, n)
    A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    Ad = d.cm_from_numpy(A)
    r0 = d.drv_abrik(ctx, Ad, m, n, k, 1e-12, iters, key=(1, 0), qr_exp=0)
    r1 = d.drv_abrik(ctx, Ad, m, n, k, 1e-12, iters, key=(1, 0), qr_exp=1)
    assert (r0["iters"], r0["triplets"]) == (r1["iters"], r1["triplets"])
    S0, S1 = r0["S"].cpu().numpy(), r1["S"].cpu().numpy()
    # same Krylov subspaces whatever the panel QR: the converged Ritz values coincide
    assert np.max(np.abs(S0[:k] - S1[:k]) / S0[:k]) <= 1e-9
    assert np.all(S1 <= s[:len(S1)] * (1 + 1e-10))                          # Ritz values stay below the singular values
    U, V = d.cm_to_numpy(r1["U"]), d.cm_to_numpy(r1["V"])
    t = r1["triplets"]
    assert np.linalg.norm(U.T @ U - np.eye(t)) <= 1e-9 and np.linalg.norm(V.T @ V - np.eye(t)) <= 1e-9


# ---------------------------------------------------------------------------------------------------
# Degenerate shapes through every driver (block larger than the matrix, single columns, wide inputs, one iteration)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,b,kw", [(300, 40, 64, {}), (50, 20, 1, {}), (50, 1, 4, {}), (64, 64, 64, {}), (20, 100, 32, {}),
                                      (301, 97, 25, dict(qrcp_wide=1))])
def test_bqrrp_degenerate_shapes(ctx, orc, m, n, b, kw):
    d = _d()
    rng = np.random.default_rng(m * 7 + n + b)
    A = rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A)
    r = d.drv_bqrrp(ctx, Ad, m, n, b, 1.0, **kw)
    assert r["rc"] == 0 and r["rank"] == min(m, n)
    _bqrrp_verify(orc, A, d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy())


@pytest.mark.parametrize("m,n,nb,pp", [(100, 10, 32, 5), (100, 1, 8, 2), (100, 40, 8, 0), (30, 100, 16, 4)])
def test_hqrrp_degenerate_shapes(ctx, orc, m, n, nb, pp):
    d = _d()
    rng = np.random.default_rng(m + n + nb)
    A = rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A)
    r = d.drv_hqrrp(ctx, Ad, m, n, nb, pp)
    o = orc.hqrrp(A, nb, pp)
    np.testing.assert_array_equal(r["J"].cpu().numpy(), o["J"])
    _bqrrp_verify(orc, A, d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy())


def test_small_shapes_rsvd_cqrrpt_abrik_stabilisers(ctx, orc):
    d = _d()
    from randlapack_amd import _lib

    rng = np.random.default_rng(4)
    for (m, n, k, b) in [(100, 50, 1, 1), (100, 20, 20, 20), (100, 50, 5, 16), (30, 80, 10, 10)]:
        A = rng.standard_normal((m, n))
        r = d.drv_rsvd(ctx, d.cm_from_numpy(A), m, n, k, b, 1e-12, 2, 1)
        o = orc.rsvd(A, k, b, 1e-12, 2, 1)
        assert (r["rc"], r["qb_rc"], r["k"]) == (o["rc"], o["qb_rc"], o["k"])
        np.testing.assert_allclose(r["S"].cpu().numpy(), o["S"], rtol=1e-9, atol=1e-10 * o["S"][0])
    for (m, n) in [(10, 8), (1000, 3)]:
        A = rng.standard_normal((m, n))
        Ad = d.cm_from_numpy(A)
        r = d.drv_cqrrpt(ctx, Ad, m, n, 1.25, 2)
        k = r["rank"]
        Q, R, J = d.cm_to_numpy(Ad)[:, :k], d.cm_to_numpy(r["R"])[:k], r["J"].cpu().numpy()
        assert r["rc"] == 0 and k == n and np.linalg.norm(A[:, J - 1] - Q @ R) <= 1e-13 * np.linalg.norm(A)
    with pytest.raises(_lib.RlhipError):       # d = (int)(1.25 * 1) = 1 row cannot hold 2 nonzeros per column (RandBLAS rejects it too)
        d.drv_cqrrpt(ctx, d.cm_from_numpy(rng.standard_normal((100, 1))), 100, 1, 1.25, 2)
    for (m, n, k, it) in [(100, 80, 4, 1), (60, 40, 20, 6), (40, 100, 4, 6)]:
        A = rng.standard_normal((m, n))
        r = d.drv_abrik(ctx, d.cm_from_numpy(A), m, n, k, 1e-12, it)
        o = orc.abrik(A, k, 1e-12, it)
        assert (r["iters"], r["triplets"]) == (o["iters"], o["triplets"])
        np.testing.assert_allclose(r["S"].cpu().numpy()[:k], o["S"][:k], rtol=1e-8)
    for kind in (0, 1, 2):
        for (m, k) in [(50, 1), (16, 16)]:
            Y = rng.standard_normal((m, k))
            Yd = d.cm_from_numpy(Y)
            rc, _ = d.drv_stab(ctx, kind, Yd, m, k)
            rco, Qo = orc.stab(kind, Y)
            assert rc == rco == 0
            np.testing.assert_allclose(d.cm_to_numpy(Yd), Qo, atol=1e-11, rtol=0)


def test_no_device_memory_growth_over_repeated_calls(ctx, orc):
    """Steady state allocates nothing: scratch comes from the context's arena, outputs from the caching pool behind rlhip_malloc."""
    import torch

    d = _d()
    rng = np.random.default_rng(2)
    A = d.cm_from_numpy(rng.standard_normal((3000, 400)))
    Aq = rng.standard_normal((3000, 64))

    def one_round():
        r = d.drv_rsvd(ctx, A, 3000, 400, 32, 16, 1e-12, 1, 1)
        del r
        Ad = d.cm_from_numpy(Aq)
        d.drv_cqrrpt(ctx, Ad, 3000, 64, 1.25, 2)
        Ab = d.cm_from_numpy(Aq)
        d.drv_bqrrp(ctx, Ab, 3000, 64, 16, 1.0)
        Ah = d.cm_from_numpy(Aq)
        d.drv_hqrrp(ctx, Ah, 3000, 64, 16, 4)
        d.drv_abrik(ctx, A, 3000, 400, 8, 1e-12, 4)

    for _ in range(3):
        one_round()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        one_round()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 <= 8 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 25 rounds"


def test_cqrrpt_orthogonalization_mode(ctx, orc):
    """CQRRPT::orthogonalization (rl_cqrrpt.hh:347-367): rank-deficient input -> all n columns of A come back orthonormal, the
    first `rank` of them spanning range(A[:, J])."""
    d = _d()
    rng = np.random.default_rng(15)
    m, n, r0 = 3000, 80, 50
    A = poly_mat(m, n, r0, rng, cond=1e3)
    Ad = d.cm_from_numpy(A)
    r = d.drv_cqrrpt(ctx, Ad, m, n, 1.25, 4, key=(2, 0), qrcp=16 + 2)
    k = r["rank"]
    assert r["rc"] == 0 and abs(k - r0) <= 5
    Q = d.cm_to_numpy(Ad)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= EPS**0.75 * np.sqrt(n) * 10
    J = r["J"].cpu().numpy()
    AP = A[:, J - 1]
    assert np.linalg.norm(AP - Q[:, :k] @ (Q[:, :k].T @ AP)) <= 1e-9 * np.linalg.norm(A)     # the leading columns carry the range


def test_bqrrp_cholqr_panel_breakdown_falls_back_to_householder(ctx, orc):
    """Kahan matrix (test matrix of BQRRP_error_analysis.cc): the preconditioned panels reach the noise floor and a panel's Gram
    matrix stops being positive definite.  The reference carries on with the half-factored Gram matrix (its CPU run returns a Q
    with ||Q'Q - I|| = 1); the device driver refactors that panel with Householder reflectors.  Same pivots as the oracle, and a
    valid QR."""
    d = _d()
    m = n = 512
    A0 = d.drv_mat_gen(ctx, "kahan", m, n, theta=1.2, perturb=1e3)["A"]
    A0n = d.cm_to_numpy(A0)
    A = A0.clone()
    out = d.drv_bqrrp(ctx, A, m, n, 64, 1.0, qr_tall=1, want_sketch=True)
    tau, J = out["tau"].cpu().numpy(), out["J"].cpu().numpy()
    ref = orc.bqrrp(A0n, 64, 1.0, qr_tall=1, sketch=d.cm_to_numpy(out["sketch"]))
    Qr = orc.ungqr(ref["A"], ref["tau"])
    assert np.linalg.norm(Qr.T @ Qr - np.eye(n)) > 0.5           # what the reference's "graceful" handling produces here
    np.testing.assert_array_equal(J, ref["J"])
    assert out["rank"] == ref["rank"] == n
    Q = orc.ungqr(d.cm_to_numpy(A), tau)
    R = np.triu(d.cm_to_numpy(A))
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-11 and 0 <= tau.min() and tau.max() <= 2.0 + 1e-12
    assert np.linalg.norm(A0n[:, J - 1] - Q @ R) / np.linalg.norm(A0n) < 1e-13
