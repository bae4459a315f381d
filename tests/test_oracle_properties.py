"""The reference's own property tests, run against the CPU oracle at the reference's
sizes and tolerances.  This is what 'pins' the factorization part of the oracle: the reference holds no
golden factors, only these norm bounds."""
import numpy as np
import pytest

from _gen import poly_mat, poly_singvals, with_singvals

EPS = np.finfo(np.float64).eps


def _qb_checks(A, Q, B, k, tol_test, A_k=None):
    # test/comps/test_qb.cc:162-174
    assert np.linalg.norm(Q.T @ Q - np.eye(Q.shape[1])) <= tol_test
    if A_k is not None:
        assert np.linalg.norm(A_k - Q @ B) <= tol_test * 10


@pytest.mark.parametrize("b_sz,p", [(2, 5), (10, 2), (10, 5), (2, 2)])
def test_qb_polynomial_decay(orc, b_sz, p):
    # test_qb.cc:236-363: 100x100, k=50, polynomial decay cond 2025, tol = eps^0.75
    rng = np.random.default_rng(0)
    m = n = 100
    k = 50
    A = with_singvals(m, n, np.concatenate([poly_singvals(k, 0.1, 2025.0, 2.0)]), rng)  # exact rank k
    tol = EPS**0.75
    rc, kf, Q, BT, _ = orc.qb(A, k, b_sz, tol, p, 1, rs_stab=0, rf_orth=0, qb_orth=0, orth_check=False)
    assert rc in (0, 3) and kf == k
    assert np.linalg.norm(Q.T @ Q - np.eye(kf)) <= EPS**0.625
    assert np.linalg.norm(A - Q @ BT.T) <= EPS**0.625 * np.linalg.norm(A)


def test_qb_zero_tol_exact_rank(orc):
    # test_qb.cc:219-229: ||A - QB|| <= tol ||A|| once the whole rank is captured
    rng = np.random.default_rng(1)
    A = with_singvals(100, 100, poly_singvals(20, 0.1, 100.0, 2.0), rng)
    rc, kf, Q, BT, _ = orc.qb(A, 30, 10, 0.0, 2, 1)
    assert np.linalg.norm(A - Q @ BT.T) <= 1e-9 * np.linalg.norm(A)


def test_rf_hqrq(orc):
    # test/comps/test_rf.cc:143-201: 100x100, k in {100, 50}, p=5, HQRQ; ||Q^T Q - I|| <= eps^0.625
    rng = np.random.default_rng(2)
    for k in (100, 50):
        A = poly_mat(100, 100, 100, rng, cond=2025.0)
        rc, Q, _ = orc.rf(A, k, 5, 1, rs_stab=1, orth_kind=1)
        assert rc == 0
        assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= EPS**0.625


def test_cholqrq_twice(orc):
    # test/comps/test_orth.cc:135-153: CholQRQ twice on a 1000x200 Gaussian-sketched matrix
    rng = np.random.default_rng(3)
    Y = poly_mat(1000, 300, 300, rng, cond=100.0) @ rng.standard_normal((300, 200))
    rc, Q1 = orc.stab(0, Y)
    assert rc == 0
    rc, Q2 = orc.stab(0, Q1)
    assert rc == 0
    assert np.linalg.norm(Q2.T @ Q2 - np.eye(200)) <= EPS**0.625


def test_cholqrq_reports_failure(orc):
    A = np.ones((50, 4))  # rank 1 -> Cholesky breaks down -> return 1 (rl_orth.hh:81-85)
    rc, _ = orc.stab(0, A)
    assert rc == 1


def test_plul_singular_input(orc):
    # test_orth.cc:109-133: PLUL on a singular input returns 0, entries finite and <= 1 in magnitude
    A = np.zeros((20, 5))
    A[:, 0] = 1.0
    rc, L = orc.stab(2, A)
    assert rc == 0 and np.all(np.isfinite(L)) and np.abs(L).max() <= 1.0


def test_rsvd_config1(orc):
    # BASELINE.json configs[0]: RSVD 4096x512 fp64, rank 64, p=2, q=1, tol=eps^0.5625 (test_rsvd.cc:176)
    rng = np.random.default_rng(4)
    m, n, k = 4096, 512, 64
    A = poly_mat(m, n, n, rng)
    r = orc.rsvd(A, k, k, EPS**0.5625, 2, 1)
    assert r["rc"] == 0 and r["qb_rc"] == 3 and r["k"] == k  # fixed-rank run ends with QB code 3
    s_true = np.linalg.svd(A, compute_uv=False)
    A_k_err = np.sqrt(np.sum(s_true[k:] ** 2))
    err = np.linalg.norm(A - (r["U"] * r["S"]) @ r["V"].T)
    assert err <= 1.25 * A_k_err + 1e-12      # two power passes get close to the optimal rank-k error
    np.testing.assert_allclose(r["S"][:10], s_true[:10], rtol=1e-6)
    assert np.linalg.norm(r["U"].T @ r["U"] - np.eye(k)) <= EPS**0.625
    assert np.linalg.norm(r["V"].T @ r["V"] - np.eye(k)) <= EPS**0.625
    assert r["next_ctr"] == (n * k // 4, 0, 0, 0)


def test_rsvd_argument_checks(orc):
    A = np.zeros((4, 4))
    assert orc.rsvd(A, 2, 2, -1.0, 0, 1)["rc"] == -1   # tol < 0  (rl_rsvd.hh:131)


def test_cqrrpt_rank_deficient(orc):
    # test/drivers/test_cqrrpt.cc:184-304 scaled down: m x 200, rank 100; eps = eps^0.85
    rng = np.random.default_rng(5)
    m, n, rank = 4000, 200, 100
    A = poly_mat(m, n, rank, rng, cond=1e6)
    d = int(1.25 * n)
    S = rng.standard_normal((d, m))
    r = orc.cqrrpt(A, S @ A, EPS**0.85)
    assert r["rc"] == 0
    assert abs(r["rank"] - rank) <= 5                      # test_cqrrpt.cc:178-179
    k = r["rank"]
    Q, R, J = r["Q"][:, :k], r["R"][:k, :], r["J"]
    atol = EPS**0.75
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= 100 * atol * np.linalg.norm(A)   # :102-104
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= atol * np.sqrt(n) * 100
    assert sorted(J.tolist()) == list(range(1, n + 1))


def test_cqrrpt_zero_matrix(orc):
    r = orc.cqrrpt(np.zeros((50, 5)), np.zeros((7, 5)), EPS**0.85)
    assert r["rc"] == 0 and r["rank"] == 0                 # rl_cqrrpt.hh:256-261


def test_orhr_col_matches_lapack_semantics(orc):
    # Householder reconstruction: Q - S = L U, V = unit-lower L, tau_i = -u_ii * d_i  (rl_util.hh:339-379)
    rng = np.random.default_rng(6)
    m, n = 60, 12
    Q = np.linalg.qr(rng.standard_normal((m, n)))[0]
    A, tau, D = orc.orhr_col(Q, output_tau=True)
    V = np.tril(A, -1) + np.eye(m, n)
    H = np.eye(m)
    for i in range(n):
        v = V[:, i:i + 1]
        H = H @ (np.eye(m) - tau[i] * (v @ v.T))
    # H[:, :n] = Q * diag(D): reconstructed reflectors reproduce Q up to the sign vector
    np.testing.assert_allclose(H[:, :n], Q * D, atol=1e-12)
    assert set(np.unique(D)).issubset({-1.0, 1.0})


# ---------------------------------------------------------------------------------------------------
# BQRRP (drivers/rl_bqrrp.hh) -- the reference's test/drivers/test_bqrrp.cc cases, scaled down.
# Checks are the reference's (test_bqrrp.cc:105-145): after ungqr + col_swap,
#   ||A[:, J] - Q R||_F <= eps^0.75 ||A||_F,  ||Q^T Q - I||_F <= eps^0.75 sqrt(min(m,n)),  all-zero input -> A == 0.
# ---------------------------------------------------------------------------------------------------
def _bqrrp_checks(orc, A, r, atol=EPS**0.75):
    m, n = A.shape
    mn = min(m, n)
    Q = orc.ungqr(r["A"], r["tau"])
    R = np.triu(r["A"])[:mn]
    J = r["J"]
    assert sorted(J.tolist()) == list(range(1, n + 1))
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= atol * max(np.linalg.norm(A), 1e-300)
    assert np.linalg.norm(Q.T @ Q - np.eye(mn)) <= atol * np.sqrt(mn)


@pytest.mark.parametrize("b_sz", [100, 140])
@pytest.mark.parametrize("opts", [(0, 2, 0), (1, 1, 1), (0, 0, 1), (0, 1, 0)])
def test_bqrrp_full_rank(orc, b_sz, opts):
    # test_bqrrp.cc:151-186 (5000 x 2000, b = 500 / 700) at 1/5 scale, every subroutine family
    rng = np.random.default_rng(7)
    A = rng.standard_normal((1000, 400))
    r = orc.bqrrp(A, b_sz, 1.0, qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2], key=(b_sz, 0))
    assert r["rc"] == 0 and r["rank"] == 400
    _bqrrp_checks(orc, A, r)


@pytest.mark.parametrize("opts", [(0, 1, 1), (0, 2, 0), (1, 0, 0)])
def test_bqrrp_single_precision_leg(orc, opts):
    """the restatement instantiated on float (oracle_bqrrp_f32: s-prefixed LAPACK): the reference's BQRRP checks at float eps
    (test_bqrrp.cc is typed on T), the same Gaussian stream rounded to float, and -- on columns whose norms are separated far beyond
    float rounding -- the same pivots as the double run"""
    rng = np.random.default_rng(7)
    m, n, b = 600, 240, 60
    A = (rng.standard_normal((m, n)) * 1.02 ** (-rng.permutation(n).astype(np.float64))).astype(np.float32)
    r32 = orc.bqrrp(A, b, 1.0, qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2], key=(3, 0))
    r64 = orc.bqrrp(A.astype(np.float64), b, 1.0, qrcp_wide=opts[0], qr_tall=opts[1], apply_trans_q=opts[2], key=(3, 0))
    assert r32["A"].dtype == np.float32 and r32["tau"].dtype == np.float32
    assert r32["rc"] == r64["rc"] == 0 and r32["rank"] == r64["rank"] == n and r32["next_ctr"] == r64["next_ctr"]
    np.testing.assert_array_equal(r32["J"], r64["J"])
    eps32 = float(np.finfo(np.float32).eps)
    A32, t32 = r32["A"].astype(np.float64), r32["tau"].astype(np.float64)
    Q = orc.ungqr(A32, t32)
    R = np.triu(A32)[:n]
    assert np.linalg.norm(A.astype(np.float64)[:, r32["J"] - 1] - Q @ R) <= eps32**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= eps32**0.75 * np.sqrt(n)
    assert np.linalg.norm(R - np.triu(r64["A"])[:n]) <= 50 * eps32 * np.linalg.norm(R) * np.sqrt(n)


def test_bqrrp_low_rank_and_step_spectrum(orc):
    rng = np.random.default_rng(8)
    A = poly_mat(600, 300, 120, rng, cond=1e3)                 # test_bqrrp.cc:188-207 (low rank)
    r = orc.bqrrp(A, 50, 1.0, key=(1, 0))
    assert r["rc"] == 0 and 120 <= r["rank"] <= 150            # rank is rounded up to a block boundary
    _bqrrp_checks(orc, A, r)
    n = 256                                                    # :209-240 step spectrum, cond 1e10, qrcp_wide = geqp3
    s = np.ones(n); s[n // 2:] = 1e-10
    A = (np.linalg.qr(rng.standard_normal((n, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    r = orc.bqrrp(A, 32, 1.0, qrcp_wide=1, key=(2, 0))
    assert r["rc"] == 0
    _bqrrp_checks(orc, A, r)
    # the leading half of the pivots must capture the large singular directions: |R[n/2-1, n/2-1]| >> |R[n/2, n/2]|
    dR = np.abs(np.diag(r["A"]))
    assert dR[n // 2 - 1] > 1e6 * dR[n // 2]


@pytest.mark.parametrize("kind", ["zero", "near_zero", "half_zero"])
def test_bqrrp_degenerate_inputs(orc, kind):
    # test_bqrrp.cc:330-412
    rng = np.random.default_rng(9)
    m, n, b = 300, 120, 30
    if kind == "zero":
        A = np.zeros((m, n))
    elif kind == "near_zero":
        A = 1e-20 * rng.standard_normal((m, n))
    else:
        A = rng.standard_normal((m, n)); A[:, n // 2:] = 0
    r = orc.bqrrp(A, b, 1.0, key=(3, 0))
    assert r["rc"] == 0
    if kind == "zero":
        assert r["rank"] == 0 and np.all(r["A"] == 0)
    if r["rank"] == 0:
        assert np.abs(r["A"]).max() <= EPS**0.75               # test_bqrrp.cc:126-129: ASSERT_NEAR(A[i], 0, atol)
    else:
        _bqrrp_checks(orc, A, r)
    if kind == "near_zero":
        assert r["rank"] == 0                                  # every entry is below eps: rl_bqrrp.hh:373-399
    if kind == "half_zero":
        assert r["rank"] <= n // 2 + b


def test_bqrrp_wide(orc):
    # test_bqrrp.cc:414-438 (2000 x 5000) at 1/10 scale
    rng = np.random.default_rng(10)
    A = rng.standard_normal((200, 500))
    r = orc.bqrrp(A, 64, 1.0, key=(4, 0))
    assert r["rc"] == 0 and r["rank"] == 200
    _bqrrp_checks(orc, A, r)


def test_bqrrp_internal_nb_and_state(orc):
    # cholqr with internal_nb < b (test_bqrrp.cc:300-328) and the RNG-state contract: one d x m fill_dense per call
    rng = np.random.default_rng(11)
    A = rng.standard_normal((500, 280))
    r = orc.bqrrp(A, 90, 1.0, qrcp_wide=0, qr_tall=1, apply_trans_q=1, internal_nb=30, key=(5, 0))
    assert r["rc"] == 0
    _bqrrp_checks(orc, A, r)
    assert r["next_ctr"][0] == (90 * 500 + 3) // 4


# ---------------------------------------------------------------------------------------------------
# HQRRP (drivers/rl_hqrrp.hh) -- test/drivers/test_hqrrp.cc style checks: GEQP3-format output verified through
# ungqr + col_swap; plus pivot quality against LAPACK's geqp3 on a graded matrix.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,nb,pp", [(300, 120, 32, 5), (200, 200, 64, 10), (150, 260, 32, 8), (500, 70, 16, 4)])
@pytest.mark.parametrize("qr_type,panel_pivoting", [(0, 1), (0, 0), (1, 0), (2, 0)])
def test_hqrrp_factorization(orc, m, n, nb, pp, qr_type, panel_pivoting):
    rng = np.random.default_rng(m + n + nb)
    A = poly_mat(m, n, min(m, n), rng, cond=1e4)
    r = orc.hqrrp(A, nb, pp, panel_pivoting, qr_type, key=(m, 0))
    assert r["rc"] == 0
    # same GEQP3-format verification as BQRRP; CholQR panels (qr_type 2, no pivoting inside the panel) lose
    # cond(panel)^2 * eps of orthogonality by construction -- the reference has the same property
    _bqrrp_checks(orc, A, r, atol=(1e-8 if qr_type == 2 else EPS**0.75))
    assert r["next_ctr"][0] == ((nb + pp) * m + 3) // 4


def test_hqrrp_pivot_quality_and_rank_reveal(orc):
    rng = np.random.default_rng(12)
    n = 96
    s = np.logspace(0, -10, n)
    A = (np.linalg.qr(rng.standard_normal((200, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    r = orc.hqrrp(A, 16, 8, 1, 0, key=(3, 0))
    dR = np.abs(np.diag(r["A"]))
    # |r_ii| tracks the singular values within a modest factor (randomized pivoting is a strong-RRQR in practice)
    assert np.all(dR[:60] <= s[:60] * 40) and np.all(dR[:60] >= s[:60] / 40)


# ---------------------------------------------------------------------------------------------------
# ABRIK (drivers/rl_abrik.hh) -- test/drivers/test_abrik.cc style: leading singular triplets against a full SVD,
# residual ||A^T U - V S|| or ||A V - U S|| (the side the last half-step closes exactly), early termination.
# ---------------------------------------------------------------------------------------------------
def _decay_mat(m, n, rng, lo=-6):
    s = np.logspace(0, lo, min(m, n))
    return (np.linalg.qr(rng.standard_normal((m, len(s))))[0] * s) @ np.linalg.qr(rng.standard_normal((n, len(s))))[0].T, s


@pytest.mark.parametrize("m,n,k,iters", [(400, 300, 8, 2), (400, 300, 8, 5), (400, 300, 8, 12), (300, 500, 4, 9), (600, 600, 16, 6)])
def test_abrik_dense_operator(orc, m, n, k, iters):
    rng = np.random.default_rng(m + k + iters)
    A, s = _decay_mat(m, n, rng)
    r = orc.abrik(A, k, 1e-12, iters, key=(1, 0))
    assert r["rc"] == 0 and r["iters"] == iters and r["triplets"] == iters * k // 2
    U, S, V = r["U"], r["S"], r["V"]
    t = r["triplets"]
    assert np.linalg.norm(U.T @ U - np.eye(t)) <= 1e-10 and np.linalg.norm(V.T @ V - np.eye(t)) <= 1e-10
    # one of the two residuals is at rounding level (which one depends on the parity of the last half-step, rl_abrik.hh:683-689)
    res = min(np.linalg.norm(A.T @ U - V * S), np.linalg.norm(A @ V - U * S))
    assert res <= 1e-10
    if iters >= 12:
        kk = min(k, t)
        assert np.max(np.abs(S[:kk] - s[:kk]) / s[:kk]) <= 1e-6      # leading block converged
    assert np.all(S <= s[:t] * (1 + 1e-12))                              # Ritz values never overshoot (Cauchy interlacing)
    assert r["next_ctr"][0] == (n * k + 3) // 4                          # one n x k Gaussian (:298-299)


def test_abrik_early_termination_on_exact_rank(orc):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((300, 20)) @ rng.standard_normal((20, 200))
    r = orc.abrik(A, 8, 1e-12, 50, key=(2, 0))
    # rank 20, block 8: the loop must stop long before max_krylov_iters = 50 (threshold :659-662 or diagonal test :455-458)
    assert r["rc"] == 0 and r["iters"] <= 8
    sref = np.linalg.svd(A, compute_uv=False)
    assert abs(r["S"][0] - sref[0]) <= 0.05 * sref[0]


# ---- linop QR drivers (test/drivers/test_orth_linop.cc: dense, sparse, composite, blocked; tolerance eps^0.75 via verify_qr)
def _linop_cases():
    import scipy.sparse as sp

    rng = np.random.default_rng(0)
    A = rng.standard_normal((100, 50))
    S = sp.random(100, 50, 0.2, random_state=rng, format="csr", data_rvs=rng.standard_normal)
    L = rng.standard_normal((100, 50))
    Rs = sp.random(50, 20, 0.3, random_state=rng, format="csr", data_rvs=rng.standard_normal)
    Sl = sp.random(100, 50, 0.2, random_state=rng, format="csr", data_rvs=rng.standard_normal)
    Rd = rng.standard_normal((50, 20))
    return [("dense", A, A), ("sparse", S, S.toarray()), ("dense*sparse", (L, Rs), L @ Rs.toarray()),
            ("sparse*dense", (Sl, Rd), Sl.toarray() @ Rd)]


@pytest.mark.parametrize("alg", ["cholqr", "scholqr3", "scholqr3_basic", "cqrrt"])
@pytest.mark.parametrize("block", [0, 10])
def test_qr_linops_oracle_properties(orc, alg, block):
    tol = np.finfo(np.float64).eps ** 0.75
    rng = np.random.default_rng(1)
    for name, op, A in _linop_cases():
        m, n = A.shape
        if alg == "cholqr":
            o = orc.cholqr_linops(op, block)
        elif alg == "scholqr3":
            o = orc.scholqr3_linops(op, block)
        elif alg == "scholqr3_basic":
            o = orc.scholqr3_linops(op, basic=True)
        else:
            Sk = rng.standard_normal((2 * n, m)) / np.sqrt(2 * n)
            o = orc.cqrrt_linops(op, Sk @ A, block)
        assert o["rc"] == 0, name
        Q, R = o["Q"], o["R"]
        assert np.allclose(R, np.triu(R))
        assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) <= tol, name
        assert np.linalg.norm(Q.T @ Q - np.eye(n)) / np.sqrt(n) <= tol, name
        # blocking is a memory optimisation only: same R as the unblocked call
        if block:
            o0 = {"cholqr": lambda: orc.cholqr_linops(op, 0), "scholqr3": lambda: orc.scholqr3_linops(op, 0),
                  "scholqr3_basic": lambda: orc.scholqr3_linops(op, basic=True), "cqrrt": lambda: orc.cqrrt_linops(op, Sk @ A, 0)}[alg]()
            np.testing.assert_allclose(R, o0["R"], rtol=0, atol=1e-12 * np.abs(R).max())


def test_qr_linops_oracle_failure_and_conditioning(orc):
    Z = np.ones((30, 3)); Z[:, 1] = 0                 # an exactly zero pivot
    assert orc.cholqr_linops(Z)["rc"] == 1
    assert orc.cqrrt_linops(np.zeros((30, 3)), np.zeros((6, 3)))["rc"] == 1
    # At cond 1e9 plain CholQR breaks down; CQRRT_linops and sCholQR3_linops still factor.  Their Q loses orthogonality like
    # cond * eps (not cond^2 * eps): the operator cannot be overwritten, so the Gram matrix is formed as M^T (A^T (A M)) and the
    # adjoint product sees the unpreconditioned A -- a property of the reference algorithm (rl_cqrrt_linops.hh:264-322).
    rng = np.random.default_rng(3)
    A = poly_mat(400, 40, 40, rng, cond=1e9)
    Sk = rng.standard_normal((80, 400)) / np.sqrt(80)
    o = orc.cqrrt_linops(A, Sk @ A)
    assert o["rc"] == 0 and np.linalg.norm(o["Q"].T @ o["Q"] - np.eye(40)) < 1e-5
    assert np.linalg.norm(A - o["Q"] @ o["R"]) / np.linalg.norm(A) < 1e-14
    o3 = orc.scholqr3_linops(A)
    assert o3["rc"] == 0 and np.linalg.norm(o3["Q"].T @ o3["Q"] - np.eye(40)) < 1e-5
    assert orc.cholqr_linops(A)["rc"] == 1


# ---- generators (test/misc/test_gen.cc: spectrum of the generated matrix equals the requested one)
@pytest.mark.parametrize("m_type,cond", [("polynomial", 1e6), ("exponential", 1e5), ("step", 1e4), ("bad_cholqr", 1e3)])
def test_mat_gen_oracle_spectra(orc, m_type, cond):
    m, n, k = 120, 60, 40
    A, nxt = orc.mat_gen(m_type, m, n, rank=k, cond_num=cond, exponent=2.0, key=(2, 0))
    want = {"polynomial": lambda: orc.gen_poly_singvals(k, 0.1, cond, 2.0), "exponential": lambda: orc.gen_exp_singvals(k, cond),
            "step": lambda: orc.gen_step_singvals(k, cond), "bad_cholqr": lambda: np.ones(k)}[m_type]()
    s = np.linalg.svd(A, compute_uv=False)
    np.testing.assert_allclose(s[:k], np.sort(want)[::-1], rtol=0, atol=1e-13)
    assert s[k:].max() < 1e-13                                             # exactly rank k
    if m_type in ("exponential", "step"):
        assert abs(want[0] / want[-1] - cond) < 1e-8 * cond                # condition number as requested
    if m_type == "polynomial":                                              # the reference's formula reaches 1/cond one index PAST the end
        assert 0.8 * cond < want[0] / want[-1] < cond
    assert nxt == ((m * k + 3) // 4 + (n * k + 3) // 4, 0, 0, 0)             # two fill_dense calls: U then V
    D, nxt_d = orc.mat_gen(m_type, m, n, rank=k, cond_num=cond, exponent=2.0, diag=True, key=(2, 0))
    assert D.shape == (k, k) and np.array_equal(np.diag(D), want) and nxt_d == (0, 0, 0, 0)


def test_mat_gen_oracle_spiked_adversarial_kahan(orc):
    m, n = 70, 12
    A, _ = orc.mat_gen("spiked", m, n, scaling=7.0, key=(1, 0))
    rows, _ = orc.repeated_fisher_yates(n // 2, m, key=(1, 0))
    assert len(set(rows.tolist())) == n // 2 and rows.min() >= 0 and rows.max() < m
    nr = np.linalg.norm(A, axis=1)
    mask = np.zeros(m, bool)
    mask[rows] = True
    np.testing.assert_allclose(nr[mask], 7.0, atol=1e-12)                  # rows of an orthogonal matrix, scaled
    np.testing.assert_allclose(nr[~mask], 1.0, atol=1e-12)
    B, _ = orc.mat_gen("adverserial", m, n + 8, scaling=1e-5, key=(1, 0))
    sb = np.linalg.svd(B, compute_uv=False)
    assert B.shape == (m, n + 8) and sb[0] < 2.0 and sb[-1] < 1e-6 * sb[0]          # graded, numerically rank deficient (as the reference warns)
    K, _ = orc.mat_gen("kahan", 9, 9, theta=1.2, perturb=1e3)
    c, s = np.cos(1.2), np.sin(1.2)
    assert np.allclose(K, np.triu(K)) and abs(K[2, 5] + c * s**2) < 1e-15 and abs(K[3, 3] - (s**3 + 1e3 * np.finfo(float).eps * 6)) < 1e-15


# ---- REVD2 / SYRF (test/drivers/test_revd2.cc: rank-deficient PSD input, rank doubling, uplo with NaNs in the unused triangle)
def _psd(m, rank, rng, decay=None):
    B = rng.standard_normal((m, rank))
    if decay is not None:
        B = B * decay
    return B @ B.T


def test_revd2_oracle_exact_rank_and_doubling(orc):
    rng = np.random.default_rng(0)
    m, rank = 200, 40
    A = _psd(m, rank, rng)
    out = orc.revd2(A, rank, 1e-8, key=(1, 0))
    assert out["k"] == rank                                                   # exact-rank input: one pass, no doubling
    rel = np.linalg.norm(A - (out["V"] * out["eigvals"]) @ out["V"].T) / np.linalg.norm(A)
    assert rel < 1e-12
    np.testing.assert_allclose(np.sort(out["eigvals"])[::-1], np.sort(np.linalg.eigvalsh(A))[::-1][:rank], rtol=1e-10)
    out2 = orc.revd2(A, 5, 1e-8, key=(1, 0))                                  # start too small: k doubles 5 -> 10 -> 20 -> 40
    assert out2["k"] == 40
    assert np.linalg.norm(A - (out2["V"] * out2["eigvals"]) @ out2["V"].T) / np.linalg.norm(A) < 1e-12


def test_revd2_oracle_uplo_with_nans(orc):
    rng = np.random.default_rng(1)
    m, rank = 120, 20
    A = _psd(m, rank, rng)
    Au, Al = np.triu(A), np.tril(A)
    Au[np.tril_indices(m, -1)] = np.nan
    Al[np.triu_indices(m, 1)] = np.nan
    ou = orc.revd2(Au, rank, 1e-8, uplo="U", key=(2, 0))
    ol = orc.revd2(Al, rank, 1e-8, uplo="L", key=(2, 0))
    Ru, Rl = (ou["V"] * ou["eigvals"]) @ ou["V"].T, (ol["V"] * ol["eigvals"]) @ ol["V"].T
    assert not np.isnan(Ru).any() and np.linalg.norm(Ru - Rl) < 1e-10 * np.linalg.norm(A)


def test_syrf_oracle_captures_range(orc):
    rng = np.random.default_rng(2)
    m, rank = 150, 30
    A = _psd(m, rank, rng)
    rc, Q, _ = orc.syrf(A, rank, 2, 1)
    assert rc == 0 and np.linalg.norm(Q.T @ Q - np.eye(rank)) < 1e-12
    assert np.linalg.norm(A - Q @ (Q.T @ A)) < 1e-10 * np.linalg.norm(A)


def test_benchmark_metric_helpers_cpu():
    """benchmarks/_common.trailing_norms == lantr on every trailing block (BQRRP_pivot_quality.cc:102-114), pure numpy"""
    import importlib.util
    import pathlib
    import sys
    import types

    src = (pathlib.Path(__file__).resolve().parent.parent / "benchmarks" / "_common.py").read_text()
    # the helper module imports torch / the device package at module scope; only the numpy helper is exercised here
    mod = types.ModuleType("bench_common_cpu")
    ns = mod.__dict__
    start = src.index("def trailing_norms")
    end = src.index("def singular_values")
    exec("import numpy as np\n" + src[start:end], ns)
    rng = np.random.default_rng(0)
    R = np.triu(rng.standard_normal((37, 37)))
    want = np.array([np.linalg.norm(R[i:, i:]) for i in range(37)])
    np.testing.assert_allclose(ns["trailing_norms"](R), want, rtol=1e-13)


# ---- operator block views and A + mu I (test infrastructure for tests/test_gpu_linops.py)
def test_oracle_linop_views_reproduce_the_reference_structure_cases(orc):
    """the structure cases of test/linops/test_linop_block_views.cc (rebased row pointers of a CSR row block, filtered and re-indexed
    columns of a CSR column block, composite views cut the left / right operand) on the numpy restatement"""
    import scipy.sparse as sp

    rng = np.random.default_rng(0)
    S = sp.random(20, 15, 0.3, random_state=np.random.default_rng(1), format="csr")
    D = S.toarray()
    rb = orc.linop_view(S, "row_block", (5, 0, 8, 15))
    assert rb.shape == (8, 15) and rb.indptr[0] == 0 and rb.nnz == S.indptr[13] - S.indptr[5]           # csr_row_block: rebased, nnz of the slice
    np.testing.assert_array_equal(rb.toarray(), D[5:13])
    cb = orc.linop_view(S, "col_block", (0, 4, 20, 7))
    assert cb.shape == (20, 7) and cb.indices.max() < 7                                                   # csr_col_block: indices re-based to the block
    np.testing.assert_array_equal(cb.toarray(), D[:, 4:11])
    np.testing.assert_array_equal(orc.linop_view(S, "submatrix", (12, 9, 8, 6)).toarray(), D[12:20, 9:15])
    L = rng.standard_normal((9, 20))
    lv, rv = orc.linop_view((L, S), "submatrix", (2, 3, 4, 5))
    np.testing.assert_allclose(lv @ rv.toarray(), (L @ D)[2:6, 3:8], atol=1e-14)
    for bad in (("row_block", (-1, 0, 3, 15)), ("row_block", (18, 0, 3, 15)), ("col_block", (0, 14, 20, 2)), ("submatrix", (0, 0, 0, 1))):
        with pytest.raises(ValueError):
            orc.linop_view(S, *bad)


def test_oracle_regsym_apply(orc):
    rng = np.random.default_rng(2)
    G0 = rng.standard_normal((12, 12)); G = G0 + G0.T
    B = rng.standard_normal((12, 3))
    U = np.triu(G) + np.tril(np.full((12, 12), 7.0), -1)                  # the strictly lower triangle is never read
    np.testing.assert_allclose(orc.regsym_apply(U, [0.5], False, B), G @ B, atol=1e-13)
    np.testing.assert_allclose(orc.regsym_apply(U, [0.5], True, B, alpha=2.0), 2.0 * (G + 0.5 * np.eye(12)) @ B, atol=1e-13)
    mus = [0.1, 0.2, 0.3]
    ref = np.column_stack([(G + mus[i] * np.eye(12)) @ B[:, i] for i in range(3)])
    np.testing.assert_allclose(orc.regsym_apply(U, mus, True, B), ref, atol=1e-13)
    with pytest.raises(ValueError):
        orc.regsym_apply(U, mus, True, B[:, :2])
