"""-m gpu: linear operators (dense / CSR / composite), the SpMM kernels behind them, and the linop QR drivers
(CholQR_linops, sCholQR3_linops[_basic], CQRRT_linops) against the numpy oracle.  The test matrix follows the reference's
test/drivers/test_orth_linop.cc (dense, dense float, sparse, composite dense*sparse / sparse*dense, blocked), with its tolerance
eps^0.75 on ||A - QR|| / ||A|| and ||Q^T Q - I|| / sqrt(n) (verify_qr, testing/rl_test_utils.hh)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
TOL = EPS**0.75


def _d():
    from randlapack_amd import device

    return device


def _sparse(m, n, density, seed):
    rng = np.random.default_rng(seed)
    return sp.random(m, n, density, random_state=rng, format="csr", data_rvs=rng.standard_normal)


def _ops(kind, seed=0):
    """-> (device operator, numpy oracle operator, dense ndarray)"""
    d = _d()
    rng = np.random.default_rng(seed)
    if kind == "dense":
        A = rng.standard_normal((300, 50))
        return d.DenseOperator(d.cm_from_numpy(A), 300, 50), A, A
    if kind == "sparse":
        S = _sparse(400, 50, 0.2, seed)
        return d.CsrOperator.from_scipy(S), S, S.toarray()
    if kind == "dense*sparse":
        L = rng.standard_normal((300, 60))
        S = _sparse(60, 20, 0.3, seed)
        return (d.DenseOperator(d.cm_from_numpy(L), 300, 60), d.CsrOperator.from_scipy(S)), (L, S), L @ S.toarray()
    if kind == "sparse*dense":
        S = _sparse(500, 70, 0.1, seed)
        Rm = rng.standard_normal((70, 30))
        return (d.CsrOperator.from_scipy(S), d.DenseOperator(d.cm_from_numpy(Rm), 70, 30)), (S, Rm), S.toarray() @ Rm
    if kind == "sparse*sparse":
        S1 = _sparse(500, 80, 0.1, seed)
        S2 = _sparse(80, 25, 0.4, seed + 1)
        return (d.CsrOperator.from_scipy(S1), d.CsrOperator.from_scipy(S2)), (S1, S2), S1.toarray() @ S2.toarray()
    raise ValueError(kind)


def _verify_qr(A, Q, R):
    n = A.shape[1]
    return np.linalg.norm(A - Q @ R) / np.linalg.norm(A), np.linalg.norm(Q.T @ Q - np.eye(n)) / np.sqrt(n)


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("nc", [1, 7, 16, 17, 32, 63, 64, 65, 130, 257, 600])
@pytest.mark.parametrize("layout", ["C", "R"])
def test_csr_spmm_matches_scipy(ctx, nc, layout):
    d = _d()
    import torch

    rng = np.random.default_rng(nc)
    m, k = 777, 333
    S = _sparse(m, k, 0.05, nc)
    S = sp.vstack([S[:100], sp.csr_matrix((5, k)), S[105:]]).tocsr()      # a run of empty rows
    op = d.CsrOperator.from_scipy(S)
    B = rng.standard_normal((k, nc))
    C0 = rng.standard_normal((m, nc))
    for alpha, beta in ((1.0, 0.0), (-0.5, 2.0)):
        if layout == "C":
            Bd, Cd = d.cm_from_numpy(B), d.cm_from_numpy(C0)
            ldb, ldc = k, m
        else:
            Bd, Cd = torch.as_tensor(B, device="cuda:0").contiguous(), torch.as_tensor(C0, device="cuda:0").contiguous()
            ldb, ldc = nc, nc
        rc = ctx.lib.rlhip_csr_spmm_f64(ctx.h, layout.encode(), m, nc, k, alpha, op.rowptr.data_ptr(), op.colidx.data_ptr(), op.vals.data_ptr(),
                                        Bd.data_ptr(), ldb, beta, Cd.data_ptr(), ldc)
        assert rc == 0
        got = d.cm_to_numpy(Cd) if layout == "C" else Cd.cpu().numpy()
        ref = alpha * (S @ B) + beta * C0
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(ref).max()))


def test_csr_transpose_and_densify(ctx):
    d = _d()
    import torch

    m, k = 500, 123
    S = _sparse(m, k, 0.07, 3)
    op = d.CsrOperator.from_scipy(S)
    nnz = S.nnz
    rpt = torch.zeros(k + 1, dtype=torch.int64, device="cuda:0")
    cit = torch.zeros(nnz, dtype=torch.int64, device="cuda:0")
    vt = torch.zeros(nnz, dtype=torch.float64, device="cuda:0")
    assert ctx.lib.rlhip_csr_transpose_f64(ctx.h, m, k, op.rowptr.data_ptr(), op.colidx.data_ptr(), op.vals.data_ptr(), rpt.data_ptr(),
                                           cit.data_ptr(), vt.data_ptr()) == 0
    St = sp.csr_matrix((vt.cpu().numpy(), cit.cpu().numpy(), rpt.cpu().numpy()), shape=(k, m))
    assert (St != S.T.tocsr()).nnz == 0
    ci = cit.cpu().numpy()
    rp = rpt.cpu().numpy()
    assert all(np.all(np.diff(ci[rp[j]:rp[j + 1]]) > 0) for j in range(k))       # deterministic order: ascending source row
    out = d.cm_empty(m, 40, device="cuda:0")
    assert ctx.lib.rlhip_csr_densify_cols_f64(ctx.h, m, rpt.data_ptr(), cit.data_ptr(), vt.data_ptr(), 17, 40, out.data_ptr(), m) == 0
    np.testing.assert_array_equal(d.cm_to_numpy(out), S.toarray()[:, 17:57])
    # out-of-range column index is an argument error, not a crash
    bad = op.colidx.clone()
    bad[0] = k
    assert ctx.lib.rlhip_csr_transpose_f64(ctx.h, m, k, op.rowptr.data_ptr(), bad.data_ptr(), op.vals.data_ptr(), rpt.data_ptr(),
                                           cit.data_ptr(), vt.data_ptr()) == -2


@pytest.mark.parametrize("case", ["short_rows", "one_dense_column", "long_and_short", "empty", "fp32_wide"])
def test_csr_transpose_both_routes_equal_scipy(ctx, case):
    """csr_transpose has two scatter routes (sparse.hip): transposed rows of at most 512 entries are scattered by entry number through
    atomics and sorted row by row; a longer row anywhere takes the stable counting sort over chunks.  Both must give scipy's transpose
    with ascending source rows inside every transposed row (= the order that makes A^T X bit-reproducible), twice the same bits."""
    d = _d()
    import torch

    rng = np.random.default_rng(7)
    dt, tdt, suf = np.float64, torch.float64, "f64"
    if case == "short_rows":
        m, k = 6000, 4100
        S = _sparse(m, k, 0.004, 11)
    elif case == "one_dense_column":                                     # column 5 holds 3000 entries: the counting-sort route
        m, k = 3000, 700
        S = _sparse(m, k, 0.01, 12).tolil(); S[:, 5] = rng.standard_normal((m, 1)); S = S.tocsr()
    elif case == "long_and_short":                                       # 513 entries in one column: just over the limit
        m, k = 2000, 90
        S = _sparse(m, k, 0.02, 13).tolil(); S[:, 40] = 0; S[:513, 40] = rng.standard_normal((513, 1)); S = S.tocsr()
    elif case == "empty":
        m, k = 50, 70
        S = sp.csr_matrix((m, k))
    else:
        m, k = 900, 30000
        S = _sparse(m, k, 0.002, 14).astype(np.float32)
        dt, tdt, suf = np.float32, torch.float32, "f32"
    S.sort_indices()
    nnz = S.nnz
    rowptr = torch.as_tensor(S.indptr.astype(np.int64), device="cuda:0")
    colidx = torch.as_tensor(S.indices.astype(np.int64), device="cuda:0") if nnz else torch.zeros(1, dtype=torch.int64, device="cuda:0")
    vals = torch.as_tensor(S.data.astype(dt), device="cuda:0") if nnz else torch.zeros(1, dtype=tdt, device="cuda:0")
    fn = getattr(ctx.lib, "rlhip_csr_transpose_" + suf)
    outs = []
    for rep in range(2):
        rpt = torch.full((k + 1,), -1, dtype=torch.int64, device="cuda:0")
        cit = torch.full((max(nnz, 1),), -1, dtype=torch.int64, device="cuda:0")
        vt = torch.zeros(max(nnz, 1), dtype=tdt, device="cuda:0")
        assert fn(ctx.h, m, k, rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), rpt.data_ptr(), cit.data_ptr(), vt.data_ptr()) == 0
        outs.append((rpt.cpu().numpy(), cit.cpu().numpy()[:nnz], vt.cpu().numpy()[:nnz]))
    rp, ci, v = outs[0]
    ref = S.T.tocsr(); ref.sort_indices()
    np.testing.assert_array_equal(rp, ref.indptr)
    np.testing.assert_array_equal(ci, ref.indices)                       # ascending source rows inside every transposed row
    np.testing.assert_array_equal(v, ref.data)
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("d_sk,nnz", [(120, 2), (500, 4), (2000, 8)])
def test_saso_apply_csr_matches_dense_path_and_is_reproducible(ctx, d_sk, nnz):
    d = _d()
    import ctypes as C
    import torch

    m, n = 3000, 90
    S = _sparse(m, n, 0.03, d_sk)
    S = sp.hstack([S[:, :40], sp.csr_matrix((m, 3)), S[:, 43:]]).tocsr()          # three empty columns
    A = S.toarray()
    St = S.T.tocsr()
    rpt = torch.as_tensor(St.indptr.astype(np.int64), device="cuda:0")
    cit = torch.as_tensor(St.indices.astype(np.int64), device="cuda:0")
    vt = torch.as_tensor(St.data, device="cuda:0")
    st = (C.c_uint32 * 4)(3, 0, 0, 0)
    key = (C.c_uint32 * 2)(7, 0)
    nxt = (C.c_uint32 * 4)()
    h = C.c_void_p()
    assert ctx.lib.rlhip_saso_create(ctx.h, d_sk, m, nnz, st, key, nxt, C.byref(h)) == 0
    rng = np.random.default_rng(0)
    B0 = rng.standard_normal((d_sk, n))
    outs = []
    for rep in range(2):
        Bd = d.cm_from_numpy(B0)
        assert ctx.lib.rlhip_saso_apply_csr_f64(ctx.h, h, n, 1.5, rpt.data_ptr(), cit.data_ptr(), vt.data_ptr(), -0.5, Bd.data_ptr(), d_sk, 0) == 0
        outs.append(d.cm_to_numpy(Bd))
    assert np.array_equal(outs[0], outs[1])                                       # integer accumulation: bitwise reproducible
    Bref = d.cm_from_numpy(B0)
    assert ctx.lib.rlhip_saso_apply_f64(ctx.h, h, n, 1.5, d.cm_from_numpy(A).data_ptr(), m, -0.5, Bref.data_ptr(), d_sk) == 0
    ref = d.cm_to_numpy(Bref)
    np.testing.assert_allclose(outs[0], ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    assert np.array_equal(outs[0][:, 40:43], -0.5 * B0[:, 40:43])                 # empty columns: beta * B only
    ctx.lib.rlhip_saso_destroy(ctx.h, h)


def test_cqrrt_linops_sparse_sketch_fallback_path(ctx, orc, monkeypatch):
    d = _d()
    op, op_np, A = _ops("sparse", 31)
    fast = d.drv_qr_linops(ctx, "cqrrt", op, d_factor=2.0, key=(1, 0), want_sketch=True)
    ctx.set_option("sparse_sketch_densify", 1)
    slow = d.drv_qr_linops(ctx, "cqrrt", op, d_factor=2.0, key=(1, 0), want_sketch=True)
    a, b = d.cm_to_numpy(fast["sketch"]), d.cm_to_numpy(slow["sketch"])
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-13 * np.abs(b).max())
    assert fast["next_ctr"] == slow["next_ctr"]


@pytest.mark.parametrize("kind", ["dense", "sparse", "dense*sparse", "sparse*dense", "sparse*sparse"])
def test_linop_apply_all_sides(ctx, kind):
    d = _d()
    rng = np.random.default_rng(5)
    op, _, A = _ops(kind, 5)
    m, n = A.shape
    nb = 37
    for side, trans in (("L", "N"), ("L", "T"), ("R", "N"), ("R", "T")):
        opA = A if trans == "N" else A.T
        if side == "L":
            B = rng.standard_normal((opA.shape[1], nb))
            ref = opA @ B
            got = d.linop_apply(ctx, op, side, trans, d.cm_from_numpy(B), opA.shape[0], nb, opA.shape[1])
        else:
            B = rng.standard_normal((nb, opA.shape[0]))
            ref = B @ opA
            got = d.linop_apply(ctx, op, side, trans, d.cm_from_numpy(B), nb, opA.shape[1], opA.shape[0])
        np.testing.assert_allclose(d.cm_to_numpy(got), ref, rtol=0, atol=1e-11 * np.abs(ref).max(), err_msg=f"{kind} {side} {trans}")


def test_trmm_left_upper(ctx):
    d = _d()
    rng = np.random.default_rng(2)
    n, k = 130, 77
    U = rng.standard_normal((n, n))           # lower triangle is garbage the routine must ignore
    B = rng.standard_normal((n, k))
    for trans in ("N", "T"):
        Bd = d.cm_from_numpy(B)
        assert ctx.lib.rlhip_trmm_f64(ctx.h, b"L", b"U", trans.encode(), b"N", n, k, 1.5, d.cm_from_numpy(U).data_ptr(), n, Bd.data_ptr(), n) == 0
        T = np.triu(U)
        ref = 1.5 * ((T if trans == "N" else T.T) @ B)
        np.testing.assert_allclose(d.cm_to_numpy(Bd), ref, rtol=0, atol=1e-12 * np.abs(ref).max())


# ------------------------------------------------------------------------------------------------ drivers
KINDS = ["dense", "sparse", "dense*sparse", "sparse*dense"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("alg,block", [("cholqr", 0), ("cholqr", 10), ("scholqr3", 0), ("scholqr3", 10), ("scholqr3_basic", 0)])
def test_cholqr_family_vs_oracle(ctx, orc, alg, kind, block):
    d = _d()
    op, op_np, A = _ops(kind, 11)
    out = d.drv_qr_linops(ctx, alg, op, block_size=block, want_Q=True)
    ref = {"cholqr": lambda: orc.cholqr_linops(op_np, block), "scholqr3": lambda: orc.scholqr3_linops(op_np, block),
           "scholqr3_basic": lambda: orc.scholqr3_linops(op_np, basic=True)}[alg]()
    assert out["rc"] == ref["rc"] == 0
    R, Q = np.triu(d.cm_to_numpy(out["R"])), d.cm_to_numpy(out["Q"])
    fact, orth = _verify_qr(A, Q, R)
    assert fact <= TOL and orth <= TOL
    # the Cholesky factor is unique: entrywise agreement with the oracle up to cond^2 * eps
    np.testing.assert_allclose(R, ref["R"], rtol=0, atol=1e-10 * np.abs(ref["R"]).max())
    np.testing.assert_allclose(Q, ref["Q"], rtol=0, atol=1e-9)
    # the Q-less mode (the classes' default; sparse operators then run with row-major intermediates) returns the same R
    R2 = np.triu(d.cm_to_numpy(d.drv_qr_linops(ctx, alg, op, block_size=block)["R"]))
    np.testing.assert_allclose(R2, ref["R"], rtol=0, atol=1e-10 * np.abs(ref["R"]).max())


@pytest.mark.parametrize("kind", KINDS + ["sparse*sparse"])
@pytest.mark.parametrize("block,dense_sketch", [(0, False), (7, False), (0, True)])
def test_cqrrt_linops_vs_oracle_shared_sketch(ctx, orc, kind, block, dense_sketch):
    d = _d()
    op, op_np, A = _ops(kind, 21)
    m, n = A.shape
    out = d.drv_qr_linops(ctx, "cqrrt", op, block_size=block, want_Q=True, d_factor=2.0, use_dense_sketch=dense_sketch, key=(1, 0),
                          want_sketch=True)
    A_hat = d.cm_to_numpy(out["sketch"])
    assert A_hat.shape == (2 * n, n)
    ref = orc.cqrrt_linops(op_np, A_hat, block)
    assert out["rc"] == ref["rc"] == 0
    R, Q = np.triu(d.cm_to_numpy(out["R"])), d.cm_to_numpy(out["Q"])
    fact, orth = _verify_qr(A, Q, R)
    assert fact <= TOL and orth <= TOL
    np.testing.assert_allclose(R, ref["R"], rtol=0, atol=1e-10 * np.abs(ref["R"]).max())
    np.testing.assert_allclose(Q, ref["Q"], rtol=0, atol=1e-9)
    R2 = np.triu(d.cm_to_numpy(d.drv_qr_linops(ctx, "cqrrt", op, block_size=block, d_factor=2.0, use_dense_sketch=dense_sketch, key=(1, 0))["R"]))
    np.testing.assert_allclose(R2, ref["R"], rtol=0, atol=1e-10 * np.abs(ref["R"]).max())      # Q-less mode
    # the sketch of an operator equals the sketch of the matrix it represents, and the state advances as in CQRRT (same SkOp)
    if not dense_sketch:
        o2 = d.drv_cqrrt(ctx, d.cm_from_numpy(A), m, n, d_factor=2.0, nnz=2, key=(1, 0), want_sketch=True)
        np.testing.assert_allclose(A_hat, d.cm_to_numpy(o2["sketch"]), rtol=0, atol=1e-12 * np.abs(A_hat).max())
        assert out["next_ctr"] == o2["next_ctr"]
    else:
        S = orc.fill_dense(2 * n, m, key=(1, 0))
        S = S[0] if isinstance(S, tuple) else S
        np.testing.assert_allclose(A_hat, S @ A, rtol=0, atol=1e-11 * np.abs(A_hat).max())


def test_qr_linops_f32_and_q_less(ctx, orc):
    d = _d()
    import torch

    rng = np.random.default_rng(4)
    A = rng.standard_normal((100, 50)).astype(np.float32)
    op = d.DenseOperator(d.cm_from_numpy(A), 100, 50)
    tol32 = float(np.finfo(np.float32).eps) ** 0.75
    for alg in ("cholqr", "scholqr3", "scholqr3_basic", "cqrrt"):
        out = d.drv_qr_linops(ctx, alg, op, want_Q=True, d_factor=2.0, key=(1, 0))
        assert out["rc"] == 0 and out["R"].dtype == torch.float32
        fact, orth = _verify_qr(A.astype(np.float64), d.cm_to_numpy(out["Q"]).astype(np.float64), np.triu(d.cm_to_numpy(out["R"])).astype(np.float64))
        assert fact <= tol32 and orth <= tol32, (alg, fact, orth)
        # Q-less call (the default mode of the classes): same R, no Q
        out2 = d.drv_qr_linops(ctx, alg, op, want_Q=False, d_factor=2.0, key=(1, 0))
        assert "Q" not in out2
        np.testing.assert_allclose(d.cm_to_numpy(out2["R"]), d.cm_to_numpy(out["R"]), rtol=0, atol=1e-5 * np.abs(A).max() * 10)


def test_qr_linops_failure_codes_and_bad_args(ctx, orc):
    d = _d()
    from randlapack_amd._lib import RlhipError

    A = np.ones((60, 4))
    A[:, 2] = 0                                           # an exactly zero pivot in the Gram matrix
    op = d.DenseOperator(d.cm_from_numpy(A), 60, 4)
    assert d.drv_qr_linops(ctx, "cholqr", op)["rc"] == 1 == orc.cholqr_linops(A)["rc"]
    Z = np.zeros((60, 4))
    opz = d.DenseOperator(d.cm_from_numpy(Z), 60, 4)
    assert d.drv_qr_linops(ctx, "cqrrt", opz, d_factor=2.0)["rc"] == 1     # zero diagonal in the sketch's R (rl_cqrrt_linops.hh:231)
    assert d.drv_qr_linops(ctx, "scholqr3", opz)["rc"] == 1
    with pytest.raises(RlhipError):
        d.drv_qr_linops(ctx, 9, op)
    L = d.DenseOperator(d.cm_from_numpy(np.ones((10, 5))), 10, 5)
    Rr = d.DenseOperator(d.cm_from_numpy(np.ones((6, 3))), 6, 3)
    with pytest.raises(RlhipError, match="must match"):
        d.drv_qr_linops(ctx, "cholqr", (L, Rr))


def test_abrik_on_sparse_operator_matches_dense(ctx, orc):
    d = _d()
    S = (_sparse(600, 200, 0.05, 8) @ sp.diags(0.85 ** np.arange(200))).tocsr()     # still sparse, decaying spectrum
    A = S.toarray()
    k, iters = 8, 8
    o_s = d.drv_abrik_linop(ctx, d.CsrOperator.from_scipy(S), k, 1e-12, max_krylov_iters=iters, key=(3, 0))
    o_d = d.drv_abrik(ctx, d.cm_from_numpy(A), 600, 200, k, 1e-12, max_krylov_iters=iters, key=(3, 0))
    assert o_s["triplets"] == o_d["triplets"] and o_s["iters"] == o_d["iters"]
    np.testing.assert_allclose(o_s["S"].cpu().numpy(), o_d["S"].cpu().numpy(), rtol=1e-10)
    sv = np.linalg.svd(A, compute_uv=False)
    t = o_s["triplets"]
    U, V, Sg = d.cm_to_numpy(o_s["U"]), d.cm_to_numpy(o_s["V"]), o_s["S"].cpu().numpy()
    assert np.linalg.norm(U.T @ U - np.eye(t)) < 1e-10 and np.linalg.norm(V.T @ V - np.eye(t)) < 1e-10
    # residual metric of the reference's ABRIK test (test_abrik.cc:60-96): sqrt(||AV - US||^2 + ||A^T U - VS||^2) / sigma_1 on the leading triplets
    lead = 4
    res = np.sqrt(np.linalg.norm(A @ V[:, :lead] - U[:, :lead] * Sg[:lead])**2 + np.linalg.norm(A.T @ U[:, :lead] - V[:, :lead] * Sg[:lead])**2)
    assert res / sv[0] < 1e-3


def test_linops_degenerate_shapes(ctx, orc):
    d = _d()
    import torch

    # an all-zero sparse operator (nnz = 0): products are beta * C, CholQR reports the breakdown, the transpose builds
    Z = sp.csr_matrix((50, 7))
    opz = d.CsrOperator.from_scipy(Z)
    B = np.arange(21.0).reshape(7, 3)
    C0 = np.ones((50, 3))
    got = d.linop_apply(ctx, opz, "L", "N", d.cm_from_numpy(B), 50, 3, 7, alpha=2.0, beta=-3.0, C_in=d.cm_from_numpy(C0))
    np.testing.assert_array_equal(d.cm_to_numpy(got), -3.0 * C0)
    assert d.drv_qr_linops(ctx, "cholqr", opz)["rc"] == 1
    # one column, one row per nonzero; n = 1 through every driver
    S1 = sp.csr_matrix(np.arange(1.0, 41.0).reshape(40, 1))
    op1 = d.CsrOperator.from_scipy(S1)
    for alg in ("cholqr", "scholqr3", "scholqr3_basic", "cqrrt"):
        out = d.drv_qr_linops(ctx, alg, op1, want_Q=True, d_factor=3.0, nnz=1)
        R, Q = d.cm_to_numpy(out["R"]), d.cm_to_numpy(out["Q"])
        assert out["rc"] == 0 and abs(abs(R[0, 0]) - np.linalg.norm(S1.toarray())) < 1e-12 * np.linalg.norm(S1.toarray())
        np.testing.assert_allclose(Q @ R, S1.toarray(), rtol=0, atol=1e-12 * 40)
    # a single dense column block wider than the operator is tall (m < n): the Gram matrix is singular -> breakdown code, no crash
    W = np.random.default_rng(0).standard_normal((5, 9))
    assert d.drv_qr_linops(ctx, "cholqr", d.DenseOperator(d.cm_from_numpy(W), 5, 9))["rc"] == 1
    # SpMM with a single dense column and with 257 (crosses the 256-column grid chunk)
    S = _sparse(300, 40, 0.1, 2)
    op = d.CsrOperator.from_scipy(S)
    for nc in (1, 257):
        X = np.random.default_rng(nc).standard_normal((40, nc))
        got = d.linop_apply(ctx, op, "L", "N", d.cm_from_numpy(X), 300, nc, 40)
        np.testing.assert_allclose(d.cm_to_numpy(got), S @ X, rtol=0, atol=1e-12)


def test_cqrrt_linops_at_scale_sparse_vs_dense_cqrrt(ctx):
    """bench_CQRRT_linops scale (10^6 x 512 sparse, 8 nonzeros per row): R from the operator route equals R from the dense CQRRT
    driver on the materialised matrix (same sketching operator, so the same sketch), and reproduces A^T A."""
    d = _d()
    import torch

    m, n, r = 1_000_000, 512, 8
    rng = np.random.default_rng(1)
    cols = np.sort(rng.integers(0, n, size=(m, r)), axis=1).astype(np.int64).ravel()
    vals = rng.standard_normal(m * r)
    op = d.CsrOperator(m, n, torch.as_tensor(np.arange(m + 1, dtype=np.int64) * r, device="cuda:0"), torch.as_tensor(cols, device="cuda:0"),
                       torch.as_tensor(vals, device="cuda:0"))
    out = d.drv_qr_linops(ctx, "cqrrt", op, d_factor=2.0, nnz=4, key=(1, 0))
    assert out["rc"] == 0
    R = torch.triu(out["R"].T)                                       # (n, n) upper triangular, row-major view of the column-major buffer
    Ad = torch.zeros((n, m), dtype=torch.float64, device="cuda:0")   # column-major m x n
    Ad.index_put_((torch.as_tensor(cols, device="cuda:0"), torch.arange(m, device="cuda:0").repeat_interleave(r)), op.vals, accumulate=True)
    G = Ad @ Ad.T                                                    # A^T A
    err = torch.linalg.norm(R.T @ R - G) / torch.linalg.norm(G)
    assert float(err) < 1e-13
    o2 = d.drv_cqrrt(ctx, Ad, m, n, d_factor=2.0, nnz=4, key=(1, 0))
    R2 = torch.triu(o2["R"].T)
    assert float(torch.linalg.norm(R - R2) / torch.linalg.norm(R2)) < 1e-11
    assert out["next_ctr"] == o2["next_ctr"]


# ------------------------------------------------------------------------------------------------ block views, CSC / COO storage, A + mu I
VIEWS = [("row_block", "mid"), ("row_block", "first"), ("row_block", "last"), ("col_block", "mid"), ("col_block", "first"), ("col_block", "last"),
         ("submatrix", "mid"), ("submatrix", "corner"), ("submatrix", "end")]


def _view_of(how, where, rows, cols):
    """the reference's cases (test/linops/test_linop_block_views.cc: *_middle, *_first, *_last, submatrix_corner / _end) scaled to the operator"""
    r = {"mid": (rows // 4, rows // 2), "first": (0, rows // 3), "last": (rows - rows // 3, rows // 3), "corner": (0, rows // 2), "end": (rows - rows // 2, rows // 2)}[where]
    c = {"mid": (cols // 4, cols // 2), "first": (0, cols // 3), "last": (cols - cols // 3, cols // 3), "corner": (0, cols // 2), "end": (cols - cols // 2, cols // 2)}[where]
    if how == "row_block":
        return (r[0], 0, r[1], cols), r[1], cols
    if how == "col_block":
        return (0, c[0], rows, c[1]), rows, c[1]
    return (r[0], c[0], r[1], c[1]), r[1], c[1]


def _dense_of(A):
    if isinstance(A, tuple):
        return _dense_of(A[0]) @ _dense_of(A[1])
    return A.toarray() if sp.issparse(A) else np.asarray(A)


@pytest.mark.parametrize("how,where", VIEWS)
@pytest.mark.parametrize("kind", ["dense", "sparse", "csc", "dense*sparse", "sparse*dense", "sparse*sparse"])
def test_linop_block_views_match_the_oracle(ctx, orc, kind, how, where):
    """row_block / col_block / submatrix of every operator type (rl_dense_linop.hh:295-330, rl_sparse_linop.hh:393-465,
    rl_composite_linop.hh:505-530): the view applied to a block of vectors, both directions, against the numpy restatement of the
    same view (oracle.linop_view) -- the reference's test_linop_block_views.cc cases at sizes where a device kernel is exercised."""
    d = _d()
    if kind == "csc":
        S = _sparse(400, 50, 0.2, 77)
        op, op_np = d.CscOperator.from_scipy(S), S
    else:
        op, op_np, _ = _ops(kind, 23)
    rows, cols = orc._op_shape(op_np)
    view, vr, vc = _view_of(how, where, rows, cols)
    V = _dense_of(orc.linop_view(op_np, how, view))
    assert V.shape == (vr, vc)
    rng = np.random.default_rng(vr + 3 * vc)
    for nb in (5, 40):
        X = rng.standard_normal((vc, nb)); Z = rng.standard_normal((vr, nb)); C0 = rng.standard_normal((vr, nb))
        Y = d.cm_to_numpy(d.linop_apply_view(ctx, op, how, view, "L", "N", d.cm_from_numpy(X), vr, nb, vc, alpha=1.5, beta=-0.5, C_in=d.cm_from_numpy(C0)))
        ref = 1.5 * V @ X - 0.5 * C0
        np.testing.assert_allclose(Y, ref, rtol=0, atol=1e-13 * max(1.0, np.abs(ref).max()) * 10)
        Yt = d.cm_to_numpy(d.linop_apply_view(ctx, op, how, view, "L", "T", d.cm_from_numpy(Z), vc, nb, vr))
        np.testing.assert_allclose(Yt, V.T @ Z, rtol=0, atol=1e-13 * max(1.0, np.abs(V.T @ Z).max()) * 10)
    if not isinstance(op, tuple):                                      # Side::Right on plain operators: C = B * view
        Bm = rng.standard_normal((7, vr))
        Yr = d.cm_to_numpy(d.linop_apply_view(ctx, op, how, view, "R", "N", d.cm_from_numpy(Bm), 7, vc, vr))
        np.testing.assert_allclose(Yr, Bm @ V, rtol=0, atol=1e-13 * max(1.0, np.abs(Bm @ V).max()) * 10)


def test_linop_block_view_argument_errors(ctx):
    """the randlapack_require checks of the block methods (negative starts, empty counts, ranges past the end)"""
    d = _d()
    op, _, _ = _ops("sparse", 5)
    dop, _, _ = _ops("dense", 5)
    X = d.cm_zeros(50, 3)
    for o in (op, dop):
        for how, view in (("row_block", (-1, 0, 10, 50)), ("row_block", (395, 0, 10, 50)), ("col_block", (0, 45, 400, 10)), ("submatrix", (0, 0, 0, 5)),
                          ("submatrix", (10, 10, 5, 0))):
            with pytest.raises(Exception):
                d.linop_apply_view(ctx, o, how, view, "L", "N", X, view[2], 3, view[3])


@pytest.mark.parametrize("storage", ["csc", "coo", "coo-duplicates"])
def test_sparse_linop_from_csc_and_coo_storage(ctx, orc, storage):
    """SparseLinOp over the other two RandBLAS sparse formats (CSCMatrix, COOMatrix; rl_sparse_linop.hh:41-113): products in both
    directions, the Frobenius norm through a CholQR_linops call (its first use), and CholQR_linops' R against the oracle's."""
    d = _d()
    rng = np.random.default_rng(11)
    S = _sparse(600, 40, 0.15, 9)
    if storage == "csc":
        op, S_np = d.CscOperator.from_scipy(S), S
    else:
        coo = S.tocoo()
        ri, ci, v = coo.row.copy(), coo.col.copy(), coo.data.copy()
        if storage == "coo-duplicates":                                 # every 7th entry split in two: the products sum duplicates
            idx = np.arange(0, len(v), 7)
            ri = np.concatenate([ri, ri[idx]]); ci = np.concatenate([ci, ci[idx]]); v = np.concatenate([v, 0.25 * v[idx]])
            v[idx] *= 0.75
        perm = rng.permutation(len(v))                                  # any order
        op, S_np = d.CooOperator.from_triplets(600, 40, ri[perm], ci[perm], v[perm]), S
    X = rng.standard_normal((40, 9)); Z = rng.standard_normal((600, 9))
    Y = d.cm_to_numpy(d.linop_apply(ctx, op, "L", "N", d.cm_from_numpy(X), 600, 9, 40))
    np.testing.assert_allclose(Y, S_np @ X, rtol=0, atol=1e-13 * np.abs(S_np @ X).max() * 10)
    Yt = d.cm_to_numpy(d.linop_apply(ctx, op, "L", "T", d.cm_from_numpy(Z), 40, 9, 600))
    np.testing.assert_allclose(Yt, S_np.T @ Z, rtol=0, atol=1e-13 * np.abs(S_np.T @ Z).max() * 10)
    r = d.drv_qr_linops(ctx, "cholqr", op, want_Q=True)
    o = orc.cholqr_linops(S_np)
    assert r["rc"] == o["rc"] == 0
    np.testing.assert_allclose(d.cm_to_numpy(r["R"]), o["R"], rtol=0, atol=1e-12 * np.abs(o["R"]).max())


@pytest.mark.parametrize("num_ops,eir", [(1, True), (1, False), (6, True), (0, True)])
def test_reg_explicit_sym_linop_vs_oracle(ctx, orc, num_ops, eir):
    """linops::RegExplicitSymLinOp (rl_sym_linops.hh:134-233): A + mu_i I applied to a block, one mu or one per column; the strictly
    lower triangle of the buffer holds NaNs (it must never be read); lda > dim."""
    d = _d()
    import torch

    rng = np.random.default_rng(4 + num_ops)
    dim, lda, n = 300, 320, 6
    G0 = rng.standard_normal((dim, dim)); G = G0 @ G0.T / dim
    buf = np.full((lda, dim), np.nan); buf[:dim] = np.triu(G) + np.tril(np.full((dim, dim), np.nan), -1)
    regs = list(rng.random(num_ops) + 0.1)
    B = rng.standard_normal((dim, n)); C0 = rng.standard_normal((dim, n))
    Ad = d.cm_from_numpy(buf)
    out = d.cm_to_numpy(d.regsym_apply(ctx, Ad, dim, regs, eir, d.cm_from_numpy(B), n, alpha=0.7, beta=0.3, C_in=d.cm_from_numpy(C0), lda=lda))
    ref = orc.regsym_apply(np.triu(G), regs, eir, B, alpha=0.7, beta=0.3, C=C0)
    assert np.isfinite(out).all()
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-13 * np.abs(ref).max() * 10)
    if num_ops == 6:                                                    # n != num_ops is rejected (rl_sym_linops.hh:209)
        with pytest.raises(Exception):
            d.regsym_apply(ctx, Ad, dim, regs, True, d.cm_from_numpy(B[:, :4]), 4, lda=lda)
