"""-m gpu: each HIP kernel family through the C ABI against the CPU oracle / numpy on the same inputs.
Tolerances are stated per test; integer work (Philox, pivots) is bit-exact."""
import json
import os
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
EPS = np.finfo(np.float64).eps


def _dev():
    from randlapack_amd import device

    return device


def relerr(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)


def test_library_is_the_hip_one(ctx):
    assert ctx.lib.rlhip_version().decode().startswith("rlhip")
    assert ctx.mfma_peak(True, 2000) > 10.0  # TFLOP/s: the MFMA pipe is really being driven


def test_philox_kat_device(ctx):
    for v in json.loads((G / "philox_kat.json").read_text()):
        out = ctx.philox(1, v["ctr"], v["key"])
        assert [int(x) for x in out] == v["out"]          # bit-exact


def test_philox_counter_blocks_match_oracle(ctx, orc):
    out = ctx.philox(5, (0xFFFFFFFD, 7, 0, 0), (11, 13)).reshape(5, 4)
    for b in range(5):
        lo = 0xFFFFFFFD + b
        ref = orc.philox((lo & 0xFFFFFFFF, 7 + (lo >> 32), 0, 0), (11, 13))
        assert list(out[b]) == list(ref)


@pytest.mark.parametrize("rows,cols,dist", [(1000, 300, 0), (17, 5, 0), (3, 1, 1), (257, 33, 1)])
def test_fill_dense_matches_oracle(ctx, orc, rows, cols, dist):
    d = _dev()
    buf = d.cm_empty(rows, cols)
    nxt = ctx.fill_dense(buf, rows, cols, ctr=(5, 0, 0, 0), key=(9, 1), dist=dist)
    ref, nxt_ref = orc.fill_dense(rows, cols, ctr=(5, 0, 0, 0), key=(9, 1), dist=dist)
    assert nxt == nxt_ref                                  # state threading is integer-exact
    # the float stage uses device libm vs glibc: a few ulp at most
    np.testing.assert_allclose(d.cm_to_numpy(buf), ref, rtol=0, atol=4e-15)


def test_fill_dense_f32_is_rounded_f64(ctx, orc):
    import torch

    d = _dev()
    buf = d.cm_empty(64, 8, dtype=torch.float32)
    ctx.fill_dense(buf, 64, 8)
    ref, _ = orc.fill_dense(64, 8)
    np.testing.assert_allclose(d.cm_to_numpy(buf), ref.astype(np.float32), rtol=0, atol=2e-7)


@pytest.mark.parametrize("m,n,k", [(300, 70, 45), (257, 256, 130), (128, 256, 64), (1000, 17, 33), (64, 300, 1000),
                                   (5, 3, 2), (513, 129, 4000), (1, 1, 1), (130, 20, 7)])
@pytest.mark.parametrize("ta", "NT")
@pytest.mark.parametrize("tb", "NT")
def test_gemm_f64(ctx, m, n, k, ta, tb):
    d = _dev()
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    A, B, C0 = rng.standard_normal((m, k)), rng.standard_normal((k, n)), rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A if ta == "N" else A.T.copy())
    Bd = d.cm_from_numpy(B if tb == "N" else B.T.copy())
    Cd = d.cm_from_numpy(C0)
    ctx.gemm(ta, tb, m, n, k, 1.5, Ad, m if ta == "N" else k, Bd, k if tb == "N" else n, -0.5, Cd, m)
    ref = 1.5 * A @ B - 0.5 * C0
    assert relerr(d.cm_to_numpy(Cd), ref) < 50 * EPS * np.sqrt(k)


def test_gemm_f32(ctx):
    import torch

    d = _dev()
    rng = np.random.default_rng(0)
    m, n, k = 300, 200, 500
    A, B = rng.standard_normal((m, k)).astype(np.float32), rng.standard_normal((k, n)).astype(np.float32)
    Cd = d.cm_zeros(m, n, dtype=torch.float32)
    ctx.gemm("N", "N", m, n, k, 1.0, d.cm_from_numpy(A), m, d.cm_from_numpy(B), k, 0.0, Cd, m)
    assert relerr(d.cm_to_numpy(Cd), A.astype(np.float64) @ B.astype(np.float64)) < 1e-5


def test_gemm_submatrix_ld_and_beta_zero_ignores_nan(ctx):
    d = _dev()
    rng = np.random.default_rng(1)
    A = rng.standard_normal((50, 40))
    B = rng.standard_normal((40, 30))
    Cfull = np.full((60, 30), np.nan)
    Ad, Bd, Cd = d.cm_from_numpy(A), d.cm_from_numpy(B), d.cm_from_numpy(Cfull)
    # use the top-left 33 x 21 x 17 sub-problem with the parents' leading dimensions
    ctx.gemm("N", "N", 33, 21, 17, 1.0, Ad, 50, Bd, 40, 0.0, Cd, 60)
    out = d.cm_to_numpy(Cd)
    np.testing.assert_allclose(out[:33, :21], A[:33, :17] @ B[:17, :21], atol=1e-12)
    assert np.isnan(out[33:, :]).all() and np.isnan(out[:33, 21:]).all()      # nothing else touched


def test_gemm_is_deterministic_run_to_run(ctx):
    d = _dev()
    rng = np.random.default_rng(2)
    m, n, k = 100, 256, 60000                         # forces the split-K path
    A, B = rng.standard_normal((k, m)), rng.standard_normal((k, n))
    Ad, Bd = d.cm_from_numpy(A), d.cm_from_numpy(B)
    outs = []
    for _ in range(3):
        Cd = d.cm_zeros(m, n)
        ctx.gemm("T", "N", m, n, k, 1.0, Ad, k, Bd, k, 0.0, Cd, m)
        outs.append(d.cm_to_numpy(Cd))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])   # bitwise
    assert relerr(outs[0], A.T @ B) < 1e-12


@pytest.mark.parametrize("m,n,k,dt", [(32, 32, 20000, "f64"), (17, 29, 8192, "f64"), (64, 32, 33001, "f64"), (5, 64, 9000, "f64"), (64, 64, 16389, "f64"),
                                      (32, 48, 40000, "f32"), (1, 1, 8192, "f64")])
def test_gemm_tn_tall_skinny_panels(ctx, m, n, k, dt):
    """C (m x n, both <= 64) = alpha A^T B + beta C over >= 8192 rows: the narrow-panel kernel (gemm.hip::gemm_tn_skinny_kernel -- ABRIK's Krylov
    blocks, CQRRT / Cholesky-QR on a few dozen columns) with sub-matrix leading dimensions, a ragged last slab, padded column tiles, alpha / beta,
    and bitwise reproducibility."""
    import torch

    d = _dev()
    rng = np.random.default_rng(m * 131 + n * 7 + k)
    npdt, tdt, tol = (np.float64, torch.float64, 1e-13) if dt == "f64" else (np.float32, torch.float32, 2e-5)
    lda, ldb, ldc = k + 3, k + 8, m + 2
    A = rng.standard_normal((lda, m)).astype(npdt); B = rng.standard_normal((ldb, n)).astype(npdt); C0 = rng.standard_normal((ldc, n)).astype(npdt)
    Ad, Bd = d.cm_from_numpy(A), d.cm_from_numpy(B)
    outs = []
    for _ in range(2):
        Cd = d.cm_from_numpy(C0)
        ctx.gemm("T", "N", m, n, k, 1.5, Ad, lda, Bd, ldb, -0.5, Cd, ldc)
        outs.append(d.cm_to_numpy(Cd))
    ref = C0.astype(np.float64).copy()
    ref[:m] = 1.5 * A[:k].astype(np.float64).T @ B[:k].astype(np.float64) - 0.5 * C0[:m]
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0] - ref).max() <= tol * np.sqrt(k) * max(1.0, np.abs(ref).max())
    assert np.array_equal(outs[0][m:], C0[m:])                                    # rows below the product are not touched


@pytest.mark.parametrize("n,k", [(32, 9000), (64, 20000), (24, 8192)])
def test_gemm_tn_tall_skinny_same_operand_full_result(ctx, n, k):
    """A^T A asked for as a GEMM (both operands the same matrix, no triangle flag -- the orthogonality checks of the drivers do that): the
    narrow-panel kernel shares the operand fragments but must still deliver the WHOLE n x n result, lower triangle included"""
    d = _dev()
    rng = np.random.default_rng(n * 3 + k)
    A = rng.standard_normal((k, n))
    Ad = d.cm_from_numpy(A)
    Cd = d.cm_from_numpy(np.full((n, n), np.nan))
    ctx.gemm("T", "N", n, n, k, 1.0, Ad, k, Ad, k, 0.0, Cd, n)
    assert relerr(d.cm_to_numpy(Cd), A.T @ A) < 1e-13


@pytest.mark.parametrize("n,k", [(32, 200000), (20, 9000), (64, 30000), (48, 8200)])
def test_syrk_upper_tall_narrow(ctx, n, k):
    """the same kernel as a Gram matrix (B is A): upper triangle only, the strictly lower triangle of C is not touched"""
    d = _dev()
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((k, n))
    C0 = rng.standard_normal((n, n))
    Cd = d.cm_from_numpy(C0)
    ctx.syrk("U", "T", n, k, 1.0, d.cm_from_numpy(A), k, 0.0, Cd, n)
    got = d.cm_to_numpy(Cd)
    assert relerr(np.triu(got), np.triu(A.T @ A)) < 1e-13
    assert np.array_equal(np.tril(got, -1), np.tril(C0, -1))


@pytest.mark.parametrize("n,k", [(256, 5000), (100, 300), (300, 2000), (1, 10), (130, 17)])
def test_syrk_upper(ctx, n, k):
    d = _dev()
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((k, n))
    Cd = d.cm_zeros(n, n)
    ctx.syrk("U", "T", n, k, 1.0, d.cm_from_numpy(A), k, 0.0, Cd, n)
    assert relerr(np.triu(d.cm_to_numpy(Cd)), np.triu(A.T @ A)) < 1e-13


@pytest.mark.parametrize("n", [1, 5, 32, 33, 100, 256, 449, 700, 1024, 1100, 2048])
def test_potrf_upper(ctx, n):
    d = _dev()
    rng = np.random.default_rng(n)
    X = rng.standard_normal((2 * n + 3, n))
    Gm = X.T @ X
    Gd = d.cm_from_numpy(Gm)
    assert ctx.potrf(n, Gd, n) == 0
    got = d.cm_to_numpy(Gd)
    R = np.triu(got)
    assert relerr(R.T @ R, Gm) < 1e-13
    assert (np.diag(R) > 0).all()
    assert np.array_equal(np.tril(got, -1), np.tril(Gm, -1))               # LAPACK's uplo contract: the strictly lower triangle is not touched


def test_potrf_reports_first_bad_minor(ctx, orc):
    d = _dev()
    Gm = np.eye(40)
    Gm[17, 17] = -1.0
    assert ctx.potrf(40, d.cm_from_numpy(Gm), 40) == 18                    # LAPACK info semantics
    Gm = np.ones((6, 6))                                                     # rank 1 -> fails at minor 2
    assert ctx.potrf(6, d.cm_from_numpy(Gm), 6) == 2
    for bad in (3, 300, 700):                                                # two-level path: first, second and last 256-block
        Gm = np.eye(700)
        Gm[bad - 1, bad - 1] = -2.0
        assert ctx.potrf(700, d.cm_from_numpy(Gm), 700) == bad
    rng = np.random.default_rng(5)                                           # numerically rank-deficient Gram matrix: same info as LAPACK
    X = rng.standard_normal((900, 300))
    Gm = np.zeros((600, 600))
    Gm[:300, :300] = X.T @ X
    Gm[300:, 300:] = Gm[:300, :300]
    Gm[300, 300] = 0.0
    assert ctx.potrf(600, d.cm_from_numpy(Gm), 600) == 301


@pytest.mark.parametrize("n,bad", [(300, 125), (300, 97), (1000, 505), (40, 1)])
def test_potrf_failure_leaves_the_leading_factor(ctx, n, bad):
    """On a non-positive pivot the rows of U above it are final, as dpotrf leaves them (benchmark/bench_general/Chol_check.cc factors an
    indefinite matrix on purpose and reads the leading block of R): every row above the failing pivot's 32-row panel row agrees with the
    Cholesky factor of the leading minor -- through the columns that panel reaches -- and so do the rows of its own diagonal block."""
    d = _dev()
    rng = np.random.default_rng(n + bad)
    X = rng.standard_normal((2 * n, n))
    G = X.T @ X
    G[bad - 1, bad - 1] = -1.0                                               # minors 1 .. bad - 1 positive definite, minor `bad` not
    Gd = d.cm_from_numpy(G)
    assert ctx.potrf(n, Gd, n) == bad
    got = np.triu(d.cm_to_numpy(Gd))
    k = bad - 1
    if k:
        U = np.linalg.cholesky(G[:k, :k]).T
        assert np.abs(got[:k, :k] - U).max() <= 1e-12 * np.abs(U).max()
        assert np.abs(got[:k, :k].T @ got[:k, :k] - G[:k, :k]).max() <= 1e-12 * np.abs(G).max()
    assert got[k, k] <= 0.0                                                  # the pivot entry holds its updated, non-positive value


@pytest.mark.parametrize("m,n", [(1000, 256), (333, 100), (2000, 600), (70, 5), (1, 1), (513, 257)])
def test_trsm_right_upper_matches_lapack(ctx, orc, m, n):
    import ctypes as C

    d = _dev()
    rng = np.random.default_rng(m + n)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
    B = rng.standard_normal((m, n))
    Bd = d.cm_from_numpy(B)
    ctx.trsm(m, n, 2.0, d.cm_from_numpy(U), n, Bd, m)
    X = d.cm_to_numpy(Bd)
    # LAPACK's own trsm on the host as the reference
    Bh = np.asfortranarray(B.copy())
    Uh = np.asfortranarray(U)
    orc.load().oracle_trsm_right_upper_f64(C.c_int64(m), C.c_int64(n), C.c_double(2.0), Uh.ctypes.data_as(C.c_void_p),
                                           C.c_int64(n), Bh.ctypes.data_as(C.c_void_p), C.c_int64(m))
    assert relerr(X, Bh) < 1e-11
    assert relerr(X @ U, 2.0 * B) < 1e-12                                  # backward-stable residual


def test_trsm_is_substitution_not_inverse(ctx):
    # graded R with cond ~1e12 (CQRRPT's preconditioning regime): residual must stay at eps*||B||
    d = _dev()
    rng = np.random.default_rng(3)
    m, n = 400, 64
    U = np.triu(rng.standard_normal((n, n))) * np.logspace(0, -12, n)[:, None] + np.diag(np.logspace(0, -12, n))
    B = rng.standard_normal((m, n)) @ U
    Bd = d.cm_from_numpy(B)
    ctx.trsm(m, n, 1.0, d.cm_from_numpy(U), n, Bd, m)
    X = d.cm_to_numpy(Bd)
    assert np.linalg.norm(X @ U - B) <= 1e-13 * np.linalg.norm(B) * n


@pytest.mark.parametrize("m,n", [(1000, 256), (40, 300), (7, 7)])
def test_trmm_right_upper(ctx, m, n):
    d = _dev()
    rng = np.random.default_rng(m * n)
    U = rng.standard_normal((n, n))       # strictly-lower garbage must be ignored
    B = rng.standard_normal((m, n))
    Bd = d.cm_from_numpy(B)
    ctx.trmm(m, n, 0.5, d.cm_from_numpy(U), n, Bd, m)
    assert relerr(d.cm_to_numpy(Bd), 0.5 * B @ np.triu(U)) < 1e-13


def test_lange_lacpy_laset_transpose(ctx):
    d = _dev()
    rng = np.random.default_rng(4)
    A = rng.standard_normal((1234, 77))
    Ad = d.cm_from_numpy(A)
    assert abs(ctx.lange_fro(1234, 77, Ad, 1234) - np.linalg.norm(A)) < 1e-10
    assert ctx.lange_fro(0, 5, Ad, 1) == 0.0
    Bd = d.cm_zeros(1234, 77)
    ctx.lacpy("U", 1234, 77, Ad, 1234, Bd, 1234)
    np.testing.assert_array_equal(d.cm_to_numpy(Bd), np.triu(A))
    ctx.laset("L", 1234, 77, 0.0, 1.0, Ad, 1234)      # LAPACK: strictly lower <- 0, diagonal <- 1
    ref = np.triu(A, 1) + np.eye(1234, 77)
    np.testing.assert_array_equal(d.cm_to_numpy(Ad), ref)
    import ctypes as C
    T = d.cm_zeros(77, 1234)
    ctx.lib.rlhip_transpose_f64(ctx.h, 1234, 77, Ad.data_ptr(), 1234, T.data_ptr(), 77, 0)
    np.testing.assert_array_equal(d.cm_to_numpy(T), ref.T)


@pytest.mark.parametrize("m,n,cond", [(300, 64, 1e8), (256, 256, 1e3), (50, 7, 10.0), (65, 33, 1e6), (512, 512, 1e3), (400, 333, 1e6), (513, 40, 1e2)])
def test_gesvdj(ctx, m, n, cond):
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n)) @ np.diag(np.logspace(0, -np.log10(cond), n)) @ np.linalg.qr(rng.standard_normal((n, n)))[0]
    Ad = d.cm_from_numpy(A)
    S = torch.empty(n, dtype=torch.float64, device="cuda")
    VT = d.cm_empty(n, n)
    info, sweeps = ctx.gesvdj(m, n, Ad, m, S, VT, n)
    assert info == 0 and sweeps > 0
    U, s, vt = d.cm_to_numpy(Ad), S.cpu().numpy(), d.cm_to_numpy(VT)
    sref = np.linalg.svd(A, compute_uv=False)
    assert np.all(np.diff(s) <= 0)
    # LAPACK's own values are only accurate to eps*sigma_max in absolute terms, so that is the yardstick
    np.testing.assert_allclose(s, sref, rtol=1e-12, atol=1e-14 * sref[0])
    assert np.abs(U * s @ vt - A).max() <= 1e-13 * np.abs(A).max() * n
    assert np.linalg.norm(U.T @ U - np.eye(n)) <= 1e-12 * n
    assert np.linalg.norm(vt @ vt.T - np.eye(n)) <= 1e-12 * n


@pytest.mark.parametrize("m,n", [(256, 256), (500, 120), (1500, 96), (3000, 256)])
def test_gesvdj_gesdd_f32(ctx, m, n):
    """fp32 SVDs run the fp64 Jacobi kernels on a widened copy (block kernel for m <= 512, per-round kernel above; gesdd: CholQR2 in fp32 +
    the widened k x k problem): singular values, reconstruction and orthogonality at the fp32 rounding level against numpy"""
    import ctypes as C
    import torch

    d = _dev()
    rng = np.random.default_rng(m * 7 + n)
    A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * np.logspace(0, -2, n) @ np.linalg.qr(rng.standard_normal((n, n)))[0]).astype(np.float32)
    sref = np.linalg.svd(A.astype(np.float64), compute_uv=False)
    e32 = float(np.finfo(np.float32).eps)
    for which in ("gesvdj", "gesdd"):
        Ad = d.cm_from_numpy(A)
        S = torch.empty(n, dtype=torch.float32, device="cuda")
        VT = d.cm_empty(n, n, dtype=torch.float32)
        sw = C.c_int()
        if which == "gesvdj":
            info = ctx.lib.rlhip_gesvdj_f32(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), VT.data_ptr(), n, C.byref(sw))
            U = Ad
        else:
            U = d.cm_empty(m, n, dtype=torch.float32)
            info = ctx.lib.rlhip_gesdd_f32(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, C.byref(sw))
        assert info == 0, which
        u, s_, vt = d.cm_to_numpy(U).astype(np.float64), S.cpu().numpy().astype(np.float64), d.cm_to_numpy(VT).astype(np.float64)
        assert np.all(np.diff(s_) <= 0)
        assert np.max(np.abs(s_ - sref)) <= 5 * e32 * sref[0], which
        assert np.linalg.norm(u * s_ @ vt - A) <= 20 * e32 * np.linalg.norm(A), which
        assert np.linalg.norm(u.T @ u - np.eye(n)) <= 10 * np.sqrt(n) * e32, which
        assert np.linalg.norm(vt @ vt.T - np.eye(n)) <= 10 * np.sqrt(n) * e32, which


@pytest.mark.parametrize("m,n,cond", [(2000, 64, 10.0), (5000, 256, 1e5), (300, 40, 1e12), (40, 40, 1e3), (3000, 512, 1e4), (576, 512, 10.0), (1000, 384, 1e9)])
def test_gesdd_tall_vs_lapack(ctx, orc, m, n, cond):
    import ctypes as C
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    A = np.linalg.qr(rng.standard_normal((m, n)))[0] * np.logspace(0, -np.log10(cond), n) @ np.linalg.qr(
        rng.standard_normal((n, n)))[0]
    Ad = d.cm_from_numpy(A)
    S = torch.empty(n, dtype=torch.float64, device="cuda")
    U, VT = d.cm_empty(m, n), d.cm_empty(n, n)
    sw = C.c_int()
    info = ctx.lib.rlhip_gesdd_f64(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n,
                                   C.byref(sw))
    assert info == 0
    u, s, vt = d.cm_to_numpy(U), S.cpu().numpy(), d.cm_to_numpy(VT)
    _, _, s_ref, _ = orc.gesdd(A)
    # same backward-error class as LAPACK's gesdd: absolute error eps * sigma_max
    assert np.max(np.abs(s - s_ref)) <= 1e-13 * s_ref[0] * np.sqrt(n)
    assert np.linalg.norm(u * s @ vt - A) <= 1e-13 * np.linalg.norm(A) * n
    assert np.linalg.norm(u.T @ u - np.eye(n)) <= 1e-11 * n


# ---------------------------------------------------------------------------------------------------
# CQRRPT building blocks: col_swap (exact KATs), geqp3 (pivots bit-exact vs LAPACK), SASO
# ---------------------------------------------------------------------------------------------------
def _col_swap_dev(ctx, A, J, k=None):
    import torch

    d = _dev()
    m, n = A.shape
    Ad = d.cm_from_numpy(A)
    Jd = torch.from_numpy(np.asarray(J, dtype=np.int64)).cuda()
    rc = ctx.lib.rlhip_col_swap_f64(ctx.h, m, n, n if k is None else k, Ad.data_ptr(), m, Jd.data_ptr())
    return rc, d.cm_to_numpy(Ad), Jd.cpu().numpy()


def test_col_swap_golden_kats_device(ctx):
    import torch

    d = _dev()
    cs = json.loads((G / "col_swap_kats.json").read_text())
    for c in cs["structured"]:                                   # test_util.cc:215-243
        A = np.array(c["A"]).reshape(c["n"], c["m"]).T
        rc, B, J = _col_swap_dev(ctx, A, c["J"])
        assert rc == 0 and list(J) == c["J"]
        np.testing.assert_array_equal(B.T.ravel(), np.array(c["expect"]))
    c = cs["lda"]                                                # test_util.cc:245-266 (lda > m, padding rows untouched)
    buf = torch.tensor(c["A"], dtype=torch.float64, device="cuda")
    Jd = torch.tensor(c["J"], dtype=torch.int64, device="cuda")
    assert ctx.lib.rlhip_col_swap_f64(ctx.h, c["m"], c["n"], c["n"], buf.data_ptr(), c["lda"], Jd.data_ptr()) == 0
    np.testing.assert_array_equal(buf.cpu().numpy(), np.array(c["expect"]))
    for c in cs["int_vector"]:                                   # test_util.cc:268-293 (prefix-only contract)
        v = torch.tensor(c["A"], dtype=torch.int64, device="cuda")
        Jd = torch.tensor(c["J"], dtype=torch.int64, device="cuda")
        assert ctx.lib.rlhip_col_swap_i64(ctx.h, c["n"], c["k"], v.data_ptr(), Jd.data_ptr()) == 0
        assert v.cpu().tolist() == c["expect"] and Jd.cpu().tolist() == c["J"]


@pytest.mark.parametrize("m,n,k,seed", [(10, 7, 7, 0), (10, 7, 4, 1), (1000, 200, 200, 2), (8, 12, 5, 3), (5, 1, 1, 5),
                                        (6, 9, 1, 6), (513, 64, 64, 7), (70, 5000, 5000, 8), (33, 4096, 2000, 9)])   # n >= 4096: host-side cycle decomposition
def test_col_swap_gather_contract_device(ctx, orc, m, n, k, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    J = rng.permutation(n) + 1
    rc, B, Jout = _col_swap_dev(ctx, A, J, k)
    assert rc == 0
    np.testing.assert_array_equal(Jout, J)
    np.testing.assert_array_equal(B, A[:, J - 1])               # bit-exact data movement
    np.testing.assert_array_equal(B, orc.col_swap(A, J, k)[1])  # == oracle == LAPACK lapmt
    assert _col_swap_dev(ctx, A, J, n + 1)[0] != 0              # k > n is an error (rl_util.hh:159-160)


@pytest.mark.parametrize("m,n,kind", [(50, 20, "scaled"), (300, 200, "scaled"), (1280, 1024, "gauss"), (200, 300, "scaled"),
                                      (640, 512, "lowrank"), (33, 33, "gauss"),
                                      (40000, 24, "scaled"), (70000, 96, "gauss")])     # tall: reflector read from its published slot, not LDS
def test_geqp3_pivots_match_lapack(ctx, orc, m, n, kind):
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    if kind == "scaled":       # well separated column norms (the bit-exact pivot variant)
        A = rng.standard_normal((m, n)) * np.logspace(0, -3, n)[rng.permutation(n)]
    elif kind == "lowrank":
        A = rng.standard_normal((m, 40)) @ rng.standard_normal((40, n))
    else:
        A = rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A)
    Jd = torch.zeros(n, dtype=torch.int64, device="cuda")
    tau = torch.zeros(min(m, n), dtype=torch.float64, device="cuda")
    assert ctx.lib.rlhip_geqp3_f64(ctx.h, m, n, Ad.data_ptr(), m, Jd.data_ptr(), tau.data_ptr()) == 0
    info, Ao, Jo, tauo = orc.geqp3(A)
    J = Jd.cpu().numpy()
    r = min(m, n)
    Rg, Ro = np.triu(d.cm_to_numpy(Ad))[:r], np.triu(Ao)[:r]
    assert sorted(J.tolist()) == list(range(1, n + 1))
    if kind == "lowrank":
        # pivots are pinned only while the remaining column norms are above rounding noise
        np.testing.assert_array_equal(J[:40], Jo[:40])
        assert np.abs(np.abs(np.diag(Rg)[:40]) - np.abs(np.diag(Ro)[:40])).max() <= 1e-12 * np.abs(Ro[0, 0])
    else:
        np.testing.assert_array_equal(J, Jo)                                     # pivot order: bit-exact
        assert np.abs(Rg - Ro).max() <= EPS**0.75 * np.abs(Ro).max()             # tau / R tolerances of
        assert np.abs(tau.cpu().numpy() - tauo).max() <= EPS**0.75              # test_bqrrp_gpu.cu:231-249
    # it is a valid QRCP regardless: A[:, J] = Q R with |R_ii| non-increasing up to noise
    dg = np.abs(np.diag(Rg))
    assert np.all(dg[1:] <= dg[:-1] * (1 + 1e-8) + 1e-12 * dg[0])


@pytest.mark.parametrize("mode", [1, 0])
def test_saso_generation_and_apply_vs_oracle(ctx, orc, mode):
    """SparseDist / SparseSkOp / sketch_general (rl_cqrrpt.hh:214-222): the device operator equals the oracle's independent
    restatement of the documented stream ENTRY BY ENTRY (mode 1 independent columns, mode 0 block affine), the state advances as
    documented, and S * A equals the dense product; ragged last row block, d > 5120 (two- and one-column LDS slabs) included."""
    import ctypes as C

    d = _dev()
    rng = np.random.default_rng(0)
    u32 = lambda v: (C.c_uint32 * len(v))(*v)
    for (dd, m, n, nnz) in [(40, 1000, 16, 4), (25, 333, 9, 2), (64, 64, 5, 8), (1280, 5000, 24, 4), (7, 50, 3, 7), (1, 5, 2, 1),
                            (6000, 13000, 6, 3), (11000, 12000, 3, 2), (48, 300, 4, 12), (4000, 9000, 5, 4)]:
        S = C.c_void_p()
        nxt = (C.c_uint32 * 4)()
        ctr, key = (0xFFFFFFF0, 3, 0, 0), (5, 9)
        assert ctx.lib.rlhip_saso_create_mode(ctx.h, dd, m, nnz, mode, u32(ctr), u32(key), nxt, C.byref(S)) == 0
        So, nxt_o = orc.saso_dense(dd, m, nnz, ctr, key, mode)
        assert tuple(nxt) == nxt_o                                                  # S.next_state: integer-exact
        Sd = d.cm_empty(dd, m)
        ctx.lib.rlhip_saso_dense_f64(ctx.h, S, Sd.data_ptr())
        Sh = d.cm_to_numpy(Sd)
        assert np.array_equal(Sh, So)                                               # generation: exact
        assert set((Sh != 0).sum(0)) == {nnz}                                       # nnz DISTINCT rows per column
        A = rng.standard_normal((m, n))
        Bd = d.cm_from_numpy(rng.standard_normal((dd, n)))
        B0 = d.cm_to_numpy(Bd)
        assert ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, 2.0, d.cm_from_numpy(A).data_ptr(), m, -1.0, Bd.data_ptr(), dd) == 0
        ref = 2.0 * So @ A - B0
        assert np.abs(d.cm_to_numpy(Bd) - ref).max() <= 1e-13 * np.abs(ref).max() * nnz * 4
        # deterministic (gather, fixed summation order): bitwise identical on a second application
        B2 = d.cm_from_numpy(B0)
        ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, 2.0, d.cm_from_numpy(A).data_ptr(), m, -1.0, B2.data_ptr(), dd)
        assert np.array_equal(d.cm_to_numpy(B2), d.cm_to_numpy(Bd))
        ctx.lib.rlhip_saso_destroy(ctx.h, S)


@pytest.mark.parametrize("mode", [1, 0])
def test_saso_embedding_distortion_on_spike_trains(ctx, orc, mode):
    """A structured adversary for an operator that works on blocks of d input rows: columns that are d-periodic spike trains
    (column c = the indicator of rows u_c, u_c + d, u_c + 2d, ...) plus columns supported inside ONE block.  With d = 8 n the
    singular values of S A / sqrt(nnz) must stay within [0.5, 2] of those of A (a Gaussian sketch gives 1 +- sqrt(n / d) = [0.65, 1.35])."""
    import ctypes as C

    d = _dev()
    rng = np.random.default_rng(4)
    n, nnz = 48, 4
    dd, m = 8 * n, 8 * n * 60
    A = np.zeros((m, n))
    offs = rng.permutation(dd)[: n // 2]
    for c, u in enumerate(offs):
        A[u::dd, c] = 1.0                                                           # d-periodic spike train
    for c in range(n // 2, n):
        t = rng.integers(0, m // dd)
        A[t * dd + rng.permutation(dd)[:40], c] = rng.standard_normal(40)           # lives inside one row block
    u32 = lambda v: (C.c_uint32 * len(v))(*v)
    S = C.c_void_p()
    nxt = (C.c_uint32 * 4)()
    assert ctx.lib.rlhip_saso_create_mode(ctx.h, dd, m, nnz, mode, u32((1, 0, 0, 0)), u32((2, 0)), nxt, C.byref(S)) == 0
    Bd = d.cm_zeros(dd, n)
    assert ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, 1.0, d.cm_from_numpy(A).data_ptr(), m, 0.0, Bd.data_ptr(), dd) == 0
    ctx.lib.rlhip_saso_destroy(ctx.h, S)
    SA = d.cm_to_numpy(Bd) / np.sqrt(nnz)
    # distortion on range(A): singular values of (S A) R^-1 with A = Q R
    R = np.linalg.qr(A, mode="r")
    sv = np.linalg.svd(SA @ np.linalg.inv(R), compute_uv=False)
    assert 0.5 <= sv[-1] and sv[0] <= 2.0, (mode, sv[0], sv[-1])


# ---------------------------------------------------------------------------------------------------
# Householder / LU building blocks of BQRRP, HQRQ, PLUL against LAPACK (scipy's OpenBLAS)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n", [(50, 20), (300, 64), (2000, 100), (16384, 512), (40, 40), (64, 100), (5000, 33)])
def test_getrf_pivots_and_factors_match_lapack(ctx, m, n):
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A)
    ip = torch.zeros(min(m, n), dtype=torch.int64, device="cuda")
    info = ctx.lib.rlhip_getrf_f64(ctx.h, m, n, Ad.data_ptr(), m, ip.data_ptr())
    ctx.sync()
    lu_ref, piv_ref, info_ref = ll.dgetrf(A)
    assert info == info_ref == 0
    np.testing.assert_array_equal(ip.cpu().numpy() - 1, piv_ref)                     # pivot rows: bit-exact
    np.testing.assert_allclose(d.cm_to_numpy(Ad), lu_ref, atol=5e-13 * np.abs(lu_ref).max(), rtol=0)
    # pivots-only variant (BQRRP's qrcp_wide reads nothing else): identical ipiv, U part still the LAPACK U
    Ad2 = d.cm_from_numpy(A)
    ip2 = torch.zeros(min(m, n), dtype=torch.int64, device="cuda")
    assert ctx.lib.rlhip_getrf_piv_f64(ctx.h, m, n, Ad2.data_ptr(), m, ip2.data_ptr()) == 0
    assert torch.equal(ip, ip2)
    k = min(m, n)
    np.testing.assert_allclose(np.triu(d.cm_to_numpy(Ad2)[:k]), np.triu(lu_ref[:k]), atol=5e-13 * np.abs(lu_ref).max(), rtol=0)


@pytest.mark.parametrize("m,n", [(70000, 96), (33000, 40), (65536, 160), (5000, 70), (1030, 33)])
def test_getrf_tall_f32_pivots_match_lapack(ctx, m, n):
    """the 4-rows-per-thread register panels (fp32, > 1024 rows): the general step with more than 64 workgroups (70000 rows) and the
    label-swapping step of lu_f32_step below that, over several panels (repeated interchanges move rows between workgroups)"""
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    rng = np.random.default_rng(m)
    A = (rng.standard_normal((m, n)) * np.logspace(0, -2, n)).astype(np.float32)
    Ad = d.cm_from_numpy(A)
    ip = torch.zeros(n, dtype=torch.int64, device="cuda")
    assert ctx.lib.rlhip_getrf_f32(ctx.h, m, n, Ad.data_ptr(), m, ip.data_ptr()) == 0
    lu_ref, piv_ref, info_ref = ll.sgetrf(A)
    assert info_ref == 0
    np.testing.assert_array_equal(ip.cpu().numpy() - 1, piv_ref)
    np.testing.assert_allclose(d.cm_to_numpy(Ad), lu_ref, atol=2e-4 * np.abs(lu_ref).max(), rtol=0)


@pytest.mark.parametrize("m,n", [(16384, 96), (33000, 40), (2000, 65), (1024, 32)])
def test_getrf_tall_f64_fast_step_pivots_match_lapack(ctx, m, n):
    """the fp64 panel step of lu_f64.hip (two-stage first-maximum reductions, rows published by the owner's wave, interchanges by label;
    up to 32768 rows; 33000 rows take the general step) against LAPACK: pivots identical, factors to rounding"""
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n)) * np.logspace(0, -3, n)
    A[5] = A[m - 7]                                                         # an exact tie between two rows of different workgroups
    Ad = d.cm_from_numpy(A)
    ip = torch.zeros(n, dtype=torch.int64, device="cuda")
    assert ctx.lib.rlhip_getrf_f64(ctx.h, m, n, Ad.data_ptr(), m, ip.data_ptr()) == 0
    lu_ref, piv_ref, info_ref = ll.dgetrf(A)
    np.testing.assert_array_equal(ip.cpu().numpy() - 1, piv_ref)
    np.testing.assert_allclose(d.cm_to_numpy(Ad), lu_ref, atol=1e-12 * np.abs(lu_ref).max(), rtol=0)


@pytest.mark.parametrize("m,n,dtype,path7", [(200000, 96, "f64", False), (300000, 40, "f64", True), (400000, 48, "f32", True), (200000, 64, "f32", False)])
def test_getrf_very_tall_panels_match_lapack(ctx, m, n, dtype, path7):
    """PLUL's shape at BASELINE configs[1] with power iterations is 200000 x 256: more rows than ONE workgroup per CU of the register
    panel kernel can hold.  200000 fp64 rows run the general register kernel on 391 co-resident workgroups (two per CU); beyond the
    resident capacity (262144 rows) the column-at-a-time panel takes over (path counter 7).  Pivots identical to LAPACK, factors to
    rounding, an exact tie between the first and the last workgroup's rows included.  (Round 3 found the 200000-row case timing out.)"""
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    npdt = np.float64 if dtype == "f64" else np.float32
    rng = np.random.default_rng(m + n)
    A = (rng.standard_normal((m, n)) * np.logspace(0, -2, n)).astype(npdt)
    A[3] = A[m - 5]
    Ad = d.cm_from_numpy(A)
    ip = torch.zeros(n, dtype=torch.int64, device="cuda")
    before = ctx.path_count(7)
    assert getattr(ctx.lib, f"rlhip_getrf_{dtype}")(ctx.h, m, n, Ad.data_ptr(), m, ip.data_ptr()) == 0
    assert (ctx.path_count(7) > before) == path7
    lu_ref, piv_ref, info_ref = (ll.dgetrf if dtype == "f64" else ll.sgetrf)(A)
    np.testing.assert_array_equal(ip.cpu().numpy() - 1, piv_ref)
    tol = (1e-12 if dtype == "f64" else 2e-4) * np.abs(lu_ref).max()
    assert np.abs(d.cm_to_numpy(Ad) - lu_ref).max() <= tol


def test_getrf_singular_reports_info_and_luqrcp_piv(ctx):
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    A = np.zeros((30, 6)); A[:, 0] = 1.0; A[3, 1] = 2.0                              # columns 2.. are exactly zero
    Ad = d.cm_from_numpy(A)
    ip = torch.zeros(6, dtype=torch.int64, device="cuda")
    info = ctx.lib.rlhip_getrf_f64(ctx.h, 30, 6, Ad.data_ptr(), 30, ip.data_ptr())
    _, piv_ref, info_ref = ll.dgetrf(A)
    assert info == info_ref > 0
    np.testing.assert_array_equal(ip.cpu().numpy() - 1, piv_ref)
    # pivot conversion (rl_bqrrp.hh:345-350): serial swaps on iota
    ipiv = np.array([5, 5, 9, 4, 7], dtype=np.int64)
    cols = 10
    J = torch.zeros(cols, dtype=torch.int64, device="cuda")
    assert ctx.lib.rlhip_luqrcp_piv(ctx.h, 5, cols, torch.from_numpy(ipiv).cuda().data_ptr(), J.data_ptr()) == 0
    ref = np.arange(1, cols + 1)
    for i in range(5):
        a = ipiv[i] - 1
        ref[a], ref[i] = ref[i], ref[a]
    np.testing.assert_array_equal(J.cpu().numpy(), ref)
    # at sketch size (the LDS-resident walk of lu.hip): positions beyond the sketch dimension chosen repeatedly, pivots inside it, identities
    rng = np.random.default_rng(11)
    for sd, cols in ((2048, 50000), (2048, 2048), (700, 701), (2048, 2100)):
        ipiv = np.array([rng.integers(i, cols) if rng.random() < 0.8 else i for i in range(sd)], dtype=np.int64) + 1
        ipiv[5:40] = np.minimum(cols, 2077)                                      # the same far row picked again and again
        ipiv = np.maximum(ipiv, np.arange(1, sd + 1))
        J = torch.zeros(cols, dtype=torch.int64, device="cuda")
        assert ctx.lib.rlhip_luqrcp_piv(ctx.h, sd, cols, torch.from_numpy(ipiv).cuda().data_ptr(), J.data_ptr()) == 0
        ref = np.arange(1, cols + 1)
        for i in range(min(sd, cols)):
            a = ipiv[i] - 1
            ref[a], ref[i] = ref[i], ref[a]
        np.testing.assert_array_equal(J.cpu().numpy(), ref)


@pytest.mark.parametrize("m,n", [(60, 12), (500, 64), (2000, 256), (300, 300), (64, 200), (512, 4096)])
def test_geqrf_ungqr_match_lapack(ctx, m, n):
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    rng = np.random.default_rng(m * 3 + n)
    A = rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A)
    k = min(m, n)
    tau = torch.zeros(k, dtype=torch.float64, device="cuda")
    assert ctx.lib.rlhip_geqrf_f64(ctx.h, m, n, Ad.data_ptr(), m, tau.data_ptr()) == 0
    ctx.sync()
    qr_ref, tau_ref, _, _ = ll.dgeqrf(A)
    tol = 1e-12 * np.abs(qr_ref).max()
    np.testing.assert_allclose(d.cm_to_numpy(Ad), qr_ref, atol=tol, rtol=0)          # same reflectors, same signs
    np.testing.assert_allclose(tau.cpu().numpy(), tau_ref, atol=1e-12, rtol=0)
    if m >= n:
        assert ctx.lib.rlhip_ungqr_f64(ctx.h, m, n, n, Ad.data_ptr(), m, tau.data_ptr()) == 0
        Q = d.cm_to_numpy(Ad)
        Qo = ll.dorgqr(qr_ref, tau_ref)[0]
        np.testing.assert_allclose(Q, Qo, atol=1e-12, rtol=0)
        assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= EPS**0.75 * np.sqrt(n)


# the register-resident block-pipelined kernel (qr_blk.hip: sketch-sized, nearly square or wide inputs up to 2048 rows / 8 columns per CU):
# the geqrf output of LAPACK entry for entry (same reflectors, same signs), in both precisions, with a ragged last chunk (n % 8 != 0), a
# row count that is not a multiple of the workgroup, a leading dimension larger than m, and columns beyond the last reflector
@pytest.mark.parametrize("m,n,dt", [(2048, 2048, "f32"), (2048, 2048 + 700, "f32"), (1280, 1024, "f64"), (1000, 1000, "f64"), (700, 653, "f64"),
                                    (2041, 2041, "f32"), (512, 4096, "f64"), (1536, 1100, "f32"), (96, 80, "f64")])
def test_geqrf_block_pipelined_matches_lapack(ctx, m, n, dt):
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    f64 = dt == "f64"
    npdt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    rng = np.random.default_rng(m + 5 * n)
    A = rng.standard_normal((m, n)).astype(npdt)
    lda = m + 6
    buf = np.full((lda, n), 7.0, dtype=npdt); buf[:m] = A
    Ad = d.cm_from_numpy(buf)
    k = min(m, n)
    tau = torch.zeros(k, dtype=tdt, device="cuda")
    before = ctx.path_count(8)
    fn = ctx.lib.rlhip_geqrf_f64 if f64 else ctx.lib.rlhip_geqrf_f32
    assert fn(ctx.h, m, n, Ad.data_ptr(), lda, tau.data_ptr()) == 0
    ctx.sync()
    assert ctx.path_count(8) == before + 1, "the block-pipelined kernel did not take this shape"
    qr_ref, tau_ref, _, _ = (ll.dgeqrf if f64 else ll.sgeqrf)(A)
    got = d.cm_to_numpy(Ad)
    tol = (2e-12 if f64 else 2e-3) * np.abs(qr_ref).max() * np.sqrt(k / 1000 + 1)       # rounding of ~k block updates per entry
    np.testing.assert_allclose(got[:m], qr_ref, atol=tol, rtol=0)
    np.testing.assert_allclose(tau.cpu().numpy(), tau_ref, atol=(1e-12 if f64 else 1e-4), rtol=0)
    assert np.all(got[m:] == 7.0)                                                      # nothing written below row m
    # the factorization itself: ||A - Q R|| at working precision
    Qref = (ll.dorgqr if f64 else ll.sorgqr)(np.asfortranarray(got[:m, :k].copy()), tau.cpu().numpy())[0]
    R = np.triu(got[:k])
    eps = np.finfo(npdt).eps
    assert np.linalg.norm(A.astype(np.float64) - Qref.astype(np.float64) @ R.astype(np.float64)) <= 30 * eps * np.linalg.norm(A) * np.sqrt(k)


@pytest.mark.parametrize("m,n", [(1024, 1024), (1024, 512), (20000, 64)])
@pytest.mark.parametrize("scale,dt", [(1e19, "f32"), (1e-21, "f32"), (1e160, "f64"), (1e-170, "f64")])
def test_geqrf_badly_scaled_input_is_rescaled(ctx, scale, dt, m, n):
    """The Householder kernels form column norms as plain sums of squares; an input whose squares leave the exponent range is
    found by a max-abs pass that stays on the device and factored as s A with s a power of two (same reflectors, same tau, R scaled
    back): LAPACK's result, on the block-pipelined, the flag-pipelined and the Cholesky-QR panel routes alike."""
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    f64 = dt == "f64"
    npdt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    rng = np.random.default_rng(11)
    A = (rng.standard_normal((m, n)) * scale).astype(npdt)
    Ad = d.cm_from_numpy(A)
    tau = torch.zeros(n, dtype=tdt, device="cuda")
    fn = ctx.lib.rlhip_geqrf_f64 if f64 else ctx.lib.rlhip_geqrf_f32
    assert fn(ctx.h, m, n, Ad.data_ptr(), m, tau.data_ptr()) == 0
    ctx.sync()
    qr_ref, tau_ref, _, _ = (ll.dgeqrf if f64 else ll.sgeqrf)(A)
    got = d.cm_to_numpy(Ad)
    assert np.all(np.isfinite(got)) and np.all(np.isfinite(tau.cpu().numpy()))
    np.testing.assert_allclose(np.abs(np.diag(got[:n])), np.abs(np.diag(qr_ref[:n])), rtol=(1e-10 if f64 else 2e-3))
    np.testing.assert_allclose(tau.cpu().numpy(), tau_ref, atol=(1e-11 if f64 else 2e-4), rtol=0)
    np.testing.assert_allclose(np.tril(got, -1), np.tril(qr_ref, -1), atol=(1e-10 if f64 else 3e-3), rtol=0)      # the reflectors do not scale


@pytest.mark.parametrize("m,n,nb", [(60, 12, 12), (500, 64, 32), (2000, 256, 256), (300, 100, 40), (2500, 2048, 2048), (1500, 1003, 1003)])
def test_orhr_col_gemqrt_larft_match_lapack(ctx, orc, m, n, nb):
    import torch

    d = _dev()
    rng = np.random.default_rng(m + nb)
    Q = np.linalg.qr(rng.standard_normal((m, n)))[0]
    Qd = d.cm_from_numpy(Q)
    Td = d.cm_zeros(nb, n)
    Dd = torch.zeros(n, dtype=torch.float64, device="cuda")
    lu0 = ctx.path_count(9)
    assert ctx.lib.rlhip_orhr_col_f64(ctx.h, m, n, nb, Qd.data_ptr(), m, Td.data_ptr(), nb, Dd.data_ptr()) == 0
    ctx.sync()
    assert ctx.path_count(9) == lu0 + (1 if n > 32 else 0)    # the sign-modified LU ran as one block-pipelined launch (qr_blk.hip); a single 32-column panel takes the LDS panel kernel
    info, Ao, To, Do = orc.lapack_orhr_col(Q, nb)
    assert info == 0
    np.testing.assert_array_equal(Dd.cpu().numpy(), Do)                               # sign vector: exact
    np.testing.assert_allclose(np.tril(d.cm_to_numpy(Qd), -1), np.tril(Ao, -1), atol=1e-12, rtol=0)
    np.testing.assert_allclose(d.cm_to_numpy(Td), To, atol=1e-12, rtol=0)
    # gemqrt (Left, Trans) against the explicit product of the LAPACK block reflectors
    Cm = rng.standard_normal((m, 7))
    Cd = d.cm_from_numpy(Cm)
    assert ctx.lib.rlhip_gemqrt_f64(ctx.h, b"L", b"T", m, 7, n, nb, Qd.data_ptr(), m, Td.data_ptr(), nb, Cd.data_ptr(), m) == 0
    Vfull = np.tril(Ao, -1) + np.eye(m, n)
    H = np.eye(m)
    for j0 in range(0, n, nb):
        jb = min(nb, n - j0)
        Vb = Vfull[:, j0:j0 + jb].copy()
        Vb[:j0] = 0
        H = H @ (np.eye(m) - Vb @ To[:jb, j0:j0 + jb] @ Vb.T)
    np.testing.assert_allclose(d.cm_to_numpy(Cd), H.T @ Cm, atol=1e-11, rtol=0)
    # tau_from_t / row_sign
    tau = torch.zeros(n, dtype=torch.float64, device="cuda")
    assert ctx.lib.rlhip_tau_from_t_f64(ctx.h, n, nb, Td.data_ptr(), nb, tau.data_ptr()) == 0
    np.testing.assert_allclose(tau.cpu().numpy(), np.array([To[i % nb, i] for i in range(n)]), atol=1e-12, rtol=0)


def test_larft_and_any_abs_gt(ctx):
    import scipy.linalg as sl
    import torch

    d = _dev()
    rng = np.random.default_rng(77)
    m, k = 300, 40
    A = rng.standard_normal((m, k))
    (qr_, tau_), _ = sl.qr(A, mode="raw")
    Vd = d.cm_from_numpy(np.asfortranarray(qr_))
    taud = torch.from_numpy(tau_).cuda()
    Td = d.cm_zeros(k, k)
    assert ctx.lib.rlhip_larft_f64(ctx.h, m, k, Vd.data_ptr(), m, taud.data_ptr(), Td.data_ptr(), k) == 0
    Vf = np.tril(qr_, -1)[:, :k] + np.eye(m, k)
    Tn = d.cm_to_numpy(Td)
    Hq = np.eye(m) - Vf @ Tn @ Vf.T
    Qs = sl.qr(A)[0]
    np.testing.assert_allclose(Hq[:, :k], Qs[:, :k], atol=1e-12, rtol=0)
    np.testing.assert_allclose(np.diag(Tn), tau_, atol=1e-13, rtol=0)
    import ctypes as C

    x = torch.zeros(1000, dtype=torch.float64, device="cuda")
    flag = C.c_int(5)
    assert ctx.lib.rlhip_any_abs_gt_f64(ctx.h, 1000, x.data_ptr(), EPS, C.byref(flag)) == 0 and flag.value == 0
    x[777] = -3e-16
    assert ctx.lib.rlhip_any_abs_gt_f64(ctx.h, 1000, x.data_ptr(), EPS, C.byref(flag)) == 0 and flag.value == 1


@pytest.mark.parametrize("dist", [0, 1])
def test_fill_dense_rows_is_a_slice_of_the_global_operator(ctx, dist):
    import ctypes as C

    d = _dev()
    m, k = 1003, 37
    full = d.cm_empty(m, k)
    ctr = (C.c_uint32 * 4)(5, 0, 0, 0); key = (C.c_uint32 * 2)(9, 1); nxt = (C.c_uint32 * 4)()
    assert ctx.lib.rlhip_fill_dense_f64(ctx.h, dist, m, k, full.data_ptr(), ctr, key, nxt) == 0
    F = d.cm_to_numpy(full)
    for row0, rows in [(0, 400), (400, 603), (17, 1), (1002, 1), (0, 1003)]:
        part = d.cm_empty(rows, k)
        nxt2 = (C.c_uint32 * 4)()
        assert ctx.lib.rlhip_fill_dense_rows_f64(ctx.h, dist, m, k, row0, rows, part.data_ptr(), rows, ctr, key, nxt2) == 0
        np.testing.assert_array_equal(d.cm_to_numpy(part), F[row0:row0 + rows])      # bit-identical
        assert list(nxt2) == list(nxt)


@pytest.mark.parametrize("m,n,kind", [(400, 32, "cluster"), (2000, 64, "cluster"), (256, 256, "identity-like"), (5000, 128, "two-clusters"), (512, 512, "identity-like"), (900, 400, "cluster")])
def test_gesdd_clustered_singular_values(ctx, m, n, kind):
    """Nearly multiple singular values: tiny cosines still need large rotation angles, so a cosine-based early exit must be
    verified (regression: the unverified shortcut left U^T U - I at 1e-9 on the B factor of an RSVD with sigma_1..32 ~ 1)."""
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    if kind == "cluster":
        s = 1.0 - 1e-7 * rng.random(n)
    elif kind == "identity-like":
        s = np.ones(n)
    else:
        s = np.concatenate([np.full(n // 2, 3.0), 1.0 + 1e-9 * rng.random(n - n // 2)])
    s = np.sort(s)[::-1]
    A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    Ad = d.cm_from_numpy(A)
    S = torch.zeros(n, dtype=torch.float64, device="cuda")
    U = d.cm_empty(m, n)
    VT = d.cm_empty(n, n)
    assert ctx.lib.rlhip_gesdd_f64(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, None) == 0
    Un, Sn, VTn = d.cm_to_numpy(U), S.cpu().numpy(), d.cm_to_numpy(VT)
    assert np.linalg.norm(Un.T @ Un - np.eye(n)) <= 1e-12 * np.sqrt(n)
    assert np.linalg.norm(VTn @ VTn.T - np.eye(n)) <= 1e-12 * np.sqrt(n)
    assert np.linalg.norm((Un * Sn) @ VTn - A) <= 1e-13 * np.linalg.norm(A)
    np.testing.assert_allclose(Sn, s, rtol=1e-12)


@pytest.mark.parametrize("m,n,kind", [(20000, 256, "flat"), (5000, 256, "cond10"), (3000, 200, "flat"), (4000, 96, "cluster"), (2000, 256, "identity")])
def test_gesdd_persistent_jacobi_equals_per_launch_sweeps(ctx, monkeypatch, m, n, kind):
    """The one-launch Jacobi (resident workgroups exchanging blocks through the uncached buffer, jacobi_persist_kernel) runs the same
    pairing and the same round arithmetic as the per-launch sweeps: S, U and V^T come out BITWISE identical, with the same sweep count;
    path counter 6 proves which one ran.  `cluster` / `identity` take the hand-back to the host's Gram verification and a relaunch.
    Option value 2 = the persistent launch with the hand-over through uncached memory only; 1 lets the workers hand blocks over through
    their XCD's L2 when the census finds them on one XCD (jacobi.hip) -- a transport change only: bitwise the same again."""
    import ctypes as C
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    if kind == "flat":
        A = rng.standard_normal((m, n))
    else:
        s = {"cond10": np.logspace(0, -1, n), "cluster": 1.0 - 1e-7 * rng.random(n), "identity": np.ones(n)}[kind]
        A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    res = {}
    ctx.set_option("gesdd_gram", 0)                    # the classic route (Cholesky-QR + Jacobi on R^T): the one that has both sweep drivers
    for mode in ("1", "0", "2"):
        ctx.set_option("jacobi_persist", int(mode))
        Ad = d.cm_from_numpy(A)
        S = torch.zeros(n, dtype=torch.float64, device="cuda")
        U, VT = d.cm_empty(m, n), d.cm_empty(n, n)
        sw = C.c_int(0)
        before = ctx.path_count(6)
        assert ctx.lib.rlhip_gesdd_f64(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, C.byref(sw)) == 0
        res[mode] = (d.cm_to_numpy(U), S.cpu().numpy(), d.cm_to_numpy(VT), sw.value, ctx.path_count(6) - before)
    (U1, S1, V1, sw1, c1), (U0, S0, V0, sw0, c0) = res["1"], res["0"]
    assert c1 == 1 and c0 == 0, "the persistent kernel did not run (or ran with the jacobi_persist option off)"
    assert sw1 == sw0 and sw1 > 0
    assert np.array_equal(S1, S0) and np.array_equal(U1, U0) and np.array_equal(V1, V0)
    U2, S2, V2, sw2, c2 = res["2"]
    assert c2 == 1 and sw2 == sw1
    assert np.array_equal(S1, S2) and np.array_equal(U1, U2) and np.array_equal(V1, V2)
    assert np.linalg.norm((U1 * S1) @ V1 - A) <= 1e-13 * np.linalg.norm(A) * np.sqrt(n)
    assert np.linalg.norm(U1.T @ U1 - np.eye(n)) <= 1e-11 * np.sqrt(n)


@pytest.mark.parametrize("m,n,dtype", [(131072, 1024, "f64"), (65536, 512, "f32"), (20000, 256, "f64")])
def test_trsm_fused_asm_and_plain_loads_agree_bitwise(ctx, monkeypatch, m, n, dtype):
    """The fused solve exists in two builds inside the library: X-operand loads issued from inline asm behind the kernel's own counted
    waits (the default when scripts/check_trsm_asm.py has proven the build's register allocation safe) and plain C++ loads.  Same
    arithmetic, same order: the solutions must be BITWISE equal -- a register the allocator moved before its load landed would show here."""
    import torch

    d = _dev()
    rng = np.random.default_rng(n)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    R = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2.0 * np.eye(n)
    B = rng.standard_normal((m, n))
    Rd = d.cm_from_numpy(R).to(tdt)
    res = {}
    for mode in ("1", "0"):
        ctx.set_option("trsm_xasm", int(mode))
        Bd = d.cm_from_numpy(B).to(tdt)
        before = ctx.path_count(2)
        ctx.trsm(m, n, 1.0, Rd, n, Bd, m)
        assert ctx.path_count(2) > before, "the fused solve did not run"
        res[mode] = Bd.clone()
    assert torch.equal(res["1"], res["0"])
    X = d.cm_to_numpy(res["1"])[:4096].astype(np.float64)
    eps = np.finfo(np.float64 if dtype == "f64" else np.float32).eps
    assert np.linalg.norm(X @ R - B[:4096]) <= 200 * eps * np.linalg.norm(B[:4096]) * np.sqrt(n)


@pytest.mark.parametrize("m,n", [(40970, 512), (49152, 256), (33000, 1024)])
def test_trsm_fused_fp32_192_row_workgroups_equal_128_row_ones_bitwise(ctx, m, n):
    """fp32 in-place fused solve: row counts whose 128-row workgroups would leave the last round over the CUs half empty run 192-row
    workgroups (twelve wavefronts; tri.hip::tf_launch_hpr).  A row of X depends on its own row of B only and both instantiations do the
    same arithmetic in the same order, so the head and the tail of the tall solve must equal, BIT FOR BIT, the same rows solved as two
    20000-row problems (128-row workgroups); residual checked in fp64 on both ends (ragged last workgroup and ragged last wavefront)."""
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    R = (np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2.0 * np.eye(n)).astype(np.float32)
    B = rng.standard_normal((m, n)).astype(np.float32)
    Rd = d.cm_from_numpy(R)
    full = d.cm_from_numpy(B)
    assert full.dtype == torch.float32
    before = ctx.path_count(2)
    ctx.trsm(m, n, 1.0, Rd, n, full, m)
    assert ctx.path_count(2) > before, "the fused solve did not run"
    X = d.cm_to_numpy(full)
    h = 20000
    for lo in (0, m - h):
        part = d.cm_from_numpy(B[lo:lo + h])
        ctx.trsm(h, n, 1.0, Rd, n, part, h)
        assert np.array_equal(d.cm_to_numpy(part), X[lo:lo + h])
    eps = np.finfo(np.float32).eps
    for sl in (slice(0, 2048), slice(m - 2048, m)):
        assert np.linalg.norm(X[sl].astype(np.float64) @ R.astype(np.float64) - B[sl]) <= 200 * eps * np.linalg.norm(B[sl]) * np.sqrt(n)


@pytest.mark.parametrize("m,n,kind,gram", [(20000, 256, "flat", True), (3000, 200, "flat", True), (5000, 128, "cond5", True), (1000, 256, "flat", True), (300, 256, "flat", None),
                                           (2000, 256, "cond100", False), (4000, 96, "cond1e6", False), (1500, 64, "rank-deficient", False)])
def test_gesdd_gram_route(ctx, monkeypatch, m, n, kind, gram):
    """The Gram route of the device SVD (one-sided Jacobi on A^T A itself, U = A W, one host read; svd.hip::gesdd_tall_gram) serves
    well-conditioned tall factors -- path counter 10 says when -- and hands everything else to the classic route untouched: either way
    the result is LAPACK's to rounding, and the two routes agree on a well-conditioned input."""
    import ctypes as C
    import torch

    d = _dev()
    rng = np.random.default_rng(3 * m + n)
    if kind == "flat":
        A = rng.standard_normal((m, n))
    else:
        s = {"cond5": np.linspace(5, 1, n), "cond100": np.logspace(0, -2, n), "cond1e6": np.logspace(0, -6, n),
             "rank-deficient": np.concatenate([np.linspace(2, 1, n - 4), np.zeros(4)])}[kind]
        A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    s_ref = np.linalg.svd(A, compute_uv=False)
    out = {}
    for route in ("1", "0"):
        ctx.set_option("gesdd_gram", int(route))
        Ad = d.cm_from_numpy(A)
        S = torch.zeros(n, dtype=torch.float64, device="cuda")
        U, VT = d.cm_empty(m, n), d.cm_empty(n, n)
        sw = C.c_int(0)
        before = ctx.path_count(10)
        assert ctx.lib.rlhip_gesdd_f64(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, C.byref(sw)) >= 0
        took = ctx.path_count(10) - before
        if gram is not None or route == "0":                        # (None: cond ~ 25, eps cond^2 sits at the route's threshold -- either route may serve it)
            assert took == (1 if (gram and route == "1") else 0), f"route {route}: Gram route taken {took} times"
        Un, Sn, VTn = d.cm_to_numpy(U), S.cpu().numpy(), d.cm_to_numpy(VT)
        keep = s_ref > 1e-12 * s_ref[0]
        np.testing.assert_allclose(Sn[keep], s_ref[keep], rtol=1e-11 if kind == "cond1e6" else 1e-12)
        assert np.all(np.diff(Sn) <= 0)
        assert np.linalg.norm((Un * Sn) @ VTn - A) <= 1e-13 * np.linalg.norm(A) * np.sqrt(n)
        assert np.linalg.norm(VTn @ VTn.T - np.eye(n)) <= 1e-11 * np.sqrt(n)
        if kind != "rank-deficient":
            assert np.linalg.norm(Un.T @ Un - np.eye(n)) <= 1e-11 * np.sqrt(n)
        out[route] = (Un, Sn, VTn)
    if gram and kind != "flat":                                     # separated singular values: the two routes deliver the same vectors up to sign
        (U1, S1, V1), (U0, S0, V0) = out["1"], out["0"]
        sgn = np.sign(np.sum(V1 * V0, axis=1))
        assert np.max(np.abs(V1 - sgn[:, None] * V0)) <= 1e-9 and np.max(np.abs(U1 - U0 * sgn)) <= 1e-9


@pytest.mark.parametrize("m,n,cond", [(20000, 128, 1e2), (20000, 64, 1e12), (100000, 32, 1.0), (9000, 16, 1e9)])
def test_geqrf_tall_skinny_matches_lapack(ctx, m, n, cond):
    """Tall-skinny geqrf: Cholesky-QR twice + Householder reconstruction when it can be trusted; for cond 1e9 / 1e12 plain Cholesky-QR
    gives up and the sketch-preconditioned retry (S A, its small QR, A R_sk^-1, Cholesky-QR twice) must take the panel -- not the
    column-by-column Householder kernel.  Either way the GEQRF-format output equals LAPACK's to rounding."""
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    rng = np.random.default_rng(m + n)
    s = np.logspace(0, -np.log10(cond), n) if cond > 1 else np.ones(n)
    A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    Ad = d.cm_from_numpy(A)
    tau = torch.zeros(n, dtype=torch.float64, device="cuda")
    pre = ctx.path_count(5)
    assert ctx.lib.rlhip_geqrf_f64(ctx.h, m, n, Ad.data_ptr(), m, tau.data_ptr()) == 0
    ctx.sync()
    assert ctx.path_count(5) - pre == (1 if cond >= 1e9 else 0)
    qr_ref, tau_ref, _, _ = ll.dgeqrf(A)
    out = d.cm_to_numpy(Ad)
    # R rows scale with the singular values: compare relative to each row of R; V and tau absolutely
    Rg, Rr = np.triu(out[:n]), np.triu(qr_ref[:n])
    assert np.linalg.norm(Rg - Rr) <= 1e-9 * np.linalg.norm(Rr)
    if cond < 1e9:   # reflectors of an ill-conditioned matrix are sensitive to eps * cond perturbations: two correct Householder QRs
        np.testing.assert_allclose(np.tril(out, -1), np.tril(qr_ref, -1), atol=1e-9, rtol=0)   # need not agree entrywise there
        np.testing.assert_allclose(tau.cpu().numpy(), tau_ref, atol=1e-9, rtol=0)
    assert ctx.lib.rlhip_ungqr_f64(ctx.h, m, n, n, Ad.data_ptr(), m, tau.data_ptr()) == 0
    Q = d.cm_to_numpy(Ad)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= 1e-12 * np.sqrt(n)
    assert np.linalg.norm(Q @ Rg - A) <= 1e-13 * np.linalg.norm(A)


# ---------------------------------------------------------------------------------------------------
# fp32 instantiations of the kernel families (BASELINE config 4 computes in fp32): each against a float64 numpy / LAPACK result on the
# same (fp32-representable) input, tolerances in units of eps32
# ---------------------------------------------------------------------------------------------------
def test_f32_kernel_families(ctx):
    import scipy.linalg as sl
    import scipy.linalg.lapack as ll
    import torch

    d = _dev()
    f32 = torch.float32
    e32 = float(np.finfo(np.float32).eps)
    rng = np.random.default_rng(32)
    m, n = 70000, 512                                       # tall enough for the MFMA block trsm (m >= 65536: two row tiles per wave)
    A = rng.standard_normal((m, n)).astype(np.float32)
    A64 = A.astype(np.float64)
    Ad = d.cm_from_numpy(A)
    # syrk (upper) + potrf
    G = d.cm_zeros(n, n, dtype=f32)
    ctx.syrk("U", "T", n, m, 1.0, Ad, m, 0.0, G, n)
    Gref = A64.T @ A64
    g = d.cm_to_numpy(G).astype(np.float64)
    assert np.abs(np.triu(g) - np.triu(Gref)).max() <= 4 * e32 * np.sqrt(m) * np.abs(Gref).max() / np.sqrt(m) * 8
    assert np.all(np.tril(g, -1) == 0)                       # LAPACK uplo contract
    assert ctx.potrf(n, G, n) == 0
    R = np.triu(d.cm_to_numpy(G)).astype(np.float64)
    assert np.linalg.norm(R.T @ R - np.triu(g) - np.triu(g, 1).T) <= 50 * e32 * np.linalg.norm(Gref)
    # trsm: Q = A R^-1 (MFMA block path at this shape), then trmm back
    ctx.trsm(m, n, 1.0, G, n, Ad, m)
    Q = d.cm_to_numpy(Ad).astype(np.float64)
    assert np.linalg.norm(Q @ R - A64) <= 20 * e32 * np.linalg.norm(A64)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= 200 * e32 * np.sqrt(n)          # cond(A)^2 * eps32 for a Gaussian A: a few
    ctx.trmm(m, n, 1.0, G, n, Ad, m)
    assert np.linalg.norm(d.cm_to_numpy(Ad).astype(np.float64) - A64) <= 40 * e32 * np.linalg.norm(A64)
    # geqrf + ungqr (pipelined kernel for this width) against LAPACK's reflectors
    m2, n2 = 3000, 200
    B = rng.standard_normal((m2, n2)).astype(np.float32)
    Bd = d.cm_from_numpy(B)
    tau = torch.zeros(n2, dtype=f32, device="cuda")
    assert ctx.lib.rlhip_geqrf_f32(ctx.h, m2, n2, Bd.data_ptr(), m2, tau.data_ptr()) == 0
    qr_ref, tau_ref, _, _ = ll.dgeqrf(B.astype(np.float64))
    assert np.abs(d.cm_to_numpy(Bd).astype(np.float64) - qr_ref).max() <= 200 * e32 * np.abs(qr_ref).max()   # same reflectors, same signs
    assert np.abs(tau.cpu().numpy().astype(np.float64) - tau_ref).max() <= 50 * e32
    assert ctx.lib.rlhip_ungqr_f32(ctx.h, m2, n2, n2, Bd.data_ptr(), m2, tau.data_ptr()) == 0
    Q2 = d.cm_to_numpy(Bd).astype(np.float64)
    assert np.linalg.norm(Q2.T @ Q2 - np.eye(n2)) <= 20 * e32 * np.sqrt(n2)
    assert np.linalg.norm(Q2 @ np.triu(qr_ref[:n2]) - B) <= 50 * e32 * np.linalg.norm(B)
    # geqp3: pivot order identical to LAPACK's on well separated column norms
    m3, n3 = 1200, 96
    Cm = (rng.standard_normal((m3, n3)) * np.logspace(0, -3, n3)[rng.permutation(n3)]).astype(np.float32)
    Cd = d.cm_from_numpy(Cm)
    J = torch.zeros(n3, dtype=torch.int64, device="cuda")
    tau3 = torch.zeros(n3, dtype=f32, device="cuda")
    assert ctx.lib.rlhip_geqp3_f32(ctx.h, m3, n3, Cd.data_ptr(), m3, J.data_ptr(), tau3.data_ptr()) == 0
    jref = ll.sgeqp3(Cm)[1]
    np.testing.assert_array_equal(J.cpu().numpy(), jref)
    Rg = np.triu(d.cm_to_numpy(Cd)[:n3]).astype(np.float64)
    _, rref = sl.qr(Cm.astype(np.float64)[:, jref - 1], mode="economic")
    assert np.abs(np.abs(np.diag(Rg)) - np.abs(np.diag(rref))).max() <= 50 * e32 * abs(rref[0, 0])
    # col_swap: exact
    perm = rng.permutation(n3) + 1
    Cd = d.cm_from_numpy(Cm)
    Jp = torch.from_numpy(perm.astype(np.int64)).cuda()
    assert ctx.lib.rlhip_col_swap_f32(ctx.h, m3, n3, n3, Cd.data_ptr(), m3, Jp.data_ptr()) == 0
    np.testing.assert_array_equal(d.cm_to_numpy(Cd), Cm[:, perm - 1])
    np.testing.assert_array_equal(Jp.cpu().numpy(), perm)


def test_lange_extreme_magnitudes_and_col_swap_rejects_non_permutations(ctx):
    """lange_fro on entries whose squares over- / underflow (LAPACK's scaled dlassq gives the right norm); col_swap's device path
    (n < 4096) leaves the matrix untouched for an index vector that is not a permutation instead of walking it forever."""
    import torch

    d = _dev()
    for scale in (1e200, 1e-200):
        A = np.array([[3.0, 0.0], [4.0, 0.0], [0.0, 12.0]]) * scale
        got = ctx.lange_fro(3, 2, d.cm_from_numpy(A), 3)
        assert abs(got / scale - 13.0) <= 1e-12
    assert ctx.lange_fro(3, 2, d.cm_zeros(3, 2), 3) == 0.0
    assert np.isnan(ctx.lange_fro(2, 2, d.cm_from_numpy(np.array([[1.0, np.nan], [0.0, 2.0]])), 2))
    A = np.arange(12.0).reshape(3, 4)
    Ad = d.cm_from_numpy(A)
    bad = torch.tensor([2, 2, 5, 1], dtype=torch.int64, device="cuda")
    assert ctx.lib.rlhip_col_swap_f64(ctx.h, 3, 4, 4, Ad.data_ptr(), 3, bad.data_ptr()) == 0
    ctx.sync()
    assert np.array_equal(d.cm_to_numpy(Ad), A)
    good = torch.tensor([4, 1, 3, 2], dtype=torch.int64, device="cuda")
    assert ctx.lib.rlhip_col_swap_f64(ctx.h, 3, 4, 4, Ad.data_ptr(), 3, good.data_ptr()) == 0
    assert np.array_equal(d.cm_to_numpy(Ad), A[:, [3, 0, 2, 1]])


@pytest.mark.parametrize("dd,m,n,nnz", [(1280, 1280 * 9 + 300, 8, 4), (96, 96 * 12 + 10, 12, 3), (1280, 1280 * 10, 4, 8), (40, 40 * 30 + 6, 16, 4)])
def test_saso_apply_lds_dma_route_whole_and_row_shards(ctx, orc, dd, m, n, nnz):
    """The LDS-DMA apply kernel (independent columns, d <= 1280, whole 4-column slabs: sketch.hip::saso_apply_dma_kernel) against the dense
    product of the oracle's operator: whole operand (ragged last block -> the register-staged kernel's partial group), and the operand cut
    into row shards whose windows start and end INSIDE a block (ragged head and tail); every result bitwise reproducible."""
    import ctypes as C

    d = _dev()
    rng = np.random.default_rng(dd + n)
    u32 = lambda v: (C.c_uint32 * len(v))(*v)
    S = C.c_void_p(); nxt = (C.c_uint32 * 4)()
    ctr, key = (11, 0, 0, 0), (2, 7)
    assert ctx.lib.rlhip_saso_create_mode(ctx.h, dd, m, nnz, 1, u32(ctr), u32(key), nxt, C.byref(S)) == 0
    So, _ = orc.saso_dense(dd, m, nnz, ctr, key, 1)
    A = rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A)
    ref = So @ A
    tol = 1e-13 * np.abs(ref).max() * nnz * 8
    Bd = d.cm_zeros(dd, n)
    before = ctx.path_count(14)
    assert ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, 1.0, Ad.data_ptr(), m, 0.0, Bd.data_ptr(), dd) == 0
    assert ctx.path_count(14) == before + 1, "the LDS-DMA kernel did not take this shape"
    B1 = d.cm_to_numpy(Bd)
    assert np.abs(B1 - ref).max() <= tol
    Bd2 = d.cm_zeros(dd, n)
    assert ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, 1.0, Ad.data_ptr(), m, 0.0, Bd2.data_ptr(), dd) == 0
    assert np.array_equal(d.cm_to_numpy(Bd2), B1)
    # three row shards with even, block-interior cuts; each rank holds only its rows (lda = its row count)
    cuts = [0, (m // 3) & ~1, (2 * m // 3 + 2) & ~1, m]
    acc = np.zeros((dd, n))
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        Ash = d.cm_from_numpy(np.ascontiguousarray(A[r0:r1]))
        Bs = d.cm_zeros(dd, n)
        assert ctx.lib.rlhip_saso_apply_rows_f64(ctx.h, S, n, 1.0, Ash.data_ptr(), r1 - r0, r0, r1 - r0, 0.0, Bs.data_ptr(), dd) == 0
        part = d.cm_to_numpy(Bs)
        assert np.abs(part - So[:, r0:r1] @ A[r0:r1]).max() <= tol
        acc += part
    assert np.abs(acc - ref).max() <= 2 * tol
    ctx.lib.rlhip_saso_destroy(ctx.h, S)


def test_gesdd_jacobi_same_xcd_route_is_taken_and_changes_no_bit():
    """The persistent Jacobi's same-XCD hand-over (jacobi.hip: workers on one XCD exchange blocks through that XCD's L2 instead of uncached
    memory) rests on an OBSERVATION -- workgroup b lands on XCD b % 8 -- guarded by a census of HW_REG_XCC_ID: if it stops holding the launch
    silently takes the uncached protocol and the gain is gone.  On a context of its own (nothing else enqueued beside it) path counter 15 must
    show that the local route WAS taken with option 1, was not with option 2 (uncached always) or 3 (ordinary launch, the profiling route),
    and all three return the same bits."""
    import ctypes as C
    import torch

    d = _dev()
    c = d.Context(0, use_torch_stream=False)
    try:
        m, n = 20000, 256
        A = d.cm_empty(m, n); c.fill_dense(A, m, n, key=(4, 0)); c.sync()
        res = {}
        for mode in (1, 2, 3):
            c.set_option("jacobi_persist", mode)
            W = A.clone()
            torch.cuda.synchronize()
            S = torch.zeros(n, dtype=torch.float64, device="cuda")
            U, VT = d.cm_empty(m, n), d.cm_empty(n, n)
            sw = C.c_int(0)
            b6, b15 = c.path_count(6), c.path_count(15)
            assert c.lib.rlhip_gesdd_f64(c.h, m, n, W.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, C.byref(sw)) == 0
            c.sync()
            res[mode] = (S.clone(), U.clone(), VT.clone(), sw.value, c.path_count(6) - b6, c.path_count(15) - b15)
        assert all(r[4] >= 1 for r in res.values()), "the persistent kernel did not run"
        assert res[1][5] >= 1, "option 1: the workers were not found on one XCD (the same-XCD hand-over was never exercised on this device)"
        assert res[2][5] == 0 and res[3][5] == 0
        for mode in (2, 3):
            assert res[mode][3] == res[1][3]
            assert torch.equal(res[mode][0], res[1][0]) and torch.equal(res[mode][1], res[1][1]) and torch.equal(res[mode][2], res[1][2]), mode
    finally:
        c.close()


@pytest.mark.parametrize("m,n,cond,dtype", [(200000, 32, 10.0, "f64"), (20000, 256, 1e3, "f64"), (4096, 64, 1e6, "f64"), (50000, 64, 10.0, "f32"),
                                            (3000, 40, 1e12, "f64"), (100, 64, 10.0, "f64")])
def test_geqrf_q_equals_geqrf_then_ungqr(ctx, m, n, cond, dtype):
    """rlhip_geqrf_q (geqrf + ungqr(m, n, n) in one pass over a tall panel: ABRIK's Krylov blocks rl_abrik.hh:333-342, HQRQ rl_orth.hh:157-162)
    against the two calls it replaces, on the device: the same Q (Householder sign convention included) and the same triangle to rounding; a
    panel it does not take (not tall: m < 2 n) comes back untouched with return code 1; an ill-conditioned one is either served through the
    sketch-preconditioned route or handed back -- never wrong."""
    import torch

    d = _dev()
    tdt = torch.float64 if dtype == "f64" else torch.float32
    eps = float(np.finfo(np.float64 if dtype == "f64" else np.float32).eps)
    rng = np.random.default_rng(m + n)
    A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * np.logspace(0, -np.log10(cond), n)) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    A0 = d.cm_from_numpy(A).to(tdt)
    # the two calls
    A2 = A0.clone()
    tau = torch.zeros(n, dtype=tdt, device="cuda")
    assert getattr(ctx.lib, f"rlhip_geqrf_{dtype}")(ctx.h, m, n, A2.data_ptr(), m, tau.data_ptr()) == 0
    R2 = torch.triu(A2[:, :n].T.clone())                       # column-major tensor: A2[j][i] = entry (i, j)
    assert getattr(ctx.lib, f"rlhip_ungqr_{dtype}")(ctx.h, m, n, n, A2.data_ptr(), m, tau.data_ptr()) == 0
    # the one call
    A1 = A0.clone()
    R1 = torch.full((n, n), float("nan"), dtype=tdt, device="cuda")
    rc = getattr(ctx.lib, f"rlhip_geqrf_q_{dtype}")(ctx.h, m, n, A1.data_ptr(), m, R1.data_ptr(), n)
    ctx.sync()
    if m < 2 * n:
        assert rc == 1 and torch.equal(A1, A0)
        return
    assert rc in (0, 1)
    if rc == 1:                                                 # handed back: the input must still be there (to rounding)
        assert float((A1 - A0).abs().max()) <= 64 * eps * float(A0.abs().max())
        assert cond >= 1e8, "a well-conditioned tall panel was not taken"
        return
    Q1, Q2 = A1.double(), A2.double()
    R1m, R2m = R1.T.double(), R2.double()
    tol = 200 * eps * max(cond, 1.0) if cond < 1e8 else 1e-3     # (the columns of Q that belong to the tiny singular values move by eps * cond between any two routes)
    assert float((Q1 @ Q1.T - torch.eye(n, device="cuda", dtype=torch.float64)).abs().max()) <= 100 * eps * np.sqrt(n)          # (n x m)(m x n): Q^T Q
    assert float((R1m - torch.triu(R1m)).abs().max()) == 0.0                                                                     # zero below the diagonal
    assert float((Q1 - Q2).abs().max()) <= tol, "Q differs from geqrf + ungqr (sign convention or values)"
    assert float((R1m - R2m).abs().max()) <= tol * float(R2m.abs().max())
    assert float(((R1m.T @ Q1).T - A0.double().T).abs().max()) <= 100 * eps * float(A0.abs().max()) * np.sqrt(n)                # A = Q R
