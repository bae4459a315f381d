"""-m gpu: the kernels and configurations that carry the headline numbers, checked at the sizes where they engage.

* the persistent stream-K GEMM (gemm_sk.hip; NN, TN and the TN upper-triangle "tri" map) entry-wise against numpy at shapes above its
  work gate, with the path asserted through rlhip_path_count, the fused ||A||_F and bitwise run-to-run equality
  (reference precedent for kernel-vs-CPU unit parity: test/comps/test_util_gpu.cu:198-385);
* BASELINE configs[1] (RSVD 200000 x 20000): Y = A * Omega on row samples and range(U) == range(Y);
* BASELINE configs[3] (BQRRP 65536 x 65536 fp32) on one device through size-independent properties, plus a 4096^2 fp32 shared-sketch
  comparison with the fp64 oracle;
* BASELINE configs[4] (ABRIK, rank 128): the largest single-device cases -- a dense 200000 x 20000 operator and a 200000 x 200000
  implicit (CSR) operator -- with the residual metric of test/drivers/test_abrik.cc:96-123.
Tolerances are stated at each assertion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = float(np.finfo(np.float64).eps)
EPS32 = float(np.finfo(np.float32).eps)


def _d():
    from randlapack_amd import device

    return device


def relerr(got, ref):
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)


# ---------------------------------------------------------------------------------------------------
# stream-K GEMM, direct
# ---------------------------------------------------------------------------------------------------
# (M, N, K, transA): W = tiles * ktiles >= 256 * 64 so that the persistent kernel takes the problem (gemm_sk.hip gate);
# 645 rows leave a peeled block for the generic kernel (an odd remainder is not made of whole 16-byte pieces), 1446 a partial last tile row
# inside the persistent launch (lda must be even: 16-byte aligned columns for the LDS DMA)
SK_SHAPES = [(38400, 256, 1024, "N"), (1280, 512, 16384, "N"), (1408 + 38, 256, 32768, "N"), (256, 256, 131072, "T"),
             (640 + 5, 256, 65536, "T"), (2048, 512, 8192, "T")]


@pytest.mark.parametrize("m,n,k,ta", SK_SHAPES)
def test_streamk_gemm_entrywise(ctx, m, n, k, ta):
    d = _d()
    rng = np.random.default_rng(m + n + k)
    A, B, C0 = rng.standard_normal((m, k)), rng.standard_normal((k, n)), rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A if ta == "N" else A.T.copy())
    Bd = d.cm_from_numpy(B)
    lda = m if ta == "N" else k
    before = ctx.path_count(0)
    Cd = d.cm_from_numpy(C0)
    ctx.gemm(ta, "N", m, n, k, 1.5, Ad, lda, Bd, k, -0.5, Cd, m)
    assert ctx.path_count(0) == before + 1, "the persistent stream-K kernel did not take this shape"
    r1 = d.cm_to_numpy(Cd)
    ref = 1.5 * (A @ B) - 0.5 * C0
    assert relerr(r1, ref) <= 50 * EPS * np.sqrt(k)          # fp64 dot products of length k, entry-wise
    # bitwise run-to-run equality (fixed share boundaries, fix-up sums partial slabs in k order)
    Cd2 = d.cm_from_numpy(C0)
    ctx.gemm(ta, "N", m, n, k, 1.5, Ad, lda, Bd, k, -0.5, Cd2, m)
    assert np.array_equal(r1, d.cm_to_numpy(Cd2))
    # beta = 0 must not read C (NaNs in the output buffer stay out of the result)
    Cd3 = d.cm_from_numpy(np.full((m, n), np.nan))
    ctx.gemm(ta, "N", m, n, k, 1.0, Ad, lda, Bd, k, 0.0, Cd3, m)
    assert relerr(d.cm_to_numpy(Cd3), A @ B) <= 50 * EPS * np.sqrt(k)


# a partial last tile row made of whole 16-byte pieces stays inside the persistent launch (its DMA pieces re-read the last valid rows and the
# epilogue drops the rows that do not exist): nothing is written below row m, nothing of the duplicated rows reaches C or the fused norm
@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("ta", ["N", "T"])
@pytest.mark.parametrize("rem", [4, 64, 124])
def test_streamk_partial_last_tile_row(ctx, dt, ta, rem):
    d = _d()
    import torch

    f64 = dt == "f64"
    npdt, tdt = (np.float64, torch.float64) if f64 else (np.float32, torch.float32)
    m, n, k = (1280 if f64 else 5120) + rem, 256, (32768 if f64 else 16384)     # above the work gate of the persistent kernel
    rng = np.random.default_rng(rem + (7 if f64 else 9))
    A = rng.standard_normal((m, k)).astype(npdt); B = rng.standard_normal((k, n)).astype(npdt)
    Ad = d.cm_from_numpy(A if ta == "N" else A.T.copy()); Bd = d.cm_from_numpy(B)
    ldc = m + 4
    C0 = rng.standard_normal((ldc, n)).astype(npdt)
    Cd = d.cm_from_numpy(C0)
    which = 0 if f64 else 1
    before = ctx.path_count(which)
    ctx.gemm(ta, "N", m, n, k, 2.0, Ad, m if ta == "N" else k, Bd, k, 1.0, Cd, ldc)
    assert ctx.path_count(which) == before + 1
    got = d.cm_to_numpy(Cd)
    ref = 2.0 * (A.astype(np.float64) @ B.astype(np.float64)) + C0[:m]
    tol = (50 * EPS if f64 else 4 * EPS32) * np.sqrt(k)
    assert relerr(got[:m], ref) <= tol
    assert np.array_equal(got[m:], C0[m:])                      # the four guard rows below every column are untouched
    if f64:
        Cz = d.cm_zeros(m, n)
        nrm, fused = ctx.gemm_norma(ta, "N", m, n, k, 1.0, Ad, m if ta == "N" else k, Bd, k, 0.0, Cz, m)
        assert fused == 1 and abs(nrm - np.linalg.norm(A)) <= 1e-13 * np.linalg.norm(A)


@pytest.mark.parametrize("m,n,k,ta", [(38400, 256, 1024, "N"), (1408 + 38, 256, 32768, "N"), (256, 256, 131072, "T"), (645, 256, 65536, "T")])
def test_streamk_fused_frobenius_norm(ctx, m, n, k, ta):
    d = _d()
    rng = np.random.default_rng(3 * m + k)
    A, B = rng.standard_normal((m, k)), rng.standard_normal((k, n))
    Ad = d.cm_from_numpy(A if ta == "N" else A.T.copy())
    Bd = d.cm_from_numpy(B)
    Cd = d.cm_zeros(m, n)
    nrm, fused = ctx.gemm_norma(ta, "N", m, n, k, 1.0, Ad, m if ta == "N" else k, Bd, k, 0.0, Cd, m)
    assert fused == 1                                        # the norm came out of the GEMM's own pass over A
    assert abs(nrm - np.linalg.norm(A)) <= 1e-13 * np.linalg.norm(A)
    assert relerr(d.cm_to_numpy(Cd), A @ B) <= 50 * EPS * np.sqrt(k)
    nrm2, _ = ctx.gemm_norma(ta, "N", m, n, k, 1.0, Ad, m if ta == "N" else k, Bd, k, 0.0, Cd, m)
    assert nrm2 == nrm                                       # deterministic reduction tree


@pytest.mark.parametrize("n,k", [(1024, 32768), (512, 65536), (256, 131072), (768, 49152)])
def test_streamk_syrk_upper_tiles(ctx, n, k):
    """C(upper) = alpha A^T A + beta C: upper triangle entry-wise, strictly lower part untouched -- through the persistent kernel's tri tile
    map from 8 tiles on (n >= 768), through the tiled kernel's upper tiles + split-K below (a 2- or 5-tile map would make every one of the
    256 workgroups write a partial slab: round 4)."""
    d = _d()
    persistent = n >= 768
    rng = np.random.default_rng(n + k)
    A = rng.standard_normal((k, n))
    C0 = rng.standard_normal((n, n))
    Ad = d.cm_from_numpy(A)
    Cd = d.cm_from_numpy(C0)
    before = ctx.path_count(0)
    ctx.syrk("U", "T", n, k, 2.0, Ad, k, 0.25, Cd, n)
    assert ctx.path_count(0) == before + (1 if persistent else 0)
    got = d.cm_to_numpy(Cd)
    ref = 2.0 * (A.T @ A) + 0.25 * C0
    iu = np.triu_indices(n)
    assert np.abs(got[iu] - ref[iu]).max() <= 50 * EPS * np.sqrt(k) * np.abs(ref).max()
    il = np.tril_indices(n, -1)
    assert np.array_equal(got[il], C0[il])                   # LAPACK syrk contract (CQRRPT's trmm relies on it)
    Cd2 = d.cm_from_numpy(C0)
    ctx.syrk("U", "T", n, k, 2.0, Ad, k, 0.25, Cd2, n)
    assert np.array_equal(got, d.cm_to_numpy(Cd2))


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[1]: the sketch pass itself
# ---------------------------------------------------------------------------------------------------
def test_rsvd_full_size_sketch_pass(ctx):
    """Y = A * Omega at 200000 x 20000 x 256 (the bench's dominant launch) against torch on row samples, B^T = A^T Y on column
    samples, and range(U) of the full RSVD == range(Y): a wrong tile anywhere in Y changes the captured subspace."""
    import torch

    d = _d()
    m, n, k = 200000, 20000, 256
    A = d.cm_empty(m, n)
    ctx.fill_dense(A, m, n, key=(7, 0))
    Om = d.cm_empty(n, k)
    ctx.fill_dense(Om, n, k, key=(0, 0))                     # the Omega RSVD draws with the default state (ctr 0, key 0)
    Y = d.cm_empty(m, k)
    before = ctx.path_count(0)
    ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)
    assert ctx.path_count(0) == before + 1
    g = torch.Generator(device="cpu").manual_seed(1)
    rows = torch.cat([torch.randint(0, m, (400,), generator=g), torch.arange(0, 130), torch.arange(m - 130, m)]).to("cuda")
    ref = A[:, rows].T @ Om.T                                # (len, n) @ (n, k)
    got = Y[:, rows].T
    assert float((ref - got).abs().max() / ref.abs().max()) <= 50 * EPS * np.sqrt(n)
    # every 128-row tile is touched by at least one checksum: column sums of Y against (1^T A) Omega
    ones = torch.ones(m, dtype=torch.float64, device="cuda")
    cs = (A @ ones) @ Om.T                                   # (n,) @ (n, k)
    assert float((Y @ ones - cs).abs().max() / cs.abs().max()) <= 1e-9
    # per-tile checksums: sums over each block of 128 rows (1562 whole row tiles + the 64-row remainder)
    mt = (m // 128) * 128
    Yp = torch.cat([Y[:, :mt].reshape(k, -1, 128).sum(-1), Y[:, mt:].sum(-1, keepdim=True)], dim=1)      # (k, tiles)
    Ap = torch.cat([A[:, :mt].reshape(n, -1, 128).sum(-1), A[:, mt:].sum(-1, keepdim=True)], dim=1)      # (n, tiles)
    reft = Om @ Ap                                           # (k, n) @ (n, tiles)
    assert float((Yp - reft).abs().max() / reft.abs().max()) <= 1e-9
    del Yp, Ap, reft
    # the TN pass on the same data
    BT = d.cm_empty(n, k)
    ctx.gemm("T", "N", n, k, m, 1.0, A, m, Y, m, 0.0, BT, n)
    assert ctx.path_count(0) == before + 2
    cols = torch.cat([torch.arange(0, n, 211), torch.arange(n - 70, n)]).to("cuda")
    refb = A[cols] @ Y.T                                     # (len, m) @ (m, k)
    assert float((refb - BT[:, cols].T).abs().max() / refb.abs().max()) <= 50 * EPS * np.sqrt(m)
    # the driver's own Y spans the same subspace: U (k, m) from RSVD with p = 0 and the default state
    r = d.drv_rsvd(ctx, A, m, n, k, k, 1e-12, 0, 1)
    U = r["U"]
    P = U @ Y.T                                              # (k, m) @ (m, k) = U^T Y
    resid = Y - (P.T @ U)                                    # (k, m): Y^T - (U^T Y)^T U
    assert float(torch.linalg.norm(resid) / torch.linalg.norm(Y)) <= 1e-10


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[3]: BQRRP 65536 x 65536 fp32 on one device
# ---------------------------------------------------------------------------------------------------
def _planted_c2_matrix(ctx, d, m, n, k, sig, eta):
    """A = U diag(sig) V^T + eta * G in HBM: U (m x k) and V (n x k) are Cholesky-QR-orthonormalised Gaussian blocks (twice), G iid N(0,1)."""
    import torch

    U = d.cm_empty(m, k); ctx.fill_dense(U, m, k, key=(101, 0))
    V = d.cm_empty(n, k); ctx.fill_dense(V, n, k, key=(102, 0))
    for X, rows in ((U, m), (V, n)):
        for _ in range(2):
            rc, fail = d.drv_stab(ctx, 0, X, rows, k)
            assert rc == 0 and not fail
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(103, 0))
    Us = U * torch.as_tensor(sig, device=U.device)[:, None]                 # column-major (k, m) view: scales column i of U by sig[i]
    ctx.gemm("N", "T", m, n, k, 1.0, Us, m, V, n, eta, A, m)                # A <- (U sig) V^T + eta * G
    ctx.sync()
    return A, U, V


@pytest.mark.parametrize("rs_stab,name", [(0, "CholQRQ"), (2, "PLUL")])
def test_rsvd_config2_power_iterations_planted_rank256(ctx, rs_stab, name):
    """BASELINE configs[1] at p = 2 (C2 is defined at p in {0, 2}) on a rank-256-plus-noise matrix of the full size
    200000 x 20000: the two power passes (rl_rs.hh:126-178) with their stabilisers running on 200000 x 256 and 20000 x 256 blocks.
    Planted sigma_i from 1 down to 0.1, noise eta = 1e-12 per entry: first-order perturbation of sigma_i is eta (1e-11 relative at the
    small end), so the computed singular values must match the PLANTED ones to 1e-10 relative; the residual ||A - U S V^T||_F must
    sit on the noise floor eta * sqrt(m n) and the return codes must be the clean ones (QB reaches tol)."""
    import torch

    d = _d()
    m, n, k = 200000, 20000, 256
    sig = np.geomspace(1.0, 0.1, k)
    eta = 1e-12
    A, Ut, Vt = _planted_c2_matrix(ctx, d, m, n, k, sig, eta)
    before = ctx.path_count(0)
    r = d.drv_rsvd(ctx, A, m, n, k, k, 1e-6, 2, 1, rs_stab=rs_stab, key=(5, 0))
    assert ctx.path_count(0) - before >= 4                                  # four passes over A, each through the stream-K kernel
    assert r["rc"] == 0 and r["qb_rc"] == 0 and r["k"] == k
    assert r["next_ctr"] == (n * k // 4, 0, 0, 0)                           # one n x k Gaussian fill, whatever p is
    S = r["S"].cpu().numpy()
    assert np.max(np.abs(S - sig) / sig) <= 1e-10, f"{name}: sigma vs planted {np.max(np.abs(S - sig) / sig):.2e}"
    U, V = r["U"], r["V"]
    I = torch.eye(k, device="cuda", dtype=torch.float64)
    assert float(torch.linalg.norm(U @ U.T - I)) <= EPS**0.75 * np.sqrt(n)
    assert float(torch.linalg.norm(V @ V.T - I)) <= EPS**0.75 * np.sqrt(n)
    # the computed subspaces are the planted ones: ||U_planted^T U|| has all singular values 1 (to the noise level)
    c = torch.linalg.svdvals(Ut @ U.T)
    assert float(c.min()) >= 1 - 1e-10
    c = torch.linalg.svdvals(Vt @ V.T)
    assert float(c.min()) >= 1 - 1e-10
    # residual on the noise floor: A <- A - (U S) V^T in place, then its Frobenius norm
    Us = U * r["S"][:, None]
    ctx.gemm("N", "T", m, n, k, -1.0, Us, m, V, n, 1.0, A, m)
    res = ctx.lange_fro(m, n, A, m)
    floor = eta * np.sqrt(float(m) * n)
    assert 0.9 * floor <= res <= 1.1 * floor, f"{name}: residual {res:.3e} vs noise floor {floor:.3e}"


def _apply_qt_householder(V, tau, X, b):
    """Q^T X for Q = H_1 ... H_k stored LAPACK-style in V (column-major tensor (n, m): V[j] is column j), in fp64 on the device,
    panel by panel through the compact-WY form built here from V and tau alone (T^-1 = striu(V^T V) + diag(1 / tau)).
    Independent of the product's own apply kernels."""
    import torch

    n, m = V.shape
    k = tau.numel()
    for j0 in range(0, k, b):
        jb = min(b, k - j0)
        Vp = V[j0:j0 + jb, j0:].to(torch.float64).T.contiguous()        # (rows, jb)
        Vp = torch.tril(Vp, -1)
        Vp[torch.arange(jb), torch.arange(jb)] = 1.0
        t = tau[j0:j0 + jb].to(torch.float64)
        z = t == 0                                                      # H = I (LAPACK's last reflector of a square matrix): drop the column
        Vp[:, z] = 0
        Tinv = torch.triu(Vp.T @ Vp, 1) + torch.diag(torch.where(z, torch.ones_like(t), 1.0 / torch.where(z, torch.ones_like(t), t)))
        W = Vp.T @ X[j0:]                                               # (jb, cols)
        W = torch.linalg.solve_triangular(Tinv.T.contiguous(), W, upper=False)   # Q^T = I - V T^T V^T
        X[j0:] -= Vp @ W
    return X


def test_bqrrp_config4_full_size_f32(ctx):
    import torch

    d = _d()
    m = n = 65536
    b = 2048
    A = d.cm_empty(m, n, dtype=torch.float32)
    ctx.fill_dense(A, m, n, key=(4, 0))
    r = d.drv_bqrrp(ctx, A, m, n, b, 1.0, key=(6, 0), qrcp_wide=0, qr_tall=1, apply_trans_q=1)   # Cholesky-QR panels, as the reference's GPU benchmark runs them
    assert r["rc"] == 0 and r["rank"] == n
    J = r["J"]
    assert torch.equal(torch.sort(J).values, torch.arange(1, n + 1, device="cuda"))      # a permutation
    assert r["next_ctr"][0] == (b * m + 3) // 4                                           # one d x m Gaussian sketch operator
    tau = r["tau"]
    assert bool(torch.isfinite(tau).all()) and float(tau[:-1].min()) >= 1.0 - 1e-5 and float(tau.max()) <= 2.0 + 1e-2   # tau in [1, 2] (fp32 reconstruction)
    dR = torch.diagonal(A).abs().double()
    assert float(dR.min()) > 0
    # QRCP quality, size-independent: block maxima of |r_ii| do not grow from block to block (pivoting acts across blocks), and inside
    # a block |r_ii| is essentially non-increasing
    blk = dR.reshape(-1, b)
    bmax = blk.max(dim=1).values
    assert bool((bmax[1:] <= bmax[:-1] * 1.05).all())
    assert float((dR[1:] <= dR[:-1] * 1.5).double().mean()) > 0.95
    # A[:, J] = Q R on a column sample: Q^T a_j must reproduce column j of R and vanish below the diagonal
    A0 = d.cm_empty(m, n, dtype=torch.float32)
    ctx.fill_dense(A0, m, n, key=(4, 0))
    g = torch.Generator(device="cpu").manual_seed(5)
    cols = torch.cat([torch.randint(0, n, (20,), generator=g), torch.tensor([0, 1, b - 1, b, n - b - 1, n - 2, n - 1])]).to("cuda")
    X = A0[(J[cols] - 1)].to(torch.float64).T.contiguous()              # (m, ncols): the sampled columns of A[:, J]
    del A0
    nrm = torch.linalg.norm(X, dim=0)
    X = _apply_qt_householder(A, tau, X, b)
    Rs = A[cols].to(torch.float64).T.contiguous()                       # (m, ncols): sampled columns of the factored matrix ...
    Rs[torch.arange(m, device="cuda")[:, None] > cols[None, :]] = 0     # ... with the reflectors below the diagonal masked out = R
    err = torch.linalg.norm(X - Rs, dim=0) / nrm
    # test_bqrrp.cc:105-107 asks ||A[:, J] - Q R|| <= eps^0.75 ||A|| at n <= a few thousand; at n = 65536 in fp32 the backward error of
    # ANY Householder QR is ~ sqrt(n) eps32 = 1.5e-5 per column, so the bound is stated in that unit (measured: 2e-5 typical, 2.2e-4 max)
    assert float(err.max()) <= 16 * np.sqrt(n) * EPS32 and float((err**2).mean().sqrt()) <= 2 * np.sqrt(n) * EPS32
    # norm preservation alone (independent of the reflector replay): ||R[:, j]|| = ||A[:, J_j]||
    assert float(((torch.linalg.norm(Rs, dim=0) - nrm).abs() / nrm).max()) <= 1e-4


def test_bqrrp_4096_f32_vs_f64_oracle_shared_sketch(ctx, orc):
    """fp32 device BQRRP against the fp64 oracle on the same (fp32-representable) matrix and the same sketch.  Column scales are
    mildly graded (a factor 44 over the matrix); the pivot order among the Gaussian columns is decided by 1 %-level fluctuations of
    the sketched column norms, far above fp32 rounding except for rare near-ties -- hence set-wise comparison per block."""
    import torch

    d = _d()
    m = n = 4096
    b = 256
    rng = np.random.default_rng(44)
    scale = 1.03 ** (-rng.permutation(n).astype(np.float64) / 32.0)
    A = (rng.standard_normal((m, n)) * scale).astype(np.float32).astype(np.float64)
    Ad = d.cm_from_numpy(A).to(torch.float32)
    r = d.drv_bqrrp(ctx, Ad, m, n, b, 1.0, want_sketch=True, key=(21, 0), qrcp_wide=0, qr_tall=1, apply_trans_q=1)
    sk = d.cm_to_numpy(r["sketch"]).astype(np.float64)
    o = orc.bqrrp(A, b, 1.0, qrcp_wide=0, qr_tall=1, apply_trans_q=1, sketch=sk)
    assert r["rc"] == o["rc"] == 0 and r["rank"] == o["rank"] == n
    J, Jo = r["J"].cpu().numpy(), o["J"]
    assert sorted(J.tolist()) == list(range(1, n + 1))
    # pivots: (nearly) the same column SET in every block of b (the order inside a block may flip on fp32-level near-ties of the
    # LU pivots, and a tie across a block boundary moves one column to the next block)
    overlap = [len(set(J[i:i + b].tolist()) & set(Jo[i:i + b].tolist())) / b for i in range(0, n, b)]
    np.testing.assert_array_equal(J[:16], Jo[:16])                      # the first pivots of the first block: exact
    assert min(overlap[:4]) == 1.0 and np.mean(overlap) >= 0.8, overlap   # (measured: 1.0 for the leading blocks, 0.89 on average)
    Aout = d.cm_to_numpy(Ad).astype(np.float64)
    dR, dRo = np.abs(np.diag(Aout)), np.abs(np.diag(o["A"]))
    # |r_ii| profile: block-wise geometric means agree to 1 % (they are invariant under reordering inside a block to first order)
    gm = np.exp(np.log(dR).reshape(-1, b).mean(1)) / np.exp(np.log(dRo).reshape(-1, b).mean(1))
    assert np.abs(gm - 1).max() <= 1e-2
    tau = r["tau"].cpu().numpy().astype(np.float64)
    Q = orc.ungqr(Aout, tau)
    R = np.triu(Aout)
    assert np.linalg.norm(A[:, J - 1] - Q @ R) <= 4 * EPS32**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= 4 * EPS32**0.75 * np.sqrt(n)
    # like for like: the restatement instantiated on float (s-prefixed LAPACK) on the same float matrix and float sketch.  Both sides now
    # carry fp32 rounding, in different summation orders: the pivot sequences agree wherever a decision is not a float-level near-tie.
    o32 = orc.bqrrp(A.astype(np.float32), b, 1.0, qrcp_wide=0, qr_tall=1, apply_trans_q=1, sketch=d.cm_to_numpy(r["sketch"]))
    assert o32["rc"] == 0 and o32["rank"] == n and o32["A"].dtype == np.float32
    J32 = o32["J"]
    np.testing.assert_array_equal(J[:b], J32[:b])                       # the whole first block: exact
    ov32 = [len(set(J[i:i + b].tolist()) & set(J32[i:i + b].tolist())) / b for i in range(0, n, b)]
    same = float(np.mean(J == J32))
    assert np.mean(ov32) >= np.mean(overlap) - 0.02 and same >= 0.25, (ov32, same)
    dR32 = np.abs(np.diag(o32["A"].astype(np.float64)))
    gm32 = np.exp(np.log(dR).reshape(-1, b).mean(1)) / np.exp(np.log(dR32).reshape(-1, b).mean(1))
    assert np.abs(gm32 - 1).max() <= 1e-2


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[4]: ABRIK, rank 128, the largest single-device operators
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,nc", [(200000, 20000, 32), (65544, 4096, 64), (50001, 2000, 16)])
def test_skinny_operator_products_entrywise(ctx, m, n, nc):
    """ABRIK's operator products on a dense linop (rl_abrik.hh:311,358; BASELINE configs[4] dense leg: 200000 x 20000 against 32
    vectors): A X and A^T Y entry by entry against numpy on row / column samples, alpha / beta honoured, ragged row counts, bitwise
    repeatable.  (A dedicated streaming kernel for these shapes was built and measured in round 3 -- 4.53 TB/s at 32 columns against
    4.54 for the tiled kernel that serves them -- and not adopted)"""
    import torch

    d = _d()
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(21, 0))
    X = d.cm_empty(n, nc); ctx.fill_dense(X, n, nc, key=(22, 0))
    Y = d.cm_empty(m, nc); ctx.fill_dense(Y, m, nc, key=(23, 0))
    C = d.cm_empty(m, nc); ctx.fill_dense(C, m, nc, key=(24, 0))
    C0 = C.clone()
    ctx.gemm("N", "N", m, nc, n, 0.75, A, m, X, n, -0.5, C, m)
    rows = torch.cat([torch.arange(0, 40), torch.arange(m // 2, m // 2 + 40), torch.arange(m - 40, m)]).cuda()
    ref = 0.75 * (A[:, rows].T.cpu().numpy() @ X.T.cpu().numpy()) - 0.5 * C0[:, rows].T.cpu().numpy()
    got = C[:, rows].T.cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.sqrt(n) * np.max(np.abs(ref))
    Z = d.cm_empty(n, nc); ctx.fill_dense(Z, n, nc, key=(25, 0))
    ctx.gemm("T", "N", n, nc, m, 1.0, A, m, Y, m, 0.0, Z, n)
    cols = torch.cat([torch.arange(0, 24), torch.arange(n - 24, n)]).cuda()
    ref = A[cols].cpu().numpy() @ Y.T.cpu().numpy()
    got = Z[:, cols].T.cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.sqrt(m) * np.max(np.abs(ref))
    Z2 = d.cm_empty(n, nc)
    ctx.gemm("T", "N", n, nc, m, 1.0, A, m, Y, m, 0.0, Z2, n)
    assert torch.equal(Z, Z2)


def _abrik_residual(AV, ATU, U, S, V):
    """test/drivers/test_abrik.cc:96-123: hypot(||A V - U S||_F, ||A^T U - V S||_F) (column-major tensors (t, rows))"""
    import torch

    n1 = torch.linalg.norm(AV - U * S[:, None])
    n2 = torch.linalg.norm(ATU - V * S[:, None])
    return float(torch.hypot(n1, n2))


def test_abrik_config5_dense_200000x20000_rank128(ctx):
    import torch

    d = _d()
    m, n, rank_true, k = 200000, 20000, 64, 32
    target = 128
    L = d.cm_empty(m, rank_true)
    Rf = d.cm_empty(n, rank_true)
    ctx.fill_dense(L, m, rank_true, key=(31, 0))
    ctx.fill_dense(Rf, n, rank_true, key=(32, 0))
    # exact rank 64 = two blocks with a flat spectrum: the third Krylov block is linearly dependent, ABRIK's sqrt(eps) test on the new
    # block's R factor (rl_abrik.hh:455-458) ends the iteration, and every triplet is then exact to rounding
    s = torch.linspace(1.0, 0.1, rank_true, dtype=torch.float64, device="cuda") / np.sqrt(float(m) * n)
    A = d.cm_empty(m, n)
    Ls = (L * s[:, None]).contiguous()
    ctx.gemm("N", "T", m, n, rank_true, 1.0, Ls, m, Rf, n, 0.0, A, m)        # A = L diag(s) R^T
    op = d.DenseOperator(A, m, n)
    iters = 2 * target // k                                                  # test_abrik.cc:139
    r = d.drv_abrik_linop(ctx, op, k, EPS**0.85, iters, key=(1, 0))
    assert r["rc"] == 0
    t = r["triplets"]
    assert t == rank_true and r["iters"] < iters                             # stopped by the rank, not by the budget
    U, S, V = r["U"], r["S"], r["V"]
    I = torch.eye(t, device="cuda", dtype=torch.float64)
    assert float(torch.linalg.norm(U @ U.T - I)) <= 1e-10 and float(torch.linalg.norm(V @ V.T - I)) <= 1e-10
    assert bool((S[:-1] >= S[1:]).all())
    c = 32                                                                   # custom_rank <= rank
    AV = (A.T @ V[:c].T).T.contiguous()                                      # (c, m)
    ATU = (A @ U[:c].T).T.contiguous()                                       # (c, n)
    res = _abrik_residual(AV, ATU, U[:c], S[:c], V[:c])
    assert res <= 10 * EPS**0.825 * float(S[0]) * np.sqrt(c)                 # the reference's bound, relative to sigma_1
    # singular values against the exact ones of L diag(s) R^T (small 64 x 64 problem)
    Ql, Rl = torch.linalg.qr(L.T)
    Qr, Rr = torch.linalg.qr(Rf.T)
    sv = torch.linalg.svdvals(Rl @ torch.diag(s) @ Rr.T)
    assert float(((S[:rank_true] - sv).abs() / sv).max()) <= 1e-9


def test_abrik_config5_implicit_200000x200000_rank128(ctx):
    """A = D1 * G * D2 held as a CSR operator (a band of 10 Gaussian entries per row, never densified): 200000 x 200000, block 32,
    8 Krylov iterations (rank 128).  Graded diagonal scalings (entries decay like exp(-(i + j) / 4)) make the leading triplets
    converge inside the iteration budget."""
    import scipy.sparse as sp
    import torch

    d = _d()
    m = n = 200000
    k, target = 32, 128
    rng = np.random.default_rng(77)
    nnz_row = 10
    rows = np.repeat(np.arange(m), nnz_row)
    colsi = (rows + np.tile(np.arange(-4, 6), m)) % n
    vals = rng.standard_normal(m * nnz_row)
    d1 = np.exp(-np.arange(m) / 4.0) + 1e-13
    d2 = np.exp(-np.arange(n) / 4.0) + 1e-13
    G = sp.csr_matrix((vals * d1[rows] * d2[colsi], (rows, colsi)), shape=(m, n))
    G.sum_duplicates()
    op = d.CsrOperator.from_scipy(G)
    iters = 2 * target // k
    r = d.drv_abrik_linop(ctx, op, k, EPS**0.85, iters, key=(2, 0))
    assert r["rc"] == 0
    t = r["triplets"]
    assert t >= k
    U, S, V = d.cm_to_numpy(r["U"]), r["S"].cpu().numpy(), d.cm_to_numpy(r["V"])
    assert np.linalg.norm(U.T @ U - np.eye(t)) <= 1e-10 and np.linalg.norm(V.T @ V - np.eye(t)) <= 1e-10
    c = 8
    res = np.hypot(np.linalg.norm(G @ V[:, :c] - U[:, :c] * S[:c]), np.linalg.norm(G.T @ U[:, :c] - V[:, :c] * S[:c]))
    assert res <= 10 * EPS**0.825 * S[0] * np.sqrt(c)
    # the dense top-left corner carries the spectrum (entries decay like exp(-(i + j) / 4)): compare with its SVD
    sv = np.linalg.svd(G[:2000, :2000].toarray(), compute_uv=False)
    assert np.max(np.abs(S[:8] - sv[:8]) / sv[:8]) <= 1e-8


# ---------------------------------------------------------------------------------------------------
# fused trsm (tri.hip::trsm_fused_kernel): B <- alpha B inv(U), tall B
# ---------------------------------------------------------------------------------------------------
def _np_trsm_right_upper(B, U, alpha):
    import scipy.linalg as sl

    return alpha * sl.solve_triangular(U, B.T, trans="T", lower=False).T      # X U = alpha B


@pytest.mark.parametrize("m,n,dtype", [(20000, 1024, "f64"), (16500, 1000, "f64"), (33000, 300, "f64"), (20000, 1024, "f32"),
                                       (16390, 520, "f32")])
def test_trsm_fused_blocks_vs_lapack(ctx, m, n, dtype):
    import torch

    d = _d()
    rng = np.random.default_rng(m + n)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
    U += np.tril(rng.standard_normal((n, n)), -1)              # strictly-lower garbage must be ignored
    B = rng.standard_normal((m, n))
    tdt = torch.float64 if dtype == "f64" else torch.float32
    npdt = np.float64 if dtype == "f64" else np.float32
    U = U.astype(npdt).astype(np.float64)
    B = B.astype(npdt).astype(np.float64)
    Bd = d.cm_from_numpy(B).to(tdt)
    before = ctx.path_count(2)
    ctx.trsm(m, n, 1.5, d.cm_from_numpy(U).to(tdt), n, Bd, m)
    assert ctx.path_count(2) == before + 1, "the fused block kernel did not take this solve"
    X = d.cm_to_numpy(Bd).astype(np.float64)
    ref = _np_trsm_right_upper(B, np.triu(U), 1.5)
    eps = EPS if dtype == "f64" else EPS32
    assert relerr(X, ref) <= 200 * eps                          # well-conditioned triangle: forward error at rounding level
    assert relerr(X @ np.triu(U), 1.5 * B) <= 100 * eps         # backward residual
    # run to run bitwise identical
    Bd2 = d.cm_from_numpy(B).to(tdt)
    ctx.trsm(m, n, 1.5, d.cm_from_numpy(U).to(tdt), n, Bd2, m)
    assert torch.equal(Bd, Bd2)


@pytest.mark.parametrize("m,n,dtype,perm,fused", [(20000, 1024, "f64", True, True), (16500, 512, "f64", False, True), (33000, 300, "f64", True, False),
                                                  (20000, 768, "f32", True, True), (900, 256, "f64", True, False)])
def test_trsm_gather_out_of_place_vs_lapack(ctx, m, n, dtype, perm, fused):
    """rlhip_trsm_gather: B = alpha (Bsrc P) inv(U) with the pivot vector read inside the solve (CQRRPT's col_swap + trsm in one pass);
    the source is left untouched; fused launch where every block qualifies, gather-copy + in-place solver elsewhere -- same numbers."""
    import torch

    d = _d()
    rng = np.random.default_rng(m * 3 + n)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
    U += np.tril(rng.standard_normal((n, n)), -1)
    B = rng.standard_normal((m, n))
    tdt = torch.float64 if dtype == "f64" else torch.float32
    npdt = np.float64 if dtype == "f64" else np.float32
    U = U.astype(npdt).astype(np.float64)
    B = B.astype(npdt).astype(np.float64)
    jp = rng.permutation(n) + 1 if perm else None
    Jd = torch.from_numpy(jp.astype(np.int64)).cuda() if perm else None
    Sd = d.cm_from_numpy(B).to(tdt)
    S0 = Sd.clone()
    Xd = d.cm_from_numpy(np.full((m, n), np.nan)).to(tdt)
    before = ctx.path_count(4)
    ctx.trsm_gather(m, n, 0.75, d.cm_from_numpy(U).to(tdt), n, Sd, m, Jd, Xd, m)
    assert ctx.path_count(4) == before + (1 if fused else 0)
    assert torch.equal(Sd, S0)                                   # the source is read only
    X = d.cm_to_numpy(Xd).astype(np.float64)
    Bp = B[:, jp - 1] if perm else B
    ref = _np_trsm_right_upper(Bp, np.triu(U), 0.75)
    eps = EPS if dtype == "f64" else EPS32
    assert relerr(X, ref) <= 200 * eps
    # the same numbers as "permute, then solve in place"
    Pd = d.cm_from_numpy(Bp).to(tdt)
    ctx.trsm(m, n, 0.75, d.cm_from_numpy(U).to(tdt), n, Pd, m)
    assert relerr(X, d.cm_to_numpy(Pd).astype(np.float64)) <= 4 * eps
    if fused and m >= 16384 and n % 256 == 0:
        assert torch.equal(Xd, Pd)                               # both take the fused kernel: identical arithmetic per entry


@pytest.mark.parametrize("m,n", [(20000, 512), (3000, 200)])
def test_trsm_gather_rejects_a_pivot_vector_that_is_not_a_permutation(ctx, m, n):
    """rlhip_trsm_gather validates jpvt on the device before anything is written (fused route: the verdict rides on the conditioning
    guard's read-back; gather-copy route: its own check): a repeated or out-of-range entry returns -7 and leaves B untouched."""
    import torch
    from randlapack_amd import _lib

    d = _d()
    rng = np.random.default_rng(n)
    U = np.triu(rng.standard_normal((n, n))) + 30 * np.eye(n)
    Ud, Src = d.cm_from_numpy(U), d.cm_from_numpy(rng.standard_normal((m, n)))
    for bad in ("repeat", "range"):
        J = np.arange(1, n + 1, dtype=np.int64)[::-1].copy()
        if bad == "repeat":
            J[7] = J[3]
        else:
            J[5] = n + 1
        B = d.cm_zeros(m, n)
        B.fill_(-3.0)
        with pytest.raises(_lib.RlhipError):
            ctx.trsm_gather(m, n, 1.0, Ud, n, Src, m, torch.from_numpy(J).cuda(), B, m)
        assert bool((B == -3.0).all())
    J = torch.arange(n, 0, -1, dtype=torch.int64, device="cuda")
    B = d.cm_zeros(m, n)
    ctx.trsm_gather(m, n, 1.0, Ud, n, Src, m, J, B, m)
    X = d.cm_to_numpy(B)
    assert np.linalg.norm(X @ U - d.cm_to_numpy(Src)[:, ::-1]) <= 1e-12 * np.linalg.norm(d.cm_to_numpy(Src)) * np.sqrt(n)


def test_trsm_gather_with_an_ill_conditioned_block_takes_the_copy_route(ctx):
    """One 256-block fails the conditioning guard -> no fused out-of-place launch: columns gathered by the copy kernel, then the
    in-place solver (fused runs around the bad block, substitution inside it).  Same accuracy as the in-place solve of the permuted input."""
    import torch
    import scipy.linalg as sl

    d = _d()
    m, n = 16640, 768
    rng = np.random.default_rng(8)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
    sv = np.array([1.0] * 8 + [1e-9] * 248)
    Mx = rng.standard_normal((256, 256))
    Qa, _ = np.linalg.qr(Mx)
    Qb, _ = np.linalg.qr(rng.standard_normal((256, 256)))
    _, Rb, _ = sl.qr((Qa * sv) @ Qb.T, pivoting=True)
    U[256:512, 256:512] = Rb
    jp = rng.permutation(n) + 1
    Bp = rng.standard_normal((m, n)) @ np.triu(U)              # the permuted right-hand side has an O(1) solution
    B = np.empty_like(Bp)
    B[:, jp - 1] = Bp                                          # ... and the source holds its columns scattered: B[:, jp - 1] = Bp
    Jd = torch.from_numpy(jp.astype(np.int64)).cuda()
    Xd = d.cm_zeros(m, n)
    b4, b2, b3 = ctx.path_count(4), ctx.path_count(2), ctx.path_count(3)
    ctx.trsm_gather(m, n, 1.0, d.cm_from_numpy(U), n, d.cm_from_numpy(B), m, Jd, Xd, m)
    assert ctx.path_count(4) == b4 and ctx.path_count(2) == b2 + 2 and ctx.path_count(3) == b3 + 8
    X = d.cm_to_numpy(Xd)
    assert np.linalg.norm(X @ np.triu(U) - Bp) <= 1e-13 * n * np.linalg.norm(Bp)


def test_trsm_fused_with_an_ill_conditioned_block_in_the_middle(ctx):
    """Blocks 0 and 2 (256 columns each) are well conditioned and go through the fused kernel; block 1 carries a graded diagonal
    (cond 1e12) and must take the substitution path (explicit inverses would lose eps * cond); the residual stays at eps ||B||."""
    d = _d()
    rng = np.random.default_rng(17)
    m, n = 17000, 768
    import scipy.linalg as sl

    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
    # the R factor of a pivoted QR of a matrix whose singular values drop from 1 to 1e-9 after the eighth: graded rows, |r_ij| <= |r_ii|
    # (what CQRRPT's R_sk looks like on a numerically rank-deficient sketch); its first 32 x 32 diagonal block has cond ~ 1e9
    sv = np.concatenate([np.ones(8), 1e-9 * np.ones(248)])
    Z = (np.linalg.qr(rng.standard_normal((400, 256)))[0] * sv) @ np.linalg.qr(rng.standard_normal((256, 256)))[0]
    U[256:512, 256:512] = sl.qr(Z, mode="economic", pivoting=True)[1]
    B = rng.standard_normal((m, n)) @ U
    Bd = d.cm_from_numpy(B)
    f0, s0 = ctx.path_count(2), ctx.path_count(3)
    ctx.trsm(m, n, 1.0, d.cm_from_numpy(U), n, Bd, m)
    assert ctx.path_count(2) == f0 + 2 and ctx.path_count(3) == s0 + 8      # two fused runs around eight substitution sub-blocks
    X = d.cm_to_numpy(Bd)
    assert np.linalg.norm(X @ U - B) <= 1e-13 * np.linalg.norm(B) * n


# ---------------------------------------------------------------------------------------------------
# fp32 twin of the stream-K kernel (BQRRP's compact-WY products at BASELINE configs[3])
# ---------------------------------------------------------------------------------------------------
# (the work gate counts K-tiles of 128 bytes per row = 32 floats: twice the K of the fp64 shapes)
@pytest.mark.parametrize("m,n,k,ta", [(38400, 256, 2048, "N"), (1280, 512, 32768, "N"), (1408 + 40, 256, 65536, "N"), (256, 256, 262144, "T"),
                                      (640 + 5, 256, 131072, "T"), (2048, 512, 16384, "T"), (4096, 2048, 2048, "N"), (2048, 4096, 4096, "T"),
                                      (8192, 2048, 16384, "T"), (16384, 1024, 2048, "N"),
                                      (1408 + 44, 256, 65536, "N")])
def test_streamk_gemm_f32_entrywise(ctx, m, n, k, ta):
    # contractions beyond 16384 are cut into chunks of 16384 that accumulate into C (beta applied by the first chunk only); a chunk goes
    # through the persistent kernel when its own shape passes the work gate, through the split-K kernel otherwise -- the numbers below
    # hold either way, the path counter is asserted for single-launch shapes only
    chunked = k > 16384

    d = _d()
    rng = np.random.default_rng(m + n + k + 1)
    A = rng.standard_normal((m, k)).astype(np.float32)
    B = rng.standard_normal((k, n)).astype(np.float32)
    C0 = rng.standard_normal((m, n)).astype(np.float32)
    Ad = d.cm_from_numpy(A if ta == "N" else A.T.copy())
    Bd = d.cm_from_numpy(B)
    lda = m if ta == "N" else k
    before = ctx.path_count(1)
    Cd = d.cm_from_numpy(C0)
    ctx.gemm(ta, "N", m, n, k, 1.5, Ad, lda, Bd, k, -0.5, Cd, m)
    if not chunked:
        assert ctx.path_count(1) == before + 1, "the fp32 stream-K kernel did not take this shape"
    r1 = d.cm_to_numpy(Cd)
    ref = 1.5 * (A.astype(np.float64) @ B.astype(np.float64)) - 0.5 * C0
    assert relerr(r1, ref) <= 4 * EPS32 * np.sqrt(k)            # fp32 fma chains of length k (measured class: ~1e-7 * sum |a b|)
    Cd2 = d.cm_from_numpy(C0)
    ctx.gemm(ta, "N", m, n, k, 1.5, Ad, lda, Bd, k, -0.5, Cd2, m)
    assert np.array_equal(r1, d.cm_to_numpy(Cd2))                # bitwise run to run
    Cd3 = d.cm_from_numpy(np.full((m, n), np.nan, dtype=np.float32))
    ctx.gemm(ta, "N", m, n, k, 1.0, Ad, lda, Bd, k, 0.0, Cd3, m)
    assert relerr(d.cm_to_numpy(Cd3), A.astype(np.float64) @ B.astype(np.float64)) <= 4 * EPS32 * np.sqrt(k)


@pytest.mark.parametrize("n,k", [(1024, 32768), (512, 131072), (768, 49152), (2048, 16384)])
def test_streamk_syrk_f32_upper_tiles(ctx, n, k):
    """fp32 Gram matrices through whatever route syrk takes BY DEFAULT: the persistent kernel's triangular tile map for contractions up to
    16384 (one fp32 fma chain per entry is only accurate that far), the split-K kernel with lower-triangle tiles skipped
    beyond -- the same contract either way: upper triangle to 4 eps sqrt(k), strictly lower triangle untouched."""
    single_launch = k <= 16384
    d = _d()
    rng = np.random.default_rng(n + k + 7)
    A = rng.standard_normal((k, n)).astype(np.float32)
    C0 = rng.standard_normal((n, n)).astype(np.float32)
    Cd = d.cm_from_numpy(C0)
    before = ctx.path_count(1)
    ctx.syrk("U", "T", n, k, 2.0, d.cm_from_numpy(A), k, 0.25, Cd, n)
    assert ctx.path_count(1) == before + (1 if single_launch else 0), "unexpected route for this contraction length"
    got = d.cm_to_numpy(Cd)
    ref = 2.0 * (A.astype(np.float64).T @ A.astype(np.float64)) + 0.25 * C0
    iu = np.triu_indices(n)
    assert np.abs(got[iu] - ref[iu]).max() <= 4 * EPS32 * np.sqrt(k) * np.abs(ref).max()
    il = np.tril_indices(n, -1)
    assert np.array_equal(got[il], C0[il])


# ---------------------------------------------------------------------------------------------------------------------
# CQRRPT's split QRCP: geqp3 in two halves, the solve in column ranges, the driver with and without the overlap
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,h", [(1280, 1024, 512), (640, 512, 256), (700, 300, 100)])
def test_geqp3_in_two_halves_is_geqp3(ctx, m, n, h):
    """rlhip_geqp3_steps (the first h steps) + geqp3 of the trailing block + its pivots applied to the finished rows and composed into jpvt
    == rlhip_geqp3: identical pivots, R and tau to rounding (the second half starts from recomputed column norms instead of down-dated ones)."""
    import torch

    d = _d()
    rng = np.random.default_rng(m + n + h)
    A0 = rng.standard_normal((m, n)) * np.logspace(0, -3, n)[rng.permutation(n)]
    f64 = torch.float64
    Af = d.cm_from_numpy(A0); Jf = torch.zeros(n, dtype=torch.int64, device="cuda"); tf = torch.zeros(n, dtype=f64, device="cuda")
    assert ctx.lib.rlhip_geqp3_f64(ctx.h, m, n, Af.data_ptr(), m, Jf.data_ptr(), tf.data_ptr()) == 0
    As = d.cm_from_numpy(A0); Js = torch.zeros(n, dtype=torch.int64, device="cuda"); ts = torch.zeros(n, dtype=f64, device="cuda")
    assert ctx.lib.rlhip_geqp3_steps_f64(ctx.h, m, n, h, As.data_ptr(), m, Js.data_ptr(), ts.data_ptr()) == 0
    part = d.cm_to_numpy(As)
    refh = d.cm_to_numpy(Af)
    assert np.array_equal(Js.cpu().numpy()[:h], Jf.cpu().numpy()[:h])                       # the leading pivots are final ...
    assert np.abs(np.triu(part[:h, :h]) - np.triu(refh[:h, :h])).max() <= 1e-13 * np.abs(refh).max()      # ... and so is the leading block of R
    J2 = torch.zeros(n - h, dtype=torch.int64, device="cuda")
    es = 8
    assert ctx.lib.rlhip_geqp3_f64(ctx.h, m - h, n - h, As.data_ptr() + (h + h * m) * es, m, J2.data_ptr(), ts.data_ptr() + h * es) == 0
    assert ctx.lib.rlhip_col_swap_f64(ctx.h, h, n - h, n - h, As.data_ptr() + h * m * es, m, J2.data_ptr()) == 0
    assert ctx.lib.rlhip_col_swap_i64(ctx.h, n - h, n - h, Js.data_ptr() + h * 8, J2.data_ptr()) == 0
    assert np.array_equal(Js.cpu().numpy(), Jf.cpu().numpy())
    got, ref = d.cm_to_numpy(As), d.cm_to_numpy(Af)
    k = min(m, n)
    assert np.abs(np.triu(got)[:k] - np.triu(ref)[:k]).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(ts.cpu().numpy()[:k] - tf.cpu().numpy()[:k]).max() <= 1e-11


def test_trsm_gather_range_pieces_are_the_whole_solve_bitwise(ctx):
    """rlhip_trsm_gather_range: the column ranges of a split solve are the same fused launch with other bounds -- two and three pieces give
    the bits of the one-piece solve; a pivot PREFIX is accepted (entries name any of nsrc source columns); a repeated entry is refused (-7)
    and a badly conditioned diagonal block returns 1, both before anything is written."""
    import torch

    d = _d()
    m, n = 33000, 1024
    rng = np.random.default_rng(77)
    U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
    Ud = d.cm_from_numpy(U)
    Src = d.cm_from_numpy(rng.standard_normal((m, n)))
    Jn = (rng.permutation(n) + 1).astype(np.int64)
    Jd = torch.from_numpy(Jn).cuda()
    whole = d.cm_zeros(m, n)
    ctx.trsm_gather(m, n, 0.5, Ud, n, Src, m, Jd, whole, m)
    fn = ctx.lib.rlhip_trsm_gather_range_f64

    def pieces(cuts):
        X = d.cm_from_numpy(np.full((m, n), np.nan))
        for c0, c1 in zip(cuts[:-1], cuts[1:]):
            assert fn(ctx.h, b"N", m, n, 0.5, Ud.data_ptr(), n, Src.data_ptr(), m, Jd.data_ptr(), X.data_ptr(), m, c0, c1) == 0
        return X
    assert torch.equal(pieces([0, 512, 1024]), whole)
    assert torch.equal(pieces([0, 256, 768, 1024]), whole)
    # a prefix alone: columns [0, 512) only -- the rest of the output stays as it was
    X = d.cm_from_numpy(np.full((m, n), -7.0))
    assert fn(ctx.h, b"N", m, n, 0.5, Ud.data_ptr(), n, Src.data_ptr(), m, Jd.data_ptr(), X.data_ptr(), m, 0, 512) == 0
    assert torch.equal(X[:512], whole[:512]) and bool((X[512:] == -7.0).all())
    Jbad = Jn.copy(); Jbad[300] = Jbad[20]
    X.fill_(-7.0)
    assert fn(ctx.h, b"N", m, n, 0.5, Ud.data_ptr(), n, Src.data_ptr(), m, torch.from_numpy(Jbad).cuda().data_ptr(), X.data_ptr(), m, 0, 512) == -7
    assert bool((X == -7.0).all())
    Ub = U.copy(); Ub[300:310, 300:310] = np.triu(np.ones((10, 10))) * 1e-9 + np.diag(np.full(10, 1e-9))      # block 1 fails the conditioning guard
    assert fn(ctx.h, b"N", m, n, 0.5, d.cm_from_numpy(Ub).data_ptr(), n, Src.data_ptr(), m, Jd.data_ptr(), X.data_ptr(), m, 0, 512) == 1
    assert bool((X == -7.0).all())
    assert fn(ctx.h, b"N", m, n, 0.5, Ud.data_ptr(), n, Src.data_ptr(), m, Jd.data_ptr(), X.data_ptr(), m, 100, 512) == -7     # bounds: multiples of 256


def test_cqrrpt_split_qrcp_equals_one_piece(ctx, monkeypatch):
    """CQRRPT with geqp3 of the sketch in two halves and the left half of the first solve beside the second one (rl_cqrrpt.hh; on for tall
    inputs on one rank) against the one-piece order (CQRRPT::split_qrcp = false): the same pivots and rank, R and Q to rounding, three
    fused out-of-place solve launches instead of two."""
    d = _d()
    m, n = 1 << 18, 512
    res = {}
    for knob in ("0", "1"):
        ctx.set_option("cqrrpt_split_qrcp", int(knob))
        A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(5, 0))
        # grade the columns a little: distinct column norms, no near-ties for the pivoting
        import torch
        A.mul_(torch.logspace(0, -2, n, dtype=torch.float64, device="cuda")[torch.randperm(n, generator=torch.Generator().manual_seed(3)).cuda()].unsqueeze(1))
        before = ctx.path_count(4)
        r = d.drv_cqrrpt(ctx, A, m, n, 1.25, 4, key=(9, 0))
        assert r["rc"] == 0 and r["rank"] == n
        res[knob] = (r["J"].cpu().numpy(), d.cm_to_numpy(r["R"]), A.clone(), ctx.path_count(4) - before)
    (J0, R0, Q0, l0), (J1, R1, Q1, l1) = res["0"], res["1"]
    assert l0 == 2 and l1 == 3
    assert np.array_equal(J0, J1)
    assert np.abs(R0 - R1).max() <= 1e-11 * np.abs(R0).max()
    import torch
    assert float((Q0 - Q1).abs().max()) <= 1e-11
    G = (Q1 @ Q1.T).cpu().numpy()
    assert np.abs(G - np.eye(n)).max() <= 1e-13 * n


@pytest.mark.parametrize("r", [200, 400])
def test_cqrrpt_split_qrcp_rank_deficient_inputs_fall_back(ctx, monkeypatch, r):
    """The split order on inputs it cannot finish: rank 200 < n / 2 (the leading block of R_sk already fails the rank criterion: no half solve is
    started, the second half of the factorization still runs on the side queue) and rank 400 (the left half of A_pre IS solved beside the second
    half, then the rank estimate says k < n and the reference's in-place order takes over while that solve may still be in flight).  Same rank,
    same leading pivots' span, same factorization quality as the one-piece order."""
    import torch

    d = _d()
    m, n = 1 << 18, 512
    g = torch.Generator(device="cuda").manual_seed(r)
    L = torch.randn((r, m), dtype=torch.float64, device="cuda", generator=g)          # column-major m x r
    Rt = torch.randn((n, r), dtype=torch.float64, device="cuda", generator=g) * torch.logspace(0, -2, r, dtype=torch.float64, device="cuda")
    A0 = Rt @ L                                                                          # (n, m) tensor = column-major m x n matrix of rank r
    res = {}
    for knob in ("0", "1"):
        ctx.set_option("cqrrpt_split_qrcp", int(knob))
        A = A0.clone()
        before = ctx.path_count(4)
        out = d.drv_cqrrpt(ctx, A, m, n, 1.25, 4, key=(3, 0))
        ctx.sync()
        res[knob] = (out["rc"], out["rank"], out["J"].cpu().numpy(), d.cm_to_numpy(out["R"]), A, ctx.path_count(4) - before)
    (rc0, k0, J0, R0, Q0, l0), (rc1, k1, J1, R1, Q1, l1) = res["0"], res["1"]
    assert rc0 == rc1 == 0
    assert k0 == k1 and 0 <= k0 - r <= 8                          # (the eps criterion may count a few noise directions: the same ones either way)
    assert l1 == l0 + (1 if r > n // 2 else 0)                     # the speculative half solve was launched only where the leading block allowed it
    k = k0
    # A P = Q R on the leading k columns of Q / rows of R, for both orders
    for (J, R, Q) in ((J0, R0, Q0), (J1, R1, Q1)):
        AP = A0[torch.from_numpy(J - 1).cuda()]                                          # rows of the (n, m) tensor = columns of A, permuted
        QR = torch.from_numpy(np.ascontiguousarray(R[:k, :].T)).cuda() @ Q[:k]           # (n, k) @ (k, m) = (Q R)^T
        assert float((AP - QR).norm() / A0.norm()) <= 1e-10
        G = (Q[:r] @ Q[:r].T).cpu().numpy()
        assert np.abs(G - np.eye(r)).max() <= 1e-9
    assert np.array_equal(J0[:r], J1[:r])                          # the same pivots on the numerically nonzero part


def test_cqrrpt_split_qrcp_fp32_and_ranged_solve_fp32(ctx, monkeypatch):
    """The fp32 instantiations of the split order: ranged fused solves bitwise the whole solve, and CQRRPT fp32 split == one piece."""
    import torch

    d = _d()
    f32 = torch.float32
    m, n = 40000, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    U = (torch.triu(torch.randn((n, n), dtype=f32, device="cuda", generator=g)) / (n ** 0.5) + 2 * torch.eye(n, dtype=f32, device="cuda")).T.contiguous()   # column-major upper
    Src = torch.randn((n, m), dtype=f32, device="cuda", generator=g)
    J = (torch.randperm(n, generator=torch.Generator().manual_seed(1)) + 1).cuda()
    whole = torch.zeros((n, m), dtype=f32, device="cuda")
    ctx.trsm_gather(m, n, 1.0, U, n, Src, m, J, whole, m)
    X = torch.full((n, m), float("nan"), dtype=f32, device="cuda")
    fn = ctx.lib.rlhip_trsm_gather_range_f32
    for c0, c1 in ((0, 256), (256, 512)):
        assert fn(ctx.h, b"N", m, n, 1.0, U.data_ptr(), n, Src.data_ptr(), m, J.data_ptr(), X.data_ptr(), m, c0, c1) == 0
    assert torch.equal(X, whole)
    m = 1 << 18
    res = {}
    for knob in ("0", "1"):
        ctx.set_option("cqrrpt_split_qrcp", int(knob))
        A = d.cm_empty(m, n, dtype=f32); ctx.fill_dense(A, m, n, key=(6, 0))
        A.mul_(torch.logspace(0, -1.5, n, dtype=f32, device="cuda")[torch.randperm(n, generator=torch.Generator().manual_seed(4)).cuda()].unsqueeze(1))
        before = ctx.path_count(4)
        r = d.drv_cqrrpt(ctx, A, m, n, 1.25, 4, key=(2, 0))
        assert r["rc"] == 0 and r["rank"] == n
        res[knob] = (r["J"].cpu().numpy(), d.cm_to_numpy(r["R"]).astype(np.float64), A.clone(), ctx.path_count(4) - before)
    (J0, R0, Q0, l0), (J1, R1, Q1, l1) = res["0"], res["1"]
    assert l1 == l0 + 1
    assert np.array_equal(J0, J1)
    assert np.abs(R0 - R1).max() <= 2e-4 * np.abs(R0).max()
    assert float((Q0 - Q1).abs().max()) <= 2e-4


def test_trsm_gather_full_size_is_deterministic_and_right_every_time(ctx):
    """1048576 x 1024 fp64 out-of-place solve with a pivot vector, six times: every run solves the system (residual at rounding level in every
    column) and returns the same bits.  Round 5: hipcc sank the loads that refill retired tiles to the end of a block, the counted wait in
    front of a rendezvous then let LDS-DMA pieces of a diagonal inverse fly, and SOME wavefronts of SOME runs produced a wrong second tile of a
    32-column sub-block (errors ~0.5 in 16 columns of ~64 rows): only repetition at full size shows that."""
    import torch

    d = _d()
    m, n = 1048576, 1024
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(3, 0))
    U = d.cm_empty(n, n); ctx.fill_dense(U, n, n, key=(2, 0))
    ctx.lib.rlhip_add_diag_f64(ctx.h, n, 40.0, U.data_ptr(), n)
    Um = torch.triu(U.T)
    ldw = m + 32
    W = torch.empty((n, ldw), dtype=torch.float64, device="cuda")
    Jp = torch.randperm(n, generator=torch.Generator().manual_seed(5)).cuda() + 1
    first = None
    for it in range(6):
        W.zero_()
        ctx.trsm_gather(m, n, 1.0, U, n, A, m, Jp, W, ldw); ctx.sync()
        X = W[:, :m]
        err = ((Um.T @ X) - A[Jp - 1]).abs().amax(dim=1)
        assert float(err.max()) <= 1e-11, (it, (err > 1e-11).nonzero().flatten()[:8].tolist())
        if first is None:
            first = X.clone()
        else:
            assert torch.equal(first, X), it
