"""-m gpu: the reference's two DEVICE classes -- BQRRP_GPU (drivers/rl_bqrrp_gpu.hh:27-149) and CQRRPT_GPU
(drivers/rl_cqrrpt_gpu.hh:23-146) -- run through the C ABI and compared with the CPU oracle on a SHARED sketch, the shape of the
reference's own GPU-vs-CPU check (test/drivers/test_bqrrp_gpu.cu:231-249: J identical, ||d tau|| <= eps^0.75, ||d R||_F <= eps^0.6),
at the sizes of its GPU tests (:261-300: 5000 x 2800, b = 900, double; 2000 x 1000, b = 300, float)."""
import numpy as np
import pytest

from _gen import poly_mat

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
EPS32 = float(np.finfo(np.float32).eps)
CHOLQR, GEQRF = 0, 1                      # BQRRPGPUSubroutines::QRTall (rl_bqrrp_gpu.hh:44-46)
ORC_QR_TALL = {CHOLQR: 1, GEQRF: 2}       # the CPU class's enum (rl_bqrrp.hh:45-49): geqrt 0, cholqr 1, geqrf 2


def _d():
    from randlapack_amd import device

    return device


def _device_sketch(d, ctx, A_dev, m, n, dd, key):
    """A_sk = S A with a Gaussian S generated on the device (what a caller of BQRRP_GPU does before the call)."""
    import torch

    S = d.cm_empty(dd, m, dtype=A_dev.dtype, device=A_dev.device)
    ctx.fill_dense(S, dd, m, key=key)
    A_sk = d.cm_empty(dd, n, dtype=A_dev.dtype, device=A_dev.device)
    ctx.gemm("N", "N", dd, n, m, 1.0, S, dd, A_dev, m, 0.0, A_sk, dd)
    torch.cuda.synchronize()
    return A_sk


def _general_checks(orc, A, Aout, tau, J, atol):
    """test_BQRRP_general / error_check (test_bqrrp_gpu.cu:118-157,163-203): ||AP - QR|| / ||A||, max column residual, ||Q'Q - I|| / sqrt(n)"""
    m, n = A.shape
    mn = min(m, n)
    Q = orc.ungqr(Aout, tau)
    R = np.triu(Aout)[:mn]
    assert sorted(J.tolist()) == list(range(1, n + 1))
    AP = A[:, J - 1]
    E = AP - Q @ R
    assert np.linalg.norm(E) <= atol * np.linalg.norm(A)
    i = int(np.argmax(np.linalg.norm(E, axis=0)))
    assert np.linalg.norm(E[:, i]) <= atol * np.linalg.norm(AP[:, i])
    assert np.linalg.norm(Q.T @ Q - np.eye(mn)) <= atol * np.sqrt(n)


@pytest.mark.parametrize("qr_tall", [CHOLQR, GEQRF])
@pytest.mark.parametrize("m,n,b,dd", [(5000, 2800, 900, 900), (1000, 1000, 250, 250), (1500, 600, 64, 100)])
def test_bqrrp_gpu_vs_oracle_shared_sketch_f64(ctx, orc, qr_tall, m, n, b, dd):
    """BQRRP_GPU_070824 / BQRRP_GPU_qrf (test_bqrrp_gpu.cu:261-300) + the compare-with-CPU check (:205-249); the third case has a
    sampling dimension that is NOT a multiple of the block size (the device class takes d directly)."""
    d = _d()
    rng = np.random.default_rng(m + n + b)
    A = rng.standard_normal((m, n))
    Ad = d.cm_from_numpy(A)
    A_sk = _device_sketch(d, ctx, Ad, m, n, dd, key=(31, 0))
    sk = d.cm_to_numpy(A_sk)
    r = d.drv_bqrrp_gpu(ctx, Ad, m, n, A_sk, dd, b, qr_tall=qr_tall, timing=True)
    o = orc.bqrrp(A, b, dd / b, qrcp_wide=0, qr_tall=ORC_QR_TALL[qr_tall], apply_trans_q=0, sketch=sk)
    assert int(dd / b * b) == dd and r["rc"] == o["rc"] == 0
    assert r["rank"] == o["rank"] == min(m, n)
    Aout, tau, J = d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy()
    np.testing.assert_array_equal(J, o["J"])                                             # ||J - J_cpu|| = 0
    mn = min(m, n)
    assert np.linalg.norm(tau[:mn] - o["tau"][:mn]) <= EPS**0.75                         # col_nrm_tau
    assert np.linalg.norm(np.triu(Aout)[:mn] - np.triu(o["A"])[:mn]) <= EPS**0.6         # norm_R_diff (absolute, as the reference's)
    _general_checks(orc, A, Aout, tau, J, EPS**0.75)
    t = r["times_us"]                                                                    # rl_bqrrp_gpu.hh:829-834
    assert len(t) == 15 and sum(t[:14]) == t[14] and min(t[:13]) >= 0
    assert t[2] == t[4] == t[6] == 0                                                     # the reference's pointer-swap "copies": none here
    assert t[1] > 0 and t[3] > 0 and t[5] > 0 and t[9] > 0 and t[11] > 0 and t[12] > 0
    assert (t[8] > 0) == (qr_tall == CHOLQR) and (t[10] > 0) == (qr_tall == CHOLQR)       # preconditioning / reconstruction: cholqr only


def test_bqrrp_gpu_default_is_geqrf_and_sketch_is_consumed(ctx):
    """the object's default qr_tall is geqrf (rl_bqrrp_gpu.hh:84): -1 (keep the default) and 1 give bitwise the same factorization"""
    d = _d()
    rng = np.random.default_rng(3)
    m, n, b = 1200, 500, 100
    A = rng.standard_normal((m, n))
    outs = []
    for qt in (-1, GEQRF, CHOLQR):
        Ad = d.cm_from_numpy(A)
        A_sk = _device_sketch(d, ctx, Ad, m, n, b, key=(2, 0))
        sk0 = d.cm_to_numpy(A_sk).copy()
        r = d.drv_bqrrp_gpu(ctx, Ad, m, n, A_sk, b, b, qr_tall=qt)
        assert "times_us" not in r
        assert not np.array_equal(d.cm_to_numpy(A_sk), sk0)                              # A_sk is workspace of the call (:354-355)
        outs.append((d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy()))
    assert all(np.array_equal(a, b_) for a, b_ in zip(outs[0], outs[1]))
    assert np.array_equal(outs[0][2], outs[2][2]) and not np.array_equal(outs[0][0], outs[2][0])


@pytest.mark.parametrize("qr_tall", [CHOLQR, GEQRF])
def test_bqrrp_gpu_single_precision(ctx, orc, qr_tall):
    """BQRRP_GPU_single_precision (test_bqrrp_gpu.cu:282-300): 2000 x 1000 float, b = 300, the general check at eps^0.60; against the
    fp64 oracle on the widened sketch the leading pivots agree exactly and every block agrees as a set up to a few rounding-level swaps"""
    import torch

    d = _d()
    rng = np.random.default_rng(17)
    m, n, b = 2000, 1000, 300
    A = rng.standard_normal((m, n)).astype(np.float32)
    Ad = d.cm_from_numpy(A)
    assert Ad.dtype == torch.float32
    A_sk = _device_sketch(d, ctx, Ad, m, n, b, key=(9, 0))
    sk = d.cm_to_numpy(A_sk).astype(np.float64)
    r = d.drv_bqrrp_gpu(ctx, Ad, m, n, A_sk, b, b, qr_tall=qr_tall, timing=True)
    assert r["rc"] == 0 and r["rank"] == n and len(r["times_us"]) == 15
    Aout, tau, J = d.cm_to_numpy(Ad).astype(np.float64), r["tau"].cpu().numpy().astype(np.float64), r["J"].cpu().numpy()
    _general_checks(orc, A.astype(np.float64), Aout, tau, J, EPS32**0.60)
    o = orc.bqrrp(A.astype(np.float64), b, 1.0, qrcp_wide=0, qr_tall=ORC_QR_TALL[qr_tall], apply_trans_q=0, sketch=sk)
    np.testing.assert_array_equal(J[:16], o["J"][:16])
    overlap = [len(set(J[i:i + b].tolist()) & set(o["J"][i:i + b].tolist())) / len(J[i:i + b]) for i in range(0, n, b)]
    assert overlap[0] >= 0.97 and np.mean(overlap) >= 0.8
    # like for like: the float instantiation of the restatement on the float sketch -- the first block's pivots are identical
    o32 = orc.bqrrp(A, b, 1.0, qrcp_wide=0, qr_tall=ORC_QR_TALL[qr_tall], apply_trans_q=0, sketch=sk.astype(np.float32))
    assert o32["rc"] == 0 and o32["rank"] == n
    np.testing.assert_array_equal(J[:b], o32["J"][:b])
    ov32 = [len(set(J[i:i + b].tolist()) & set(o32["J"][i:i + b].tolist())) / len(J[i:i + b]) for i in range(0, n, b)]
    assert np.mean(ov32) >= 0.8


def test_bqrrp_gpu_rank_deficient_and_zero_inputs(ctx, orc):
    """low-rank input: the loop stops at the block whose R_sk diagonal drops below tol, rank = the block-rounded bound, same as the CPU
    class on the same sketch; an all-zero matrix returns rank 0 and leaves A zero (test_bqrrp_gpu.cu:177-182)"""
    d = _d()
    rng = np.random.default_rng(8)
    m, n, b, k = 900, 400, 64, 100
    A = poly_mat(m, n, k, rng, cond=1e3)                  # exact rank k (test_bqrrp.cc:188-207)
    for qt in (CHOLQR, GEQRF):
        Ad = d.cm_from_numpy(A)
        A_sk = _device_sketch(d, ctx, Ad, m, n, b, key=(4, 0))
        sk = d.cm_to_numpy(A_sk)
        r = d.drv_bqrrp_gpu(ctx, Ad, m, n, A_sk, b, b, qr_tall=qt)
        o = orc.bqrrp(A, b, 1.0, qrcp_wide=0, qr_tall=ORC_QR_TALL[qt], apply_trans_q=0, sketch=sk)
        assert r["rank"] == o["rank"] == 128
        J = r["J"].cpu().numpy()
        assert sorted(J.tolist()) == list(range(1, n + 1))
        np.testing.assert_array_equal(J[:64], o["J"][:64])
        Aout, tau = d.cm_to_numpy(Ad), r["tau"].cpu().numpy()
        kk = r["rank"]
        Q = orc.ungqr(Aout[:, :kk].copy(), tau[:kk])
        assert np.linalg.norm(A[:, J - 1] - Q @ np.triu(Aout)[:kk]) <= 1e-10 * np.linalg.norm(A)
    Z = d.cm_from_numpy(np.zeros((100, 40)))
    Zsk = d.cm_from_numpy(np.zeros((10, 40)))
    r = d.drv_bqrrp_gpu(ctx, Z, 100, 40, Zsk, 10, 10)
    assert r["rc"] == 0 and r["rank"] == 0 and not d.cm_to_numpy(Z).any()


def test_bqrrp_gpu_rejects_a_sampling_dimension_below_the_block(ctx):
    from randlapack_amd import _lib

    d = _d()
    A = d.cm_from_numpy(np.ones((50, 20)))
    sk = d.cm_from_numpy(np.ones((4, 20)))
    with pytest.raises(_lib.RlhipError):
        d.drv_bqrrp_gpu(ctx, A, 50, 20, sk, 4, 8)


@pytest.mark.parametrize("m,n,lda,ldr,cond", [(1000, 200, 1000, 200, 1.0), (1000, 200, 1007, 203, 1e6), (3000, 64, 3001, 64, 1.0)])
def test_cqrrpt_gpu_vs_oracle_shared_sketch(ctx, orc, m, n, lda, ldr, cond):
    """CQRRPT_GPU::call with HOST matrices and general lda / ldr (rl_cqrrpt_gpu.hh:117-127); the sketch the class factored is exported
    and handed to the oracle: rank and J identical, ||dR||_F <= eps^0.6 ||R||, the reference's CQRRPT residual checks
    (test_cqrrpt.cc:102-104), padding rows of A and R untouched, the 8-entry times vector, the RNG state advanced."""
    d = _d()
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n)) if cond == 1.0 else poly_mat(m, n, n, rng, cond=cond)
    eps_user = EPS**0.85
    r = d.drv_cqrrpt_gpu(ctx, A, 1.25, 4, eps=eps_user, key=(13, 0), want_sketch=True, timing=True, lda=lda, ldr=ldr)
    o = orc.cqrrpt(A, r["sketch"], eps_user)
    assert r["rc"] == o["rc"] == 0 and r["rank"] == o["rank"]
    k = r["rank"]
    np.testing.assert_array_equal(r["J"][:k], o["J"][:k])
    Q, R = r["Q"][:, :k], r["R"][:k]
    assert np.linalg.norm(R - o["R"][:k]) <= EPS**0.6 * np.linalg.norm(o["R"])
    assert np.linalg.norm(A[:, r["J"] - 1] - Q @ R) <= EPS**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q.T @ Q - np.eye(k)) <= EPS**0.75 * np.sqrt(n)
    if lda > m:
        assert np.isnan(r["A_buf"].reshape(n, lda)[:-1, m:]).all()                       # rows m..lda of A: untouched
    if ldr > n:
        assert (r["R_buf"].reshape(n, ldr)[:-1, n:] == 7.0).all()                        # rows n..ldr of R: untouched
    assert np.array_equal(np.tril(r["R"], -1), np.zeros((n, n)))
    t = r["times_us"]
    assert len(t) == 8 and sum(t[:7]) == t[7] and min(t[:6]) >= 0
    assert r["next_ctr"] != (0, 0, 0, 0)
    # the same call again from the same state reproduces the factorization bit for bit (deterministic kernels, same SASO)
    r2 = d.drv_cqrrpt_gpu(ctx, A, 1.25, 4, eps=eps_user, key=(13, 0), lda=lda, ldr=ldr)
    assert np.array_equal(r2["J"], r["J"]) and np.array_equal(r2["R"], r["R"]) and np.array_equal(r2["Q"], r["Q"])


def test_cqrrpt_gpu_hqrrp_option_and_f32(ctx, orc):
    """no_hqrrp = 0 routes the sketch's QRCP through hqrrp (rl_cqrrpt_gpu.hh:218-222); float instantiation"""
    d = _d()
    rng = np.random.default_rng(21)
    m, n = 2000, 128
    A = poly_mat(m, n, n, rng, cond=1e3)
    r = d.drv_cqrrpt_gpu(ctx, A, 1.25, 4, key=(1, 0), no_hqrrp=0)
    assert r["rc"] == 0 and r["rank"] == n and sorted(r["J"].tolist()) == list(range(1, n + 1))
    assert np.linalg.norm(A[:, r["J"] - 1] - r["Q"] @ r["R"]) <= EPS**0.75 * np.linalg.norm(A)
    A32 = A.astype(np.float32)
    r = d.drv_cqrrpt_gpu(ctx, A32, 1.25, 4, key=(1, 0))
    assert r["rc"] == 0 and r["Q"].dtype == np.float32 and r["rank"] == n
    Q, R = r["Q"].astype(np.float64), r["R"].astype(np.float64)
    assert np.linalg.norm(A32[:, r["J"] - 1] - Q @ R) <= EPS32**0.75 * np.linalg.norm(A32)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) <= EPS32**0.75 * np.sqrt(n)
