"""-m gpu: the symmetric (Nystrom) path -- ExplicitSymLinOp, SYPS, SYRF, REVD2 -- against the numpy oracle; test matrix of
test/drivers/test_revd2.cc (exact-rank PSD input, rank doubling, Upper vs Lower storage with NaNs in the unused triangle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _d():
    from randlapack_amd import device

    return device


def _psd(m, rank, rng, decay=None):
    B = rng.standard_normal((m, rank))
    if decay is not None:
        B = B * decay
    return B @ B.T


def _recon(out, d):
    V, ev = d.cm_to_numpy(out["V"]), out["eigvals"].cpu().numpy()
    return V, ev, (V * ev) @ V.T


@pytest.mark.parametrize("m,rank,k0", [(200, 40, 40), (200, 40, 5), (500, 64, 64), (333, 50, 20)])
def test_revd2_vs_oracle(ctx, orc, m, rank, k0):
    d = _d()
    rng = np.random.default_rng(m + k0)
    A = _psd(m, rank, rng)
    out = d.drv_revd2(ctx, d.cm_from_numpy(A), m, k0, 1e-8, key=(1, 0))
    ref = orc.revd2(A, k0, 1e-8, key=(1, 0))
    assert out["rc"] == 0 and out["k"] == ref["k"] and out["next_ctr"] == ref["next_ctr"]    # same rank-doubling trajectory
    V, ev, R = _recon(out, d)
    assert np.linalg.norm(A - R) / np.linalg.norm(A) < 1e-11                 # ||A - V E V'|| / ||A|| (test_revd2.cc:165-170)
    np.testing.assert_allclose(ev, ref["eigvals"], rtol=0, atol=1e-9 * ref["eigvals"].max())
    kk = min(out["k"], rank)
    assert np.linalg.norm(V[:, :kk].T @ V[:, :kk] - np.eye(kk)) < 1e-10
    # the invariant subspace agrees with the oracle's (individual vectors only up to sign / clusters)
    Vo = ref["V"][:, :kk]
    assert np.linalg.norm(V[:, :kk] - Vo @ (Vo.T @ V[:, :kk])) < 1e-7


def test_revd2_uplo_ignores_the_other_triangle(ctx, orc):
    d = _d()
    rng = np.random.default_rng(9)
    m, rank = 160, 24
    A = _psd(m, rank, rng)
    Au, Al = np.triu(A), np.tril(A)
    Au[np.tril_indices(m, -1)] = np.nan                                       # test_revd2.cc:123-128
    Al[np.triu_indices(m, 1)] = np.nan
    ou = d.drv_revd2(ctx, d.cm_from_numpy(Au), m, rank, 1e-8, uplo="U", key=(2, 0))
    ol = d.drv_revd2(ctx, d.cm_from_numpy(Al), m, rank, 1e-8, uplo="L", key=(2, 0))
    Ru, Rl = _recon(ou, d)[2], _recon(ol, d)[2]
    assert not np.isnan(Ru).any() and not np.isnan(Rl).any()
    assert np.linalg.norm(Ru - Rl) < 1e-10 * np.linalg.norm(A)
    assert np.linalg.norm(Ru - A) < 1e-10 * np.linalg.norm(A)


def test_revd2_decaying_spectrum_tolerance_controls_rank(ctx, orc):
    d = _d()
    rng = np.random.default_rng(4)
    m = 400
    A = _psd(m, m, rng, decay=0.7 ** np.arange(m))                            # eigenvalues decay like 0.49^i
    loose = d.drv_revd2(ctx, d.cm_from_numpy(A), m, 4, 1e-2 * np.linalg.norm(A, 2), key=(3, 0), syps_passes=3)
    tight = d.drv_revd2(ctx, d.cm_from_numpy(A), m, 4, 1e-9 * np.linalg.norm(A, 2), key=(3, 0), syps_passes=3)
    assert loose["k"] < tight["k"] <= m
    for out, tol in ((loose, 1e-2), (tight, 1e-9)):
        R = _recon(out, d)[2]
        assert np.linalg.norm(A - R, 2) <= 50 * tol * np.linalg.norm(A, 2)     # the estimator's 5x slack and then some
    ref = orc.revd2(A, 4, 1e-9 * np.linalg.norm(A, 2), p=3, key=(3, 0))
    assert tight["k"] == ref["k"]


@pytest.mark.parametrize("orth_kind", [0, 1, 2])
def test_syrf_vs_oracle(ctx, orc, orth_kind):
    d = _d()
    rng = np.random.default_rng(5)
    m, rank = 300, 32
    A = _psd(m, rank, rng)
    out = d.drv_syrf(ctx, d.cm_from_numpy(A), m, rank, syps_passes=2, passes_per_stab=1, orth_kind=orth_kind, key=(6, 0))
    rc, Qo, nxt = orc.syrf(A, rank, 2, 1, orth_kind=orth_kind, key=(6, 0))
    Q = d.cm_to_numpy(out["Q"])
    assert out["rc"] == rc == 0 and out["next_ctr"] == nxt
    if orth_kind == 2:          # PLUL stabilises (unit lower-trapezoidal factor) but does not orthonormalise: compare with the oracle
        np.testing.assert_allclose(Q, Qo, rtol=0, atol=1e-8)
        return
    assert np.linalg.norm(Q.T @ Q - np.eye(rank)) < 1e-10
    assert np.linalg.norm(A - Q @ (Q.T @ A)) < 1e-9 * np.linalg.norm(A)
    assert np.linalg.norm(Q - Qo @ (Qo.T @ Q)) < 1e-8                          # same range as the oracle's basis


def test_revd2_bad_arguments(ctx):
    d = _d()
    from randlapack_amd._lib import RlhipError

    A = d.cm_from_numpy(np.eye(10))
    with pytest.raises(RlhipError, match="k=0"):
        d.drv_revd2(ctx, A, 10, 0, 1e-8)
    with pytest.raises(RlhipError, match="tol"):
        d.drv_revd2(ctx, A, 10, 2, -1.0)
    with pytest.raises(RlhipError):
        d.drv_revd2(ctx, A, 10, 2, 1e-8, uplo="X")


def test_revd2_at_scale(ctx):
    """m = 12000 PSD matrix of rank 300 generated on the device; REVD2 from k = 64 doubles to 512 and reconstructs it"""
    d = _d()
    import torch

    m, rank = 12000, 300
    g = torch.Generator(device="cuda:0").manual_seed(0)
    B = torch.randn((rank, m), dtype=torch.float64, device="cuda:0", generator=g)          # column-major m x rank
    A = B.T @ B                                                                             # symmetric: its own column-major image
    out = d.drv_revd2(ctx, A, m, 64, 1e-6 * float(torch.linalg.norm(A)), key=(1, 0))
    assert out["rc"] == 0 and out["k"] == 512
    V, ev = out["V"], out["eigvals"]                                                        # V: (k, m)
    R = (V.T * ev) @ V
    assert float(torch.linalg.norm(A - R) / torch.linalg.norm(A)) < 1e-10
    top = torch.linalg.eigvalsh(B @ B.T).flip(0)                                            # the 300 nonzero eigenvalues
    assert float(torch.max(torch.abs(ev[:rank] - top) / top)) < 1e-9
    assert float(ev[rank:].abs().max()) < 1e-8 * float(top[0])
