"""-m gpu: the device test-matrix generators (include/RandLAPACK_amd/rl_gen.hh) against the numpy restatement of the reference's
testing/rl_gen.hh.  Both sides draw from the same Philox stream, and both orthonormalise with LAPACK-convention Householder QR, so
the matrices agree entrywise; the reference's own check (test/misc/test_gen.cc) is on the spectrum."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _d():
    from randlapack_amd import device

    return device


@pytest.mark.parametrize("m_type,kw", [
    ("polynomial", dict(cond_num=1e6, exponent=2.0)), ("exponential", dict(cond_num=1e5)), ("step", dict(cond_num=1e4)),
    ("bad_cholqr", dict(cond_num=1e3)), ("gaussian", {}), ("spiked", dict(scaling=7.0)), ("adverserial", dict(scaling=1e-5)),
    ("kahan", dict(theta=1.2, perturb=1e3))])
def test_mat_gen_matches_oracle(ctx, orc, m_type, kw):
    d = _d()
    m, n = (300, 300) if m_type == "kahan" else (500, 120)
    rank = 80 if m_type in ("polynomial", "exponential", "step", "bad_cholqr") else None
    out = d.drv_mat_gen(ctx, m_type, m, n, rank=rank, key=(4, 0), **kw)
    ref, nxt = orc.mat_gen(m_type, m, n, rank=rank, key=(4, 0), **kw)
    A = d.cm_to_numpy(out["A"])
    assert out["next_ctr"] == nxt                                       # the state is consumed in the reference's order
    np.testing.assert_allclose(A, ref, rtol=0, atol=2e-12 * max(1.0, np.abs(ref).max()))
    if rank:
        s = np.linalg.svd(A, compute_uv=False)
        want = np.linalg.svd(ref, compute_uv=False)
        np.testing.assert_allclose(s[:rank], want[:rank], rtol=0, atol=1e-13)
        assert s[rank:].max() < 1e-13


def test_mat_gen_diag_f32_rank_check_and_bad_type(ctx, orc):
    d = _d()
    import torch
    from randlapack_amd._lib import RlhipError

    out = d.drv_mat_gen(ctx, "polynomial", 50, 40, rank=30, cond_num=1e3, exponent=2.0, diag=True)
    D = d.cm_to_numpy(out["A"])
    assert D.shape == (30, 30) and np.array_equal(np.diag(D), orc.gen_poly_singvals(30, 0.1, 1e3, 2.0)) and np.count_nonzero(D) == 30
    assert out["next_ctr"] == (0, 0, 0, 0)
    o32 = d.drv_mat_gen(ctx, "exponential", 200, 64, rank=64, cond_num=100.0, dtype=torch.float32, key=(1, 0))
    s = np.linalg.svd(d.cm_to_numpy(o32["A"]).astype(np.float64), compute_uv=False)
    np.testing.assert_allclose(s, orc.gen_exp_singvals(64, 100.0), rtol=0, atol=2e-5)
    # spiked with an enormous spike is numerically rank deficient; check_true_rank reports what the SVD sees (rl_util.hh:426-448)
    osp = d.drv_mat_gen(ctx, "spiked", 300, 40, scaling=1e20, check_true_rank=True)
    A = d.cm_to_numpy(osp["A"])
    sv = np.linalg.svd(A, compute_uv=False)
    small = np.nonzero(sv <= 5 * np.finfo(float).eps * sv[0])[0]
    assert osp["rank"] == (small[0] - 1 if small.size else 40)
    with pytest.raises(RlhipError):
        d.drv_mat_gen(ctx, 8, 10, 10)


def test_mat_gen_at_scale_spectrum(ctx):
    """C1-style input generated in HBM at a size the host route would not want: 65536 x 2048, rank 512"""
    d = _d()
    m, n, k = 65536, 2048, 512
    out = d.drv_mat_gen(ctx, "polynomial", m, n, rank=k, cond_num=1e6, exponent=2.0)
    import torch

    A = out["A"]                                     # (n, m) row-major == m x n column-major
    G = (A @ A.T).cpu().numpy()                      # n x n Gram matrix = V S^2 V^T
    ev = np.sort(np.linalg.eigvalsh(G))[::-1]
    from oracle import gen_poly_singvals
    want = gen_poly_singvals(k, 0.1, 1e6, 2.0)
    np.testing.assert_allclose(np.sqrt(np.abs(ev[:64])), want[:64], rtol=1e-9)    # through the Gram matrix: eps * n * few
    assert abs(ev[k:]).max() < 1e-12
