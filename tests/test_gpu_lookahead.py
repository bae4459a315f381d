"""-m gpu: BQRRP's look-ahead (rl_bqrrp.hh::detail::bqrrp_factor: sketch down-date + next QRCP of the sketch on a side queue beside the tail of
the compact-WY apply) is the DEFAULT path of the C4 headline (65536^2 fp32), where no oracle can run.  BQRRP::lookahead_min_elems is a member,
so these tests force the side-queue path at oracle sizes and hold it to the same bar as the serial loop (reference: drivers/rl_bqrrp.hh:306-661,
tolerance precedent test/drivers/test_bqrrp_gpu.cu:231-249): pivots EXACTLY the oracle's and the serial loop's, R to eps^0.6, tau to 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
EPS32 = np.finfo(np.float32).eps
NEVER = 1 << 62


def _d():
    from randlapack_amd import device as d

    return d


def _graded(m, n, rng, decades):
    """Gaussian columns with well-separated scales in a random order: no pivot decision is a rounding-level near-tie"""
    return rng.standard_normal((m, n)) * np.logspace(0, -decades, n)[rng.permutation(n)]


@pytest.mark.parametrize("qrcp_wide", [0, 1])          # luqr, geqp3
@pytest.mark.parametrize("qr_tall", [1, 2])            # cholqr, geqrf
@pytest.mark.parametrize("apply_q", [1, 0])            # gemqrt, ormqr
def test_bqrrp_lookahead_f64_pivots_exact_vs_oracle_and_serial(ctx, orc, qrcp_wide, qr_tall, apply_q):
    d = _d()
    big = (qrcp_wide, qr_tall, apply_q) == (0, 1, 1)                   # the C4 triple also at 4096 x 4096, b = 512
    m, n, b = (4096, 4096, 512) if big else (2048, 1536, 256)
    rng = np.random.default_rng(7 * m + n + 100 * qrcp_wide + 10 * qr_tall + apply_q)
    A = _graded(m, n, rng, 4.0)
    res = {}
    for name, thresh in (("lookahead", 0), ("serial", NEVER)):
        Ad = d.cm_from_numpy(A)
        before = ctx.path_count(12)
        with ctx.options(bqrrp_lookahead_min_elems=thresh):
            r = d.drv_bqrrp(ctx, Ad, m, n, b, 1.0, want_sketch=True, key=(21, 0), qrcp_wide=qrcp_wide, qr_tall=qr_tall, apply_trans_q=apply_q)
        assert r["rc"] == 0
        res[name] = (d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy(), r["rank"], d.cm_to_numpy(r["sketch"]), ctx.path_count(12) - before)
    Fl, tl, Jl, kl, sk, nl = res["lookahead"]
    Fs, ts, Js, ks, sk2, ns = res["serial"]
    assert ns == 0 and nl == min(m, n) // b - 1, f"side-queue iterations: forced {nl}, serial {ns}"
    assert np.array_equal(sk, sk2), f"the two calls formed different sketches: max {np.abs(sk - sk2).max():.3e} at {int((sk != sk2).sum())} entries"
    o = orc.bqrrp(A, b, 1.0, qrcp_wide=qrcp_wide, qr_tall=qr_tall, apply_trans_q=apply_q, sketch=sk)
    assert o["rc"] == 0 and kl == ks == o["rank"]
    np.testing.assert_array_equal(Jl, Js)                               # the two orders: the same pivots, bit for bit
    np.testing.assert_array_equal(Jl, o["J"])                           # ... and the oracle's
    mn = min(m, n)
    Ro = np.triu(o["A"])[:mn]
    for F, tau in ((Fl, tl), (Fs, ts)):
        assert np.linalg.norm(np.triu(F)[:mn] - Ro) <= EPS**0.6 * np.linalg.norm(Ro)
        np.testing.assert_allclose(tau, o["tau"], atol=1e-9, rtol=0)
    # the two orders differ only in which GEMM kernel runs the tail of the apply: factors equal to rounding, far inside the oracle tolerance
    assert np.abs(Fl - Fs).max() <= 1e-11 * np.abs(Fs).max()
    Q = orc.ungqr(Fl, tl)
    assert np.linalg.norm(A[:, Jl - 1] - Q @ np.triu(Fl)[:mn]) <= EPS**0.75 * np.linalg.norm(A)
    assert np.linalg.norm(Q.T @ Q - np.eye(mn)) <= EPS**0.75 * np.sqrt(mn)


@pytest.mark.parametrize("qr_tall", [0, 1])            # BQRRPGPUSubroutines: cholqr, geqrf
def test_bqrrp_gpu_class_lookahead_equals_serial_and_oracle(ctx, orc, qr_tall):
    """the reference's device class (rl_bqrrp_gpu.hh) runs the same loop: sketch from the caller, look-ahead forced"""
    d = _d()
    m, n, b = 2048, 2048, 256
    rng = np.random.default_rng(31 + qr_tall)
    A = _graded(m, n, rng, 4.0)
    S = rng.standard_normal((b, m))
    sk = S @ A
    res = {}
    for name, thresh in (("lookahead", 0), ("serial", NEVER)):
        Ad, Sk = d.cm_from_numpy(A), d.cm_from_numpy(sk)
        before = ctx.path_count(12)
        with ctx.options(bqrrp_lookahead_min_elems=thresh):
            r = d.drv_bqrrp_gpu(ctx, Ad, m, n, Sk, b, b, qr_tall=qr_tall)
        res[name] = (d.cm_to_numpy(Ad), r["tau"].cpu().numpy(), r["J"].cpu().numpy(), r["rank"], ctx.path_count(12) - before)
    assert res["lookahead"][4] == n // b - 1 and res["serial"][4] == 0
    o = orc.bqrrp(A, b, 1.0, qrcp_wide=0, qr_tall=1 if qr_tall == 0 else 2, apply_trans_q=0, sketch=sk)
    for F, tau, J, k, _ in res.values():
        assert k == o["rank"]
        np.testing.assert_array_equal(J, o["J"])
        Ro = np.triu(o["A"])[:n]
        assert np.linalg.norm(np.triu(F)[:n] - Ro) <= EPS**0.6 * np.linalg.norm(Ro)
        np.testing.assert_allclose(tau, o["tau"], atol=1e-9, rtol=0)


@pytest.mark.parametrize("lookahead", [0, NEVER])
def test_bqrrp_f32_graded_columns_all_blocks_pivots_exact(ctx, orc, lookahead):
    """fp32 (C4's dtype) on columns separated far beyond float rounding: EVERY block's pivots equal the float leg of the oracle and the double
    oracle on the same float matrix and the same sketch -- not only the first block (the Gaussian full-size cases can only be held to
    block-wise overlap: their pivot decisions include float-level near-ties)."""
    import torch

    d = _d()
    m, n, b = 3072, 1024, 128
    rng = np.random.default_rng(5)
    A = _graded(m, n, rng, 4.5).astype(np.float32)
    Ad = d.cm_from_numpy(A.astype(np.float64)).to(torch.float32)
    before = ctx.path_count(12)
    with ctx.options(bqrrp_lookahead_min_elems=lookahead):
        r = d.drv_bqrrp(ctx, Ad, m, n, b, 1.0, want_sketch=True, key=(3, 0), qrcp_wide=0, qr_tall=1, apply_trans_q=1)
    assert (ctx.path_count(12) - before) == (n // b - 1 if lookahead == 0 else 0)
    sk = d.cm_to_numpy(r["sketch"])
    assert sk.dtype == np.float32
    o32 = orc.bqrrp(A, b, 1.0, qrcp_wide=0, qr_tall=1, apply_trans_q=1, sketch=sk)
    o64 = orc.bqrrp(A.astype(np.float64), b, 1.0, qrcp_wide=0, qr_tall=1, apply_trans_q=1, sketch=sk.astype(np.float64), tol=float(EPS32))
    J = r["J"].cpu().numpy()
    assert r["rank"] == o32["rank"] == o64["rank"] == n
    np.testing.assert_array_equal(o32["J"], o64["J"])                   # the input is what it claims: separated beyond float rounding
    np.testing.assert_array_equal(J, o32["J"])                          # all 8 blocks, exact
    F = d.cm_to_numpy(Ad).astype(np.float64)
    Ro = np.triu(o64["A"])[:n]
    assert np.linalg.norm(np.triu(F)[:n] - Ro) <= EPS32**0.6 * np.linalg.norm(Ro)
    np.testing.assert_allclose(r["tau"].cpu().numpy(), o64["tau"], atol=2e-4, rtol=0)


def test_repeated_calls_with_a_crowded_output_pool_form_the_same_sketch(ctx):
    """Regression (round 5): with the context's caching pool full of other sizes the sketching operator of every BQRRP call used to be
    hipMalloc'ed and hipFree'd around its product, and the look-ahead created and destroyed a side context (64 MiB arena, uncached exchange
    buffer) per call -- device memory unmapped and remapped between launches, after which the SECOND call's sketch came out wrong in the
    contributions of its first 128 rows.  Now the pool keeps every block tracked (least recently freed idle block evicted when
    the table is full) and the side queue is the context's cached one: the same call twice forms the same sketch, bit for bit, and equals
    S A computed independently."""
    import ctypes as C

    d = _d()
    ptrs = []
    for i in range(300):                                            # crowd the pool: 300 idle blocks of distinct sizes
        p = C.c_void_p()
        assert ctx.lib.rlhip_malloc(ctx.h, C.byref(p), 4096 * (i + 1) + 256) == 0
        ptrs.append(p)
    for p in ptrs:
        assert ctx.lib.rlhip_free(ctx.h, p) == 0
    m, n, b = 2048, 1536, 256
    rng = np.random.default_rng(3)
    A = _graded(m, n, rng, 4.0)
    S = d.cm_empty(b, m); ctx.fill_dense(S, b, m, key=(21, 0))
    ref = d.cm_to_numpy(S) @ A
    sks = []
    for thresh in (0, NEVER, 0, NEVER):
        Ad = d.cm_from_numpy(A)
        with ctx.options(bqrrp_lookahead_min_elems=thresh):
            r = d.drv_bqrrp(ctx, Ad, m, n, b, 1.0, want_sketch=True, key=(21, 0), qrcp_wide=1, qr_tall=1, apply_trans_q=1)
        sks.append(d.cm_to_numpy(r["sketch"]))
    for sk in sks:
        assert np.abs(sk - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.array_equal(sk, sks[0])
    ctx.lib.rlhip_trim(ctx.h)
