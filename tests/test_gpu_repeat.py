"""-m gpu: repeat-and-compare at FULL size for every kernel that stages operands by LDS-DMA (global -> LDS without a register, completion
counted by s_waitcnt): gemm_sk_kernel NN / TN / triangular map (fp64) and NN / TN (fp32), saso_apply_dma_kernel, gemm_tn_skinny_kernel.

Why: round 5's out-of-place solve RACED in SOME wavefronts of SOME runs at full size only (a counted wait that let DMA pieces fly across a
rendezvous) -- 630 tests were green.  A race of that kind shows as run-to-run differences, so every kernel of the family is launched >= 20
times on the same operands, INTERLEAVED with other kernels of the same context (different LDS images, different arena traffic between the
launches), and every result must equal the first bit for bit; the first result is also checked against an independent product on a sample.
The build-time counterpart is scripts/check_lds_dma_asm.py (no LDS-DMA request may be outstanding at an s_barrier)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPS = 20


def _d():
    from randlapack_amd import device as d

    return d


def _noise(ctx, it):
    """a few unrelated launches between two repetitions: a small solve, a small Gram matrix, a fill -- other LDS images, other scratch"""
    import torch

    d = _d()
    k = 256 + 32 * (it % 3)
    X = d.cm_empty(4096, k); ctx.fill_dense(X, 4096, k, key=(90 + it, 0))
    G = torch.zeros((k, k), dtype=torch.float64, device="cuda")
    ctx.syrk("U", "T", k, 4096, 1.0, X, 4096, 0.0, G, k)
    ctx.lib.rlhip_add_diag_f64(ctx.h, k, 1.0e4, G.data_ptr(), k)
    ctx.trsm(4096, k, 1.0, G, k, X, 4096)


def _check_sample(got_cols, ref_cols, tol):
    err = float((got_cols - ref_cols).abs().max() / ref_cols.abs().max())
    assert err <= tol, err


def test_gemm_sk_nn_tn_fp64_full_size_same_bits_every_time(ctx):
    """C2's two passes (drivers/rl_rsvd.hh -> comps/rl_rf.hh:123, comps/rl_qb.hh:218): Y = A Omega and B^T = A^T Q, 200000 x 20000 x 256 fp64"""
    import torch

    d = _d()
    m, n, k = 200000, 20000, 256
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7, 0))
    Om = d.cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(1, 0))
    Y = d.cm_empty(m, k)
    Q = d.cm_empty(m, k); ctx.fill_dense(Q, m, k, key=(2, 0))
    BT = d.cm_empty(n, k)
    sk0 = ctx.path_count(0)
    firstY = firstB = None
    for it in range(REPS):
        Y.fill_(float("nan")); BT.fill_(float("nan"))
        ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)
        _noise(ctx, it)
        ctx.gemm("T", "N", n, k, m, 1.0, A, m, Q, m, 0.0, BT, n)
        ctx.sync()
        if firstY is None:
            firstY, firstB = Y.clone(), BT.clone()
            rows = torch.arange(0, m, 997, device="cuda")
            _check_sample(Y[:, rows], (A[:, rows].T @ Om.T).T, 1e-12)                     # column-major tensors: X[j] is column j
            cols = torch.arange(0, n, 97, device="cuda")
            _check_sample(BT[:, cols], Q @ A[cols].T, 1e-12)
        else:
            assert torch.equal(Y, firstY), f"Y differs in repetition {it}: {int((Y != firstY).sum())} entries"
            assert torch.equal(BT, firstB), f"B^T differs in repetition {it}: {int((BT != firstB).sum())} entries"
    assert ctx.path_count(0) - sk0 >= 2 * REPS, "the persistent stream-K kernel did not serve these products"


def test_gemm_sk_tri_fp64_full_size_same_bits_every_time(ctx):
    """C3's Gram matrix (drivers/rl_cqrrpt.hh:300-345 -> syrk): 1048576 x 1024 fp64 with CQRRPT's padded leading dimension"""
    import torch

    d = _d()
    m, n = 1048576, 1024
    ld = m + 32
    A = torch.empty((n, ld), dtype=torch.float64, device="cuda")
    ctx.fill_dense(A, ld, n, key=(3, 0))
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    first = None
    for it in range(REPS):
        G.fill_(float("nan"))
        ctx.lib.rlhip_laset_f64(ctx.h, b"G", n, n, 0.0, 0.0, G.data_ptr(), n)
        ctx.syrk("U", "T", n, m, 1.0, A, ld, 0.0, G, n)
        _noise(ctx, it)
        ctx.sync()
        if first is None:
            first = G.clone()
            At = A[:, :m]                                                                   # row j = column j of the matrix
            ref = At[:64] @ At.T                                                            # rows 0..63 of A^T A
            got = G.T[:64]                                                                  # (column-major tensor: G[j][i] = entry (i, j))
            up = torch.arange(n, device="cuda").unsqueeze(0) >= torch.arange(64, device="cuda").unsqueeze(1)
            _check_sample(got * up, ref * up, 1e-12)
        else:
            assert torch.equal(G, first), f"Gram matrix differs in repetition {it}: {int((G != first).sum())} entries"


def test_gemm_sk_nn_tn_fp32_full_size_same_bits_every_time(ctx):
    """one chunk of C4's compact-WY apply (drivers/rl_bqrrp.hh:535-547): W = V^T C (2048 x 16384 x 65536) and C -= V W (65536 x 16384 x 2048), fp32"""
    import torch

    d = _d()
    m, nc, b = 65536, 16384, 2048
    V = d.cm_empty(m, b, dtype=torch.float32); ctx.fill_dense(V, m, b, key=(11, 0))
    Cm = d.cm_empty(m, nc, dtype=torch.float32); ctx.fill_dense(Cm, m, nc, key=(12, 0))
    W = d.cm_empty(b, nc, dtype=torch.float32)
    Cw = torch.empty_like(Cm)
    sk0 = ctx.path_count(1)
    firstW = firstC = None
    for it in range(REPS):
        W.fill_(float("nan")); Cw.copy_(Cm)
        ctx.gemm("T", "N", b, nc, m, 1.0, V, m, Cw, m, 0.0, W, b)
        _noise(ctx, it)
        ctx.gemm("N", "N", m, nc, b, -1.0 / 256.0, V, m, W, b, 1.0, Cw, m)
        ctx.sync()
        if firstW is None:
            firstW, firstC = W.clone(), Cw.clone()
            cols = torch.arange(0, nc, 257, device="cuda")
            refW = (V.double() @ Cm[cols].double().T).T
            _check_sample(W[cols].double(), refW, 1e-4)
            refC = Cm[cols].double() - (refW @ V.double()) / 256.0
            _check_sample(Cw[cols].double(), refC, 1e-4)
        else:
            assert torch.equal(W, firstW), f"W differs in repetition {it}: {int((W != firstW).sum())} entries"
            assert torch.equal(Cw, firstC), f"C differs in repetition {it}: {int((Cw != firstC).sum())} entries"
    assert ctx.path_count(1) - sk0 >= 2 * REPS, "the persistent fp32 stream-K kernel did not serve these products"


def test_saso_apply_dma_full_size_same_bits_every_time(ctx):
    """C3's sketch (drivers/rl_cqrrpt.hh:214-222): S (1280 x 1048576, 4 nonzeros per column) times A (1048576 x 1024 fp64)"""
    import torch

    d = _d()
    m, n, dd, nnz = 1048576, 1024, 1280, 4
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(5, 0))
    B = d.cm_empty(dd, n)
    u32 = lambda t: (C.c_uint32 * len(t))(*t)
    S = C.c_void_p(); nxt = (C.c_uint32 * 4)()
    assert ctx.lib.rlhip_saso_create_mode(ctx.h, dd, m, nnz, 1, u32((0, 0, 0, 0)), u32((7, 0)), nxt, C.byref(S)) == 0
    dma0 = ctx.path_count(14)
    first = None
    for it in range(REPS):
        B.fill_(float("nan"))
        assert ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, 1.0, A.data_ptr(), m, 0.0, B.data_ptr(), dd) == 0
        _noise(ctx, it)
        ctx.sync()
        if first is None:
            first = B.clone()
            # independent check on a row sample of S: densify 64 columns' worth through the dense operator of a SMALL twin is not possible (the
            # operator is a function of (d, m)); instead check linearity on the full operator: S (2 A) = 2 S A bit for bit is trivial, so compare
            # with the row-range entry point, which walks the same lists through a different kernel route
            B2 = d.cm_empty(dd, n)
            half = m // 2
            assert ctx.lib.rlhip_saso_apply_rows_f64(ctx.h, S, n, 1.0, A.data_ptr(), m, 0, half, 0.0, B2.data_ptr(), dd) == 0
            assert ctx.lib.rlhip_saso_apply_rows_f64(ctx.h, S, n, 1.0, A[:, half:].contiguous().data_ptr(), m - half, half, m - half, 1.0, B2.data_ptr(), dd) == 0
            ctx.sync()
            _check_sample(B2, first, 1e-12)
        else:
            assert torch.equal(B, first), f"sketch differs in repetition {it}: {int((B != first).sum())} entries"
    assert ctx.path_count(14) - dma0 >= REPS, "the LDS-DMA apply kernel did not serve these sketches"
    ctx.lib.rlhip_saso_destroy(ctx.h, S)


@pytest.mark.parametrize("ka,kb", [(32, 32), (64, 32), (48, 48)])
def test_gemm_tn_skinny_full_size_same_bits_every_time(ctx, ka, kb):
    """ABRIK's panel products at C5 (drivers/rl_abrik.hh:333-420): X^T Y with X 200000 x ka, Y 200000 x kb"""
    import torch

    d = _d()
    m = 200000
    X = d.cm_empty(m, ka); ctx.fill_dense(X, m, ka, key=(21, 0))
    Y = d.cm_empty(m, kb); ctx.fill_dense(Y, m, kb, key=(22, 0))
    G = d.cm_empty(ka, kb)
    first = None
    for it in range(REPS):
        G.fill_(float("nan"))
        ctx.gemm("T", "N", ka, kb, m, 1.0, X, m, Y, m, 0.0, G, ka)
        _noise(ctx, it)
        ctx.sync()
        if first is None:
            first = G.clone()
            _check_sample(G, Y @ X.T, 1e-12)
        else:
            assert torch.equal(G, first), f"X^T Y differs in repetition {it}"
