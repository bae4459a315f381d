"""One kernel at the shape its roofline claim is made for, launched three times -- the workload side of scripts/pmc_all.py (which runs
this file under `rocprofv3 --kernel-trace --pmc ...`, one counter group per process, and writes profiles-ready json).

  python scripts/pmc_workloads.py <name>        names: see WORKLOADS at the bottom (`--list` prints them with their kernel filters)

Every workload states the kernel-name substring that identifies its dispatches and the ALGORITHMIC bytes / flops of one launch, so the
counter totals can be put beside them."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REPS = 2


def _ctx():
    import torch  # noqa: F401
    from randlapack_amd import device as d

    return d, d.Context(0)


def gemm_sk_nn():
    d, ctx = _ctx()
    m, n, k = 200000, 20000, 256
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7, 0))
    Om = d.cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(0, 0))
    Y = d.cm_empty(m, k)
    for _ in range(REPS):
        ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)
    ctx.sync()


def gemm_sk_tn():
    d, ctx = _ctx()
    m, n, k = 200000, 20000, 256
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7, 0))
    Q = d.cm_empty(m, k); ctx.fill_dense(Q, m, k, key=(1, 0))
    BT = d.cm_empty(n, k)
    for _ in range(REPS):
        ctx.gemm("T", "N", n, k, m, 1.0, A, m, Q, m, 0.0, BT, n)
    ctx.sync()


def gemm_sk_tri():
    d, ctx = _ctx()
    import torch

    m, n = 1048576, 1024
    ldw = m + 32                          # the padded leading dimension CQRRPT gives its scratch matrix (a power-of-two column stride camps on channels)
    A = torch.empty((n, ldw), dtype=torch.float64, device="cuda")
    T0 = d.cm_empty(m, 1)
    for j in range(0, n, 64):             # (fill through a packed block: fill_dense writes ld = rows)
        blk = d.cm_empty(m, 64); ctx.fill_dense(blk, m, 64, key=(3, j)); ctx.sync()
        A[j:j + 64, :m] = blk
    del T0
    G = d.cm_zeros(n, n)
    for _ in range(REPS):
        ctx.syrk("U", "T", n, m, 1.0, A, ldw, 0.0, G, n)
    ctx.sync()


def _trsm(oop):
    d, ctx = _ctx()
    import torch

    m, n = 1048576, 1024
    B = d.cm_empty(m, n); ctx.fill_dense(B, m, n, key=(1, 0))
    U = d.cm_empty(n, n); ctx.fill_dense(U, n, n, key=(2, 0))
    ctx.lib.rlhip_add_diag_f64(ctx.h, n, C.c_double(40.0), U.data_ptr(), n)
    if oop:
        W = d.cm_empty(m, n)
        J = torch.arange(n, 0, -1, dtype=torch.int64, device="cuda")
        for _ in range(REPS):
            ctx.trsm_gather(m, n, 1.0, U, n, B, m, J, W, m)
    else:
        for _ in range(REPS):
            ctx.fill_dense(B, m, n, key=(1, 0))
            ctx.trsm(m, n, 1.0, U, n, B, m)
    ctx.sync()


def trsm_fused_f32_192():
    """BQRRP's Cholesky-QR panel solve at a row count that takes the 192-row workgroups (49152 x 2048 fp32, in place)"""
    d, ctx = _ctx()
    import torch

    m, n = 49152, 2048
    B = d.cm_empty(m, n, dtype=torch.float32); ctx.fill_dense(B, m, n, key=(1, 0))
    U = d.cm_empty(n, n, dtype=torch.float32); ctx.fill_dense(U, n, n, key=(2, 0))
    ctx.lib.rlhip_add_diag_f32(ctx.h, n, C.c_float(60.0), U.data_ptr(), n)
    for _ in range(REPS):
        ctx.trsm(m, n, 1.0, U, n, B, m)
    ctx.sync()


def trsm_fused():
    _trsm(False)


def trsm_fused_oop():
    _trsm(True)


def saso_apply():
    d, ctx = _ctx()
    m, n, dd, nnz = 1048576, 1024, 1280, 4
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(3, 0))
    B = d.cm_zeros(dd, n)
    u32 = lambda t: (C.c_uint32 * len(t))(*t)
    S = C.c_void_p(); nxt = (C.c_uint32 * 4)()
    assert ctx.lib.rlhip_saso_create_mode(ctx.h, dd, m, nnz, 1, u32((0, 0, 0, 0)), u32((7, 0)), nxt, C.byref(S)) == 0
    for _ in range(REPS):
        assert ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, C.c_double(1.0), A.data_ptr(), m, C.c_double(0.0), B.data_ptr(), dd) == 0
    ctx.sync()
    ctx.lib.rlhip_saso_destroy(ctx.h, S)


def _gemm_f32(ta):
    d, ctx = _ctx()
    import torch

    # the compact-WY apply of BQRRP at C4 (first iteration): W (2048 x 63488) = V^T C over 65536 rows in chunks of 16384 / C -= V W
    rows, b, rest = 65536, 2048, 16384
    V = d.cm_empty(rows, b, dtype=torch.float32); ctx.fill_dense(V, rows, b, key=(1, 0))
    Cm = d.cm_empty(rows, rest, dtype=torch.float32); ctx.fill_dense(Cm, rows, rest, key=(2, 0))
    W = d.cm_zeros(b, rest, dtype=torch.float32)
    for _ in range(REPS):
        if ta == "T":
            ctx.gemm("T", "N", b, rest, 16384, 1.0, V, rows, Cm, rows, 0.0, W, b)
        else:
            ctx.gemm("N", "N", rows, rest, b, -1.0, V, rows, W, b, 1.0, Cm, rows)
    ctx.sync()


def gemm_f32_tn():
    _gemm_f32("T")


def gemm_f32_nn():
    _gemm_f32("N")


def c5_operator(d):
    """the banded Gaussian CSR operator of the C5 line (scripts/bench_other.py abrik): 200000 x 200000, 10 nonzeros per row, graded scalings"""
    import numpy as np
    import scipy.sparse as sp
    m = n = 200000
    rng = np.random.default_rng(77)
    nnz_row = 10
    rows = np.repeat(np.arange(m), nnz_row)
    colsi = (rows + np.tile(np.arange(-4, 6), m)) % n
    vals = rng.standard_normal(m * nnz_row)
    d1 = np.exp(-np.arange(m) / 4.0) + 1e-13
    d2 = np.exp(-np.arange(n) / 4.0) + 1e-13
    G = sp.csr_matrix((vals * d1[rows] * d2[colsi], (rows, colsi)), shape=(m, n)); G.sum_duplicates()
    return d.CsrOperator.from_scipy(G), G.nnz


def spmm_c5():
    d, ctx = _ctx()
    op, nnz = c5_operator(d)
    m = n = 200000
    X = d.cm_empty(n, 32); ctx.fill_dense(X, n, 32, key=(9, 0))
    Y = d.linop_apply(ctx, op, "L", "N", X, m, 32, n)
    for _ in range(REPS + 2):
        d.linop_apply(ctx, op, "L", "N", X, m, 32, n, C_in=Y)
    ctx.sync()


def getrf_panel_f32():
    d, ctx = _ctx()
    import torch

    m, n = 65536, 2048
    A = torch.randn((n, m), dtype=torch.float32, device="cuda")
    ip = torch.zeros(n, dtype=torch.int64, device="cuda")
    for _ in range(REPS):
        B = A.clone()
        assert ctx.lib.rlhip_getrf_piv_f32(ctx.h, m, n, B.data_ptr(), m, ip.data_ptr()) >= 0
    ctx.sync()


def jacobi():
    d, ctx = _ctx()
    ctx.set_option("jacobi_persist", 0)           # rocprofv3 --pmc aborts on cooperative launches (rc -11): the per-launch sweeps carry the counters
    import torch

    n, k = 20000, 256
    BT = d.cm_empty(n, k); ctx.fill_dense(BT, n, k, key=(4, 0))
    S = torch.zeros(k, dtype=torch.float64, device="cuda")
    U = d.cm_empty(n, k); VT = d.cm_empty(k, k)
    for _ in range(REPS):
        W = BT.clone()
        assert ctx.lib.rlhip_gesdd_f64(ctx.h, n, k, W.data_ptr(), n, S.data_ptr(), U.data_ptr(), n, VT.data_ptr(), k, None) == 0
    ctx.sync()


def jacobi_persist():
    """the kernel that IS on the C2 path (jacobi_persist_kernel, ~2 ms of the shard step), launched the ordinary way (option 3: rocprofv3 --pmc
    aborts on cooperative launches): the workers alone, no clock holders, uncached hand-over"""
    d, ctx = _ctx()
    ctx.set_option("jacobi_persist", 3)
    import torch

    n, k = 20000, 256
    BT = d.cm_empty(n, k); ctx.fill_dense(BT, n, k, key=(4, 0))
    S = torch.zeros(k, dtype=torch.float64, device="cuda")
    U = d.cm_empty(n, k); VT = d.cm_empty(k, k)
    for _ in range(REPS):
        W = BT.clone()
        assert ctx.lib.rlhip_gesdd_f64(ctx.h, n, k, W.data_ptr(), n, S.data_ptr(), U.data_ptr(), n, VT.data_ptr(), k, None) == 0
    ctx.sync()


def qrcp_tag():
    d, ctx = _ctx()
    import torch

    dd, n = 1280, 1024
    A = d.cm_empty(dd, n); ctx.fill_dense(A, dd, n, key=(5, 0))
    J = torch.zeros(n, dtype=torch.int64, device="cuda")
    tau = torch.zeros(n, dtype=torch.float64, device="cuda")
    for _ in range(REPS):
        W = A.clone()
        assert ctx.lib.rlhip_geqp3_f64(ctx.h, dd, n, W.data_ptr(), dd, J.data_ptr(), tau.data_ptr()) == 0
    ctx.sync()


GB = 1e9
# name -> (function, kernel-name filter, what, algorithmic bytes per launch, flops per launch, bound)
WORKLOADS = {
    "gemm_sk_nn": (gemm_sk_nn, "gemm_sk_kernel<double, false,", "Y = A*Omega, 200000 x 20000 x 256 fp64 (C2 pass 1)",
                   8.0 * (200000 * 20000 + 20000 * 256 + 200000 * 256), 2.0 * 200000 * 20000 * 256, "mfma"),
    "gemm_sk_tn": (gemm_sk_tn, "gemm_sk_kernel<double, true,", "B^T = A^T*Q, 20000 x 256 x 200000 fp64 (C2 pass 2)",
                   8.0 * (200000 * 20000 + 20000 * 256 + 200000 * 256), 2.0 * 200000 * 20000 * 256, "mfma"),
    "gemm_sk_tri": (gemm_sk_tri, "gemm_sk_kernel<double, true,", "Gram matrix A^T A (upper 128-blocks), 1048576 x 1024 fp64 with CQRRPT's padded leading dimension m + 32 (C3)",
                    8.0 * (1048576 * 1024 + 1024 * 1024 / 2), 1.0 * 1048576 * 1024 * 1024, "mfma"),
    "trsm_fused": (trsm_fused, "trsm_fused_kernel<double, 8, 32, 0", "X U = B in place, 1048576 x 1024 fp64 (C3 second solve, reference order)",
                   8.0 * (2 * 1048576 * 1024 + 1024 * 1024 / 2), 1.0 * 1048576 * 1024 * 1024, "mfma"),
    "trsm_fused_oop": (trsm_fused_oop, "trsm_fused_kernel<double, 8, 32, 1", "W = (A P) inv(U) out of place with the pivoting folded in, 1048576 x 1024 fp64 (C3)",
                       8.0 * (2 * 1048576 * 1024 + 1024 * 1024 / 2), 1.0 * 1048576 * 1024 * 1024, "mfma"),
    "trsm_fused_f32_192": (trsm_fused_f32_192, "trsm_fused_kernel<float, 12, 32, 0", "X U = B in place, 49152 x 2048 fp32, 192-row workgroups (a Cholesky-QR panel of C4)",
                           4.0 * (2 * 49152 * 2048 + 2048 * 2048 / 2), 1.0 * 49152 * 2048 * 2048, "mfma"),
    "saso_apply": (saso_apply, "saso_apply_dma_kernel<double", "A_hat = S*A, S 1280 x 1048576 with 4 nonzeros per column, A 1048576 x 1024 fp64 (C3)",
                   8.0 * 1048576 * 1024 + 8.0 * 1280 * 1024, 2.0 * 4 * 1048576 * 1024, "hbm"),
    "gemm_f32_tn": (gemm_f32_tn, "gemm_sk_kernel<float, true,", "W = V^T C, 2048 x 16384 x 16384 fp32 (one chunk of C4's compact-WY apply)",
                    4.0 * (16384 * 2048 + 16384 * 16384 + 2048 * 16384), 2.0 * 2048 * 16384 * 16384, "mfma"),
    "gemm_f32_nn": (gemm_f32_nn, "gemm_sk_kernel<float, false,", "C -= V W, 65536 x 16384 x 2048 fp32 (C4's compact-WY apply)",
                    4.0 * (65536 * 2048 + 2 * 65536 * 16384 + 2048 * 16384), 2.0 * 65536 * 16384 * 2048, "mfma"),
    "spmm_c5": (spmm_c5, "csr_spmm", "Y = A X, A 200000 x 200000 CSR with 2e6 nonzeros, X 32 columns fp64 (C5's operator product; the SpMM kernel of the launch)",
                (2000000 * 32 + 200000 * 32) * 8.0 + 16.0 * 2000000, 2.0 * 2000000 * 32, "hbm"),
    "getrf_panel_f32": (getrf_panel_f32, "getrf_panel_f32_kernel", "row-pivoted LU panel steps of the 65536 x 2048 fp32 transposed sketch (C4 qrcp_wide)",
                        None, None, "latency"),
    "jacobi": (jacobi, "jacobi_block_kernel", "one-sided Jacobi on the 256 x 256 factor of the RSVD tail (C2), PER-ROUND kernel (not the one on the C2 path: see jacobi_persist)", None, None, "latency"),
    "jacobi_persist": (jacobi_persist, "jacobi_persist_kernel", "one-sided Jacobi on the 256 x 256 Gram factor of the RSVD tail (C2): the persistent kernel itself, workers alone, ordinary launch", None, None, "latency"),
    "qrcp_tag": (qrcp_tag, "qrcp_tag_kernel", "geqp3 of the 1280 x 1024 fp64 sketch (C3)", None, None, "latency"),
}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "--list":
        print(json.dumps({k: dict(filter=v[1], what=v[2], algorithmic_bytes=v[3], flops=v[4], bound=v[5]) for k, v in WORKLOADS.items()}))
        sys.exit(0)
    WORKLOADS[sys.argv[1]][0]()
