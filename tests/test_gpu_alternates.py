"""-m gpu: every RETAINED alternate kernel path (the environment knobs of DESIGN.md section 7 that select a different kernel, read once
per process) runs the same small battery against LAPACK / numpy in its own process: pivots identical, factors and products to
rounding.  The defaults are what the rest of the suite exercises; this file keeps the other side of each switch from rotting.
(Pure tuning constants and the kernels nothing selects any more were removed in round 3.)"""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

BATTERY = textwrap.dedent("""
    import ctypes as C
    import numpy as np, torch, scipy.linalg.lapack as ll
    from randlapack_amd import device as d
    ctx = d.Context(0)
    rng = np.random.default_rng(5)
    EPS = np.finfo(np.float64).eps
    # 1. pivoted QR of a sketch-sized matrix: pivots identical to LAPACK's dgeqp3
    m, n = 640, 256
    A = rng.standard_normal((m, n)) * np.logspace(0, -3, n)[rng.permutation(n)]
    Ad = d.cm_from_numpy(A); J = torch.zeros(n, dtype=torch.int64, device="cuda"); tau = torch.zeros(n, dtype=torch.float64, device="cuda")
    assert ctx.lib.rlhip_geqp3_f64(ctx.h, m, n, Ad.data_ptr(), m, J.data_ptr(), tau.data_ptr()) == 0
    qr_ref, jp_ref, tau_ref, _, info = ll.dgeqp3(A)
    assert np.array_equal(J.cpu().numpy(), jp_ref), "geqp3 pivots"
    assert np.abs(np.abs(np.triu(d.cm_to_numpy(Ad))[:n]) - np.abs(np.triu(qr_ref)[:n])).max() <= 1e-11 * np.abs(qr_ref).max()
    # 2. row-pivoted LU, tall panels in both precisions: pivots identical to LAPACK
    for (mm, nn, dt, fn, ref) in ((40000, 96, np.float64, ctx.lib.rlhip_getrf_f64, ll.dgetrf), (3000, 200, np.float64, ctx.lib.rlhip_getrf_f64, ll.dgetrf),
                                  (70000, 64, np.float32, ctx.lib.rlhip_getrf_f32, ll.sgetrf), (5000, 96, np.float32, ctx.lib.rlhip_getrf_f32, ll.sgetrf)):
        B = (rng.standard_normal((mm, nn)) * np.logspace(0, -2, nn)).astype(dt)
        Bd = d.cm_from_numpy(B); ip = torch.zeros(nn, dtype=torch.int64, device="cuda")
        assert fn(ctx.h, mm, nn, Bd.data_ptr(), mm, ip.data_ptr()) == 0
        lu_ref, piv_ref, _ = ref(B)
        assert np.array_equal(ip.cpu().numpy() - 1, piv_ref), ("getrf pivots", mm, nn, dt)
        assert np.abs(d.cm_to_numpy(Bd) - lu_ref).max() <= (1e-12 if dt is np.float64 else 2e-4) * np.abs(lu_ref).max()
    # 3. right-upper triangular solve, tall (fused kernel when it is on) and short
    for (mm, nn) in ((20000, 512), (3000, 300)):
        U = np.triu(rng.standard_normal((nn, nn))) + 25 * np.eye(nn)
        Bm = rng.standard_normal((mm, nn))
        Bd = d.cm_from_numpy(Bm)
        ctx.trsm(mm, nn, 1.0, d.cm_from_numpy(U), nn, Bd, mm)
        X = d.cm_to_numpy(Bd)
        assert np.linalg.norm(X @ U - Bm) <= 1e-13 * np.linalg.norm(Bm) * np.sqrt(nn), ("trsm", mm, nn)
    # 4. Cholesky, one-workgroup and two-level sizes
    for nn in (256, 1024):
        G0 = rng.standard_normal((nn + 50, nn)); G = G0.T @ G0
        Gd = d.cm_from_numpy(G)
        assert ctx.potrf(nn, Gd, nn) == 0
        R = np.triu(d.cm_to_numpy(Gd))
        assert np.linalg.norm(R.T @ R - G) <= 1e-13 * np.linalg.norm(G) * np.sqrt(nn), ("potrf", nn)
    # 5. thin SVD of a tall factor (Cholesky-QR + Jacobi), well and badly conditioned
    for cond in (3.0, 1e6):
        mm, nn = 3000, 256
        s = np.logspace(0, -np.log10(cond), nn)
        S0 = (np.linalg.qr(rng.standard_normal((mm, nn)))[0] * s) @ np.linalg.qr(rng.standard_normal((nn, nn)))[0].T
        Sd = d.cm_from_numpy(S0); Sv = torch.zeros(nn, dtype=torch.float64, device="cuda"); U = d.cm_empty(mm, nn); VT = d.cm_empty(nn, nn)
        assert ctx.lib.rlhip_gesdd_f64(ctx.h, mm, nn, Sd.data_ptr(), mm, Sv.data_ptr(), U.data_ptr(), mm, VT.data_ptr(), nn, None) == 0
        sv = Sv.cpu().numpy()
        assert np.max(np.abs(sv - s)) <= 1e-13 * np.sqrt(nn), ("gesdd sigma", cond)
        u, vt = d.cm_to_numpy(U), d.cm_to_numpy(VT)
        assert np.linalg.norm((u * sv) @ vt - S0) <= 1e-13 * np.sqrt(nn), ("gesdd residual", cond)
        assert np.linalg.norm(u.T @ u - np.eye(nn)) <= 1e-11 * np.sqrt(nn)
    # 6. the big-product shapes of the persistent GEMM, both precisions, both layouts, and the Gram map
    for dt, tol in ((torch.float64, 1e-13), (torch.float32, 3e-5)):
        for (ta, mm, nn, kk) in (("N", 4096, 256, 2048), ("T", 2048, 256, 32768)):
            Am = torch.randn((kk, mm) if ta == "N" else (mm, kk), dtype=dt, device="cuda")         # column-major storage
            Bm = torch.randn((nn, kk), dtype=dt, device="cuda")
            Cm = torch.zeros((nn, mm), dtype=dt, device="cuda")
            lda = mm if ta == "N" else kk
            ctx.gemm(ta, "N", mm, nn, kk, 1.0, Am, lda, Bm, kk, 0.0, Cm, mm)
            An = Am.T.double().cpu().numpy() if ta == "N" else Am.double().cpu().numpy()
            ref = An @ Bm.T.double().cpu().numpy()
            assert np.abs(Cm.T.double().cpu().numpy() - ref).max() <= tol * np.sqrt(kk) * np.abs(ref).max(), ("gemm", dt, ta)
    Am = torch.randn((512, 40000), dtype=torch.float64, device="cuda")                             # 40000 x 512, Gram matrix
    Gm = torch.zeros((512, 512), dtype=torch.float64, device="cuda")
    ctx.syrk("U", "T", 512, 40000, 1.0, Am, 40000, 0.0, Gm, 512)
    ref = np.triu((Am @ Am.T).cpu().numpy())
    assert np.abs(np.triu(Gm.T.cpu().numpy()) - ref).max() <= 1e-13 * 200 * np.abs(ref).max()
    # 7. unpivoted Householder QR: sketch-sized (pipelined kernel), tall-skinny well and badly conditioned (Cholesky-QR panels, preconditioned retry)
    for (mm, nn, cond) in ((1280, 512, 10.0), (20000, 64, 1e2), (20000, 64, 1e10)):
        Q0 = (np.linalg.qr(rng.standard_normal((mm, nn)))[0] * np.logspace(0, -np.log10(cond), nn)) @ np.linalg.qr(rng.standard_normal((nn, nn)))[0].T
        Qd = d.cm_from_numpy(Q0); tq = torch.zeros(nn, dtype=torch.float64, device="cuda")
        assert ctx.lib.rlhip_geqrf_f64(ctx.h, mm, nn, Qd.data_ptr(), mm, tq.data_ptr()) == 0
        qr_ref, tau_ref, _, _ = ll.dgeqrf(Q0)
        got = d.cm_to_numpy(Qd)
        assert np.abs(np.abs(np.triu(got)[:nn]) - np.abs(np.triu(qr_ref)[:nn])).max() <= 1e-9 * np.abs(qr_ref).max(), ("geqrf R", mm, nn, cond)
    # 8. HQRRP with pivoted tall panels: a valid GEQP3-format factorization
    mm, nn = 4096, 512
    H0 = rng.standard_normal((mm, nn)) * np.logspace(0, -4, nn)[rng.permutation(nn)]
    Hd = d.cm_from_numpy(H0)
    r = d.drv_hqrrp(ctx, Hd, mm, nn, nb_alg=64, pp=10, panel_pivoting=1, qr_type=0, key=(2, 0))
    Jh = r["J"].cpu().numpy(); Ho = d.cm_to_numpy(Hd); th = r["tau"].cpu().numpy()
    assert sorted(Jh.tolist()) == list(range(1, nn + 1))
    qfull, _, info = ll.dorgqr(np.asfortranarray(Ho[:, :nn].copy()), th)
    assert np.linalg.norm(H0[:, Jh - 1] - qfull @ np.triu(Ho)[:nn]) <= EPS**0.75 * np.linalg.norm(H0)
    # 9. Cholesky-QR of a tall 256-column block (one-stream route when it is on) and the small-product kernel (256^3, all transpositions)
    Y0 = rng.standard_normal((20000, 256)) @ (np.eye(256) + 0.01 * rng.standard_normal((256, 256)))
    Yd = d.cm_from_numpy(Y0)
    rc, fail = d.drv_stab(ctx, 0, Yd, 20000, 256)
    Qy = d.cm_to_numpy(Yd)
    assert rc == 0 and not fail and np.linalg.norm(Qy.T @ Qy - np.eye(256)) <= 1e-12 * 256, "cholqrq"
    assert np.linalg.norm(Y0 - Qy @ (Qy.T @ Y0)) <= 1e-12 * np.linalg.norm(Y0)
    for ta in "NT":
        for tb in "NT":
            P0, P1 = rng.standard_normal((256, 256)), rng.standard_normal((256, 256))
            Cd = d.cm_zeros(256, 256)
            ctx.gemm(ta, tb, 256, 256, 256, 1.0, d.cm_from_numpy(P0), 256, d.cm_from_numpy(P1), 256, 0.0, Cd, 256)
            ref = (P0 if ta == "N" else P0.T) @ (P1 if tb == "N" else P1.T)
            assert np.abs(d.cm_to_numpy(Cd) - ref).max() <= 1e-13 * 16 * np.abs(ref).max(), ("small gemm", ta, tb)
    # 10. sparse operator products with narrow, medium and wide right-hand sides, both directions
    import scipy.sparse as sp
    Ssp = sp.random(3000, 1700, 0.01, random_state=np.random.default_rng(3), format="csr", data_rvs=np.random.default_rng(4).standard_normal).tocsr()
    op = d.CsrOperator.from_scipy(Ssp)
    for nb_ in (7, 16, 32, 100, 200):
        X = rng.standard_normal((1700, nb_)); Z = rng.standard_normal((3000, nb_))
        Y = d.cm_to_numpy(d.linop_apply(ctx, op, "L", "N", d.cm_from_numpy(X), 3000, nb_, 1700))
        assert np.abs(Y - Ssp @ X).max() <= 1e-13 * np.abs(Ssp @ X).max() * 10, ("spmm", nb_)
        Yt = d.cm_to_numpy(d.linop_apply(ctx, op, "L", "T", d.cm_from_numpy(Z), 1700, nb_, 3000))
        assert np.abs(Yt - Ssp.T @ Z).max() <= 1e-13 * np.abs(Ssp.T @ Z).max() * 10, ("spmm^T", nb_)
    print("BATTERY OK")
""")

# knob -> alternate value (the default is the other one); each selects a different kernel / algorithm
ALTERNATES = [("RLHIP_TRSM_FUSED", "0"), ("RLHIP_TRSM_BLK", "0"), ("RLHIP_STREAMK", "0"), ("RLHIP_STREAMK_F32", "0"), ("RLHIP_STREAMK_F32", "2"),
              ("RLHIP_STREAMK_F32_CHUNK", "0"), ("RLHIP_RECOVER_V", "0"), ("RLHIP_CHOLQR2_SKIP", "0"), ("RLHIP_JACOBI_PERSIST", "0"),
              ("RLHIP_QR_PIPE", "0"), ("RLHIP_QR_BLK", "0"), ("RLHIP_QRCP_TAG", "0"), ("RLHIP_LU_TAG", "0"), ("RLHIP_LU_REG_PANEL", "0"), ("RLHIP_LU_F64_FAST", "0"),
              ("RLHIP_LU_F32_FAST", "0"), ("RLHIP_HQRRP_TALL_PANEL", "0"), ("RLHIP_GEQRF_PRECOND", "0"), ("RLHIP_TRSM_FUSED_MIN_ROWS", "1000"),
              # round 4
              ("RLHIP_GESDD_GRAM", "0"), ("RLHIP_JACOBI_HOLD", "0"), ("RLHIP_JACOBI_QW", "32"), ("RLHIP_CHOLQRQ_FUSED", "0"), ("RLHIP_TRSM_XASM", "0"),
              ("RLHIP_GEMM_SMALL", "0"), ("RLHIP_GEQRF_SCALE_GUARD", "0"), ("RLHIP_SPMM_NARROW", "0"), ("RLHIP_SPMM_CMOUT", "0")]


@pytest.mark.parametrize("knob,value", [("(defaults)", "")] + ALTERNATES)
def test_alternate_kernel_paths_pass_the_battery(knob, value):
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    if value:
        env[knob] = value
    r = subprocess.run([sys.executable, "-c", BATTERY], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "BATTERY OK" in r.stdout, f"{knob}={value}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
