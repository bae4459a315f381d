"""BQRRP benchmark mains on the device library, writing the reference's file formats.

  python -m benchmarks.bqrrp speed_mat_size   <dir> <num_runs> <row/col ratio> <cols/block ratio> <m1> [m2 ...]
        (benchmark/bench_BQRRP/BQRRP_speed_comparisons_mat_size.cc) -> _BQRRP_speed_comparisons_mat_size_num_info_lines_7.txt
  python -m benchmarks.bqrrp speed_block_size <dir> <num_runs> <m> <n> <b1> [b2 ...]
        (BQRRP_speed_comparisons_block_size.cc) -> _BQRRP_speed_comparisons_block_size_num_info_lines_7.txt
  python -m benchmarks.bqrrp runtime_breakdown <dir> <qr_tall: cholqr|geqrf> <num_runs> <m> <n> <b1> [b2 ...]
        (BQRRP_runtime_breakdown.cc) -> _BQRRP_runtime_breakdown_num_info_lines_7.txt
  python -m benchmarks.bqrrp pivot_quality     <dir> <m> <n> <block_size> [mat_type]
        (BQRRP_pivot_quality.cc) -> _BQRRP_pivot_quality_metric_{1,2}_num_info_lines_6.txt
  python -m benchmarks.bqrrp error_analysis    <dir> <bqrrp|geqp3> <num_runs> <m> <n> <b1> [b2 ...]
        (BQRRP_error_analysis.cc) -> _BQRRP_error_analysis_num_info_lines_5.txt: per block size, one row per matrix type
        (polynomial, staircase, spiked, Kahan): avg ||AP - QR|| / ||A||, max - avg, avg ||Q'Q - I|| / sqrt(n), max - avg
"""
from __future__ import annotations

import sys
import time

import numpy as np

from randlapack_amd import device as d

from . import _common as c

QR_TALL = {"geqrt": 0, "cholqr": 1, "geqrf": 2}


def _speed_row(ctx, m, n, b, d_factor):
    """the seven timings of one line of the speed-comparison files: BQRRP+CholQR, BQRRP+QRF, HQRRP, HQRRP+QRF, HQRRP+CholQR, QRF, QP3
    (every algorithm on a freshly generated copy of the same Gaussian matrix)"""
    row = []
    for what in ("bqrrp_cholqr", "bqrrp_qrf", "hqrrp", "hqrrp_qrf", "hqrrp_cholqr", "qrf", "qp3"):
        A = c.regen(ctx, "gaussian", m, n)
        fn = {"bqrrp_cholqr": lambda: d.drv_bqrrp(ctx, A, m, n, b, d_factor, qr_tall=1),
              "bqrrp_qrf": lambda: d.drv_bqrrp(ctx, A, m, n, b, d_factor, qr_tall=2),
              # as the reference's mains call it: oversampling (d_factor - 1) * b, no pivoting inside the panel (:82,146,160,173)
              "hqrrp": lambda: d.drv_hqrrp(ctx, A, m, n, nb_alg=b, pp=int((d_factor - 1) * b), panel_pivoting=0, qr_type=0),
              "hqrrp_qrf": lambda: d.drv_hqrrp(ctx, A, m, n, nb_alg=b, pp=int((d_factor - 1) * b), panel_pivoting=0, qr_type=1),
              "hqrrp_cholqr": lambda: d.drv_hqrrp(ctx, A, m, n, nb_alg=b, pp=int((d_factor - 1) * b), panel_pivoting=0, qr_type=2),
              "qrf": lambda: c.geqrf(ctx, A, m, n), "qp3": lambda: c.geqp3(ctx, A, m, n)}[what]
        row.append(c.timed_us(fn))
        del A
    return row


def speed_mat_size(argv):
    directory, numruns, ratio, blk_ratio = argv[0], int(argv[1]), float(argv[2]), float(argv[3])
    m_sz = [int(x) for x in argv[4:]]
    ctx = d.Context(0)
    d_factor = 1.0
    path = c.out_path(directory, "_BQRRP_speed_comparisons_mat_size_num_info_lines_7.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the BQRRP speed comparison benchmark, recording the time it takes to perform BQRRP and alternative QR and QRCP factorizations."
                "\nFile format: 7 columns, containing time for each algorithm: BQRRP+CholQR, BQRRP+QRF, HQRRP, HQRRP+QRF, HQRRP+CholQR, QRF, QP3;"
                "               rows correspond to BQRRP runs with varying mat sizes, with numruns repititions of each mat size."
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput row sizes:{', '.join(map(str, m_sz))}, , input row/column ratio: {ratio}"
                f"\nAdditional parameters: BQRRP columns/block size ratio: {blk_ratio} num runs per size {numruns} BQRRP d factor: {d_factor:f}\n")
    t_all = time.perf_counter()
    for m in m_sz:
        n = int(m / ratio)
        b = max(1, int(n / blk_ratio))
        for _ in range(numruns):
            row = _speed_row(ctx, m, n, b, d_factor)
            with open(path, "a") as f:
                f.write(",  ".join(map(str, row)) + ",\n")
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


def speed_block_size(argv):
    directory, numruns, m, n = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    b_sz = [int(x) for x in argv[4:]]
    ctx = d.Context(0)
    d_factor = 1.0
    path = c.out_path(directory, "_BQRRP_speed_comparisons_block_size_num_info_lines_7.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the BQRRP speed comparison benchmark, recording the time it takes to perform BQRRP and alternative QR and QRCP factorizations."
                "\nFile format: 7 columns, containing time for each algorithm: BQRRP+CholQR, BQRRP+QRF, HQRRP, HQRRP+QRF, HQRRP+CholQR, QRF, QP3;"
                "               rows correspond to BQRRP runs with block sizes varying as specified, with numruns repititions of each block size."
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: BQRRP block sizes: {''.join(str(b) + ', ' for b in b_sz)}num runs per size {numruns} BQRRP d factor: {d_factor:f}\n")
    t_all = time.perf_counter()
    for b in b_sz:
        for _ in range(numruns):
            row = _speed_row(ctx, m, n, b, d_factor)
            with open(path, "a") as f:
                f.write(",  ".join(map(str, row)) + ",\n")
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


def runtime_breakdown(argv):
    directory, qr_tall, numruns, m, n = argv[0], argv[1], int(argv[2]), int(argv[3]), int(argv[4])
    b_sz = [int(x) for x in argv[5:]]
    ctx = d.Context(0)
    d_factor = 1.0
    path = c.out_path(directory, "_BQRRP_runtime_breakdown_num_info_lines_7.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the BQRRP runtime breakdown benchmark, recording the time it takes to perform every subroutine in BQRRP."
                "\nFile format: 10 data columns, each corresponding to a given BQRRP subroutine: skop_t_dur, preallocation_t_dur, qrcp_wide_t_dur, panel_preprocessing_t_dur, qr_tall_t_dur, q_reconstruction_t_dur, apply_transq_t_dur, sample_update_t_dur, t_other, total_t_dur"
                "               rows correspond to BQRRP runs with block sizes varying as specified, with numruns repititions of each block size"
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: Tall QR subroutine {qr_tall} BQRRP block sizes: {', '.join(map(str, b_sz))}, num runs per size {numruns} BQRRP d factor: {d_factor:f}\n")
    t_all = time.perf_counter()
    for b in b_sz:
        for _ in range(numruns):
            A = c.regen(ctx, "gaussian", m, n)
            t = d.drv_bqrrp(ctx, A, m, n, b, d_factor, qr_tall=QR_TALL[qr_tall], timing=True)["times_us"]
            # the device driver draws its workspace from the context's arena: the reference's preallocation column is 0
            cols = [t[0], 0] + list(t[1:])
            with open(path, "a") as f:
                f.write(", ".join(map(str, cols)) + ", \n")
            del A
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


def pivot_quality(argv):
    directory, m, n, b = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    m_type = argv[4] if len(argv) > 4 else "polynomial"
    kw = dict(cond_num=1e10, exponent=2.0) if m_type in ("polynomial", "exponential", "step") else {}
    ctx = d.Context(0)
    d_factor = 1.0
    hdr = (f"\nNum OMP threads:0 (device: MI355X)\nInput type:{c.MAT_TYPE_IDS[m_type]}\nInput size:{m} by {n}"
           f"\nAdditional parameters: BQRRP block size: {b} BQRRP d factor: {d_factor:f}\n")
    # metric 1: trailing-block norm ratios ||R_qp3[i:, i:]|| / ||R_bqrrp[i:, i:]||  (BQRRP_pivot_quality.cc:116-176)
    A = c.regen(ctx, m_type, m, n, **kw)
    c.geqp3(ctx, A, m, n)
    R_qp3 = c.upper_factor(A, m, n)
    A = c.regen(ctx, m_type, m, n, **kw)
    d.drv_bqrrp(ctx, A, m, n, b, d_factor, qr_tall=1)
    R_bq = c.upper_factor(A, m, n)
    p1 = c.out_path(directory, "_BQRRP_pivot_quality_metric_1_num_info_lines_6.txt")
    with np.errstate(divide="ignore", invalid="ignore"):
        ratios = c.trailing_norms(R_qp3) / c.trailing_norms(R_bq)
    with open(p1, "a") as f:
        f.write("Description: Results of the BQRRP pivot quality benchmark for the metric of ratios of the norms of R factors output by QP3 and BQRRP."
                "\nFile format: File output is one-line." + hdr)
        f.write("".join(f"{x:g},  " for x in ratios) + "\n")
    # metric 2: |R_ii| / sigma_i, line one GEQP3, line two BQRRP (the order the reference writes them, :268-283)
    A = c.regen(ctx, m_type, m, n, **kw)
    S = c.singular_values(ctx, A, m, n)
    p2 = c.out_path(directory, "_BQRRP_pivot_quality_metric_2_num_info_lines_6.txt")
    with open(p2, "a") as f, np.errstate(divide="ignore", invalid="ignore"):
        f.write("Description: Results of the BQRRP pivot quality benchmark for the metric of ratios of the diagonal R entries to true singular values."
                "\nFile format: Line one contains BQRRP retults, line 2 contains GEQP3 retults." + hdr)
        f.write("".join(f"{x:g},  " for x in np.abs(np.diag(R_qp3)) / S) + "\n")
        f.write("".join(f"{x:g},  " for x in np.abs(np.diag(R_bq)) / S) + "\n")
    return p1, p2


def error_analysis(argv):
    import torch

    directory, alg, num_runs, m, n = argv[0], argv[1], int(argv[2]), int(argv[3]), int(argv[4])
    b_sz = [int(x) for x in argv[5:]]
    ctx = d.Context(0)
    tests = [("polynomial", dict(cond_num=1e10, exponent=2.0)), ("step", dict(cond_num=1e10)), ("spiked", dict(scaling=1e10)),
             ("kahan", dict(theta=1.2, perturb=1e3))]
    path = c.out_path(directory, "_BQRRP_error_analysis_num_info_lines_5.txt")
    with open(path, "a") as f:
        f.write(f"Description: Results from the {alg} error analysis; putput rows capture results per given matrix type, columns capture results per error type."
                "\nAt the moment, i test polynomial, staircase and spiked matrices with reconstructiuon error, max column norm error and orthogonality loss."
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: BQRRP block sizes: {', '.join(map(str, b_sz))}, \n")
    for b in (b_sz if alg == "bqrrp" else b_sz[:1]):
        for m_type, kw in tests:
            if m_type == "kahan" and m != n:
                continue                                               # the Kahan matrix is square (rl_gen.hh:408-434)
            rec, orth = [], []
            for run in range(num_runs):
                A0 = c.regen(ctx, m_type, m, n, key=(run, 0), **kw)
                A = A0.clone()
                if alg == "bqrrp":
                    out = d.drv_bqrrp(ctx, A, m, n, b, 1.0, qr_tall=1, tol=float(np.finfo(np.float64).eps ** 0.75))
                    J, tau, k = out["J"], out["tau"], out["rank"]
                else:
                    J, tau = c.geqp3(ctx, A, m, n)
                    k = min(m, n)
                # Q (m x k) from the reflectors, R (k x n) upper-trapezoidal (error_check(), BQRRP_error_analysis.cc:60-103)
                Q = A[:k].clone()                                       # first k columns (column-major m x k as a (k, m) tensor)
                getattr(ctx.lib, "rlhip_ungqr_f64")(ctx.h, m, k, k, Q.data_ptr(), m, tau.data_ptr())
                R = torch.triu(A[:, :k].T)                              # (k, n)
                AP = A0[(J - 1).long()]                                 # permuted columns, (n, m)
                resid = AP - R.T @ Q                                    # (n, m) == (A P - Q R)^T
                rec.append(float(torch.linalg.norm(resid) / torch.linalg.norm(A0)))
                G = Q @ Q.T
                orth.append(float(torch.linalg.norm(G - torch.eye(k, dtype=G.dtype, device=G.device)) / np.sqrt(n)))
            ar, ao = float(np.mean(rec)), float(np.mean(orth))
            with open(path, "a") as f:
                f.write(f"{ar:.14e},  {max(rec) - ar:.14e},  {ao:.14e},  {max(orth) - ao:.14e},\n")
    return path


# ---------------------------------------------------------------------------------------------------------------------
# BQRRP_subroutines_speed.cc: the alternatives of BQRRP's three subroutines, timed in isolation
# ---------------------------------------------------------------------------------------------------------------------
def _lib_call(ctx, name, A, *args):
    rc = getattr(ctx.lib, f"rlhip_{name}_{d._suffix(A)[0]}")(ctx.h, *args)
    assert rc == 0, (name, rc)


def _geqrt(ctx, A, m, n, nb, Tm, tau):
    """lapack::geqrt as the C++ layer composes it (rl_lapackpp.hh): geqrf + one compact-WY T per nb-wide block"""
    _lib_call(ctx, "geqrf", A, m, n, A.data_ptr(), m, tau.data_ptr())
    es = A.element_size()
    for i in range(0, n, nb):
        ib = min(nb, n - i)
        _lib_call(ctx, "larft", A, m - i, ib, A.data_ptr() + (i + i * m) * es, m, tau.data_ptr() + i * es, Tm.data_ptr() + i * n * es, n)


def subroutines_speed(argv):
    """<dir> <num_runs> <num_rows> <num_cols (increasing)...>   (BQRRP_subroutines_speed.cc:354-435)
    Three blocks of rows, in the reference's order: wide QRCP (GEQP3, LUQR), tall QR (GEQRF, GEQR, CHOLQR, CHOLQR_PRECOND,
    CHOLQR_HOUSE_REST, CHOLQR_R_RESTORE, then GEQRT per block size nb_start, 2 nb_start, ... n), apply Q^T (ORMQR, then GEMQRT per
    block size).  `lapack::geqr` (LAPACK's tall-skinny dispatcher) has no separate device kernel: its column repeats the device
    geqrf, whose tall panels already take the CholQR2 + reconstruction route."""
    import torch

    directory, numruns, m = argv[0], int(argv[1]), int(argv[2])
    n_sz = [int(x) for x in argv[3:]]
    nb_start = n_sz[0]
    ctx = d.Context(0)
    dev = "cuda:0"
    path = c.out_path(directory, "_BQRRP_subroutines_speed_num_info_lines_10.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the BQRRP subroutines benchmark, recording time for the alternative options of the three main BQRRP subroutines: wide_qrcp, tall qr and application of transpose orthonormal matrix."
                "\nFile format: the format varies for each subroutine"
                "               \n qrcp_wide: the first two columns show ORMQR and GEMM time, the third and any subsequent columns show time for GEMQRT with a given block size (from nb_start to n as specified). Rows vary from n_sz smallest to largest element in powers of two (with numruns runs per size)."
                "               \n qr_tall:   six columns with timing for different tall QR candidates and their related parts: GEQRF, GEQR, CHOLQR, CHOLQR_PREPROCESSING, CHOLQR_HOUSEHOLDER_RESTORATION, CHOLQR_UNTO_PRECONDITIONING."
                "               \n apply_Q:   three columns with tall QRCP candidates: GEQP3, LUQR"
                "               \n In all cases, rows vary from n_sz smallest to largest element in powers of two (with numruns runs per size)."
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput size:{m} by {''.join(str(x) + ', ' for x in n_sz)}"
                f"\nAdditional parameters num runs per size {numruns} nb_start {nb_start}\n")
    t_all = time.perf_counter()
    f64 = torch.float64
    # ---- wide QRCP on an n x m sketch-shaped matrix (:131-186)
    for n in n_sz:
        for _ in range(numruns):
            A = c.regen(ctx, "gaussian", n, m)
            dur_geqp3 = c.timed_us(lambda: c.geqp3(ctx, A, n, m))
            A = c.regen(ctx, "gaussian", n, m)
            At = d.cm_empty(m, n)
            piv = torch.zeros(n, dtype=torch.int64, device=dev)
            J = torch.zeros(m, dtype=torch.int64, device=dev)
            tau = torch.zeros(n, dtype=f64, device=dev)

            def luqr():
                _lib_call(ctx, "transpose", A, n, m, A.data_ptr(), n, At.data_ptr(), m, 0)
                _lib_call(ctx, "getrf_piv", A, m, n, At.data_ptr(), m, piv.data_ptr())
                assert ctx.lib.rlhip_luqrcp_piv(ctx.h, n, m, piv.data_ptr(), J.data_ptr()) == 0
                _lib_call(ctx, "col_swap", A, n, m, m, A.data_ptr(), n, J.data_ptr())
                _lib_call(ctx, "geqrf", A, n, m, A.data_ptr(), n, tau.data_ptr())
            dur_luqr = c.timed_us(luqr)
            with open(path, "a") as f:
                f.write(f"{dur_geqp3},  {dur_luqr},\n")
            del A, At
    # ---- tall QR on the m x n panel (:188-296)
    for n in n_sz:
        # R factor of the sketch's QRCP: the preconditioner of the CholQR candidate (:210-217)
        A = c.regen(ctx, "gaussian", m, n)
        S = d.cm_empty(n, m)
        ctx.fill_dense(S, n, m)
        A_sk = d.cm_empty(n, n)
        ctx.gemm("N", "N", n, n, m, 1.0, S, n, A, m, 0.0, A_sk, n)
        c.geqp3(ctx, A_sk, n, n)
        del S
        tau = torch.zeros(n, dtype=f64, device=dev)
        Tm = d.cm_zeros(n, n)
        R = d.cm_zeros(n, n)
        D = torch.zeros(n, dtype=f64, device=dev)
        for _ in range(numruns):
            cols = []
            nb = nb_start
            while nb <= n:
                A = c.regen(ctx, "gaussian", m, n)
                dur_geqrt = c.timed_us(lambda: _geqrt(ctx, A, m, n, nb, Tm, tau))
                if nb == nb_start:
                    A = c.regen(ctx, "gaussian", m, n)
                    dur_geqrf = c.timed_us(lambda: _lib_call(ctx, "geqrf", A, m, n, A.data_ptr(), m, tau.data_ptr()))
                    A = c.regen(ctx, "gaussian", m, n)
                    dur_geqr = c.timed_us(lambda: _lib_call(ctx, "geqrf", A, m, n, A.data_ptr(), m, tau.data_ptr()))
                    A = c.regen(ctx, "gaussian", m, n)
                    dur_pre = c.timed_us(lambda: ctx.trsm(m, n, 1.0, A_sk, n, A, m))

                    def cholqr():
                        ctx.laset("G", n, n, 0.0, 0.0, R, n)
                        ctx.syrk("U", "T", n, m, 1.0, A, m, 0.0, R, n)
                        ctx.potrf(n, R, n)
                        ctx.trsm(m, n, 1.0, R, n, A, m)
                    dur_cholqr = c.timed_us(cholqr)
                    dur_house = c.timed_us(lambda: _lib_call(ctx, "orhr_col", A, m, n, n, A.data_ptr(), m, Tm.data_ptr(), n, D.data_ptr()))

                    def restore():
                        _lib_call(ctx, "row_sign", A, n, R.data_ptr(), n, D.data_ptr())
                        ctx.trmm(n, n, 1.0, A_sk, n, R, n)
                        ctx.lacpy("U", n, n, R, n, A, m)
                    dur_restore = c.timed_us(restore)
                    cols += [dur_geqrf, dur_geqr, dur_cholqr, dur_pre, dur_house, dur_restore]
                cols.append(dur_geqrt)
                nb *= 2
            with open(path, "a") as f:
                f.write("".join(f"{x},  " for x in cols) + "\n")
        del A, A_sk
    # ---- Q^T applied to an m x (m - n) block (:298-352)
    for n in n_sz:
        if m - n <= 0:
            continue
        tau = torch.zeros(n, dtype=f64, device=dev)
        Tm = d.cm_zeros(n, n)
        Tb = d.cm_zeros(n, n)
        R = d.cm_zeros(n, n)
        D = torch.zeros(n, dtype=f64, device=dev)
        for _ in range(numruns):
            cols = []
            nb = nb_start
            while nb <= n:
                A = c.regen(ctx, "gaussian", m, n)
                B = c.regen(ctx, "gaussian", m, m - n, key=(1, 0))
                ctx.laset("G", n, n, 0.0, 0.0, R, n)
                ctx.syrk("U", "T", n, m, 1.0, A, m, 0.0, R, n)
                ctx.potrf(n, R, n)
                ctx.trsm(m, n, 1.0, R, n, A, m)                       # orthonormal m x n panel
                Ag = A.clone()
                _lib_call(ctx, "orhr_col", Ag, m, n, nb, Ag.data_ptr(), m, Tb.data_ptr(), n, D.data_ptr())
                dur_gemqrt = c.timed_us(lambda: _lib_call(ctx, "gemqrt", Ag, b"L", b"T", m, m - n, n, nb, Ag.data_ptr(), m, Tb.data_ptr(), n, B.data_ptr(), m))
                if nb == nb_start:
                    _lib_call(ctx, "orhr_col", A, m, n, n, A.data_ptr(), m, Tm.data_ptr(), n, D.data_ptr())
                    _lib_call(ctx, "tau_from_t", A, n, n, Tm.data_ptr(), n, tau.data_ptr())
                    B2 = c.regen(ctx, "gaussian", m, m - n, key=(1, 0))
                    T2 = d.cm_zeros(n, n)

                    def ormqr():                                       # lapack::ormqr as the C++ layer composes it: larft + one k x k block
                        _lib_call(ctx, "larft", A, m, n, A.data_ptr(), m, tau.data_ptr(), T2.data_ptr(), n)
                        _lib_call(ctx, "gemqrt", A, b"L", b"T", m, m - n, n, n, A.data_ptr(), m, T2.data_ptr(), n, B2.data_ptr(), m)
                    cols.append(c.timed_us(ormqr))
                    del B2
                cols.append(dur_gemqrt)
                nb *= 2
                del A, B, Ag
            with open(path, "a") as f:
                f.write("".join(f"{x},  " for x in cols) + "\n")
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


def hqrrp_runtime_breakdown(argv):
    """<dir> <num_runs> <num_rows> <num_cols> <block_sizes...>   (HQRRP_runtime_breakdown.cc:99-168): 27 numbers per run -- hqrrp's
    `timing` array (rl_hqrrp.hh:1144-1164; layout in include/RandLAPACK_amd/rl_hqrrp.hh)"""
    directory, numruns, m, n = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    b_sz = [int(x) for x in argv[4:]]
    ctx = d.Context(0)
    d_factor = 1.0
    path = c.out_path(directory, "_HQRRP_runtime_breakdown_num_info_lines_7.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the HQRRP runtime breakdown benchmark, recording the time it takes to perform every subroutine in HQRRP."
                "\nFile format: 26 data columns, each corresponding to a given HQRRP subroutine (please see /RandLAPACK/drivers/rl_hqrrp.hh for details)"
                "\nrows correspond to HQRRP runs with block sizes varying as specified, with numruns repititions of each block size"
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: HQRRP block sizes: {''.join(str(b) + ', ' for b in b_sz)}num runs per size {numruns} HQRRP d factor: {d_factor:f}\n")
    t_all = time.perf_counter()
    for b in b_sz:
        for _ in range(numruns):
            A = c.regen(ctx, "gaussian", m, n)
            o = d.drv_hqrrp_timed(ctx, A, m, n, nb_alg=b, pp=int((d_factor - 1) * b), panel_pivoting=0, qr_type=0)
            with open(path, "a") as f:
                f.write("".join(f"{x:g}, " for x in o["times_us"]) + "\n")
            del A
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


MAINS = {"subroutines_speed": subroutines_speed, "hqrrp_runtime_breakdown": hqrrp_runtime_breakdown, "speed_mat_size": speed_mat_size, "speed_block_size": speed_block_size, "runtime_breakdown": runtime_breakdown, "pivot_quality": pivot_quality, "error_analysis": error_analysis}

if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in MAINS:
        print(__doc__)
        sys.exit(1)
    print(MAINS[sys.argv[1]](sys.argv[2:]))
