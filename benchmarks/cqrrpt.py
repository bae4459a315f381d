"""CQRRPT benchmark mains on the device library, writing the reference's file formats.

  python -m benchmarks.cqrrpt speed             <dir> <num_runs> <m> <n1> [n2 ...]
        (benchmark/bench_CQRRPT/CQRRPT_speed_comparisons.cc) -> _CQRRPT_speed_comparisons_num_info_lines_7.txt
        8 columns: GEQP3, GEQRF, CQRRPT default (geqp3), CQRRPT hqrrp, CQRRPT bqrrp, sCholQR3, GEQR, GEQPT
  python -m benchmarks.cqrrpt runtime_breakdown <dir> <num_runs> <m> <n1> [n2 ...]
        (CQRRPT_runtime_breakdown.cc): the 8 CQRRPT.times columns
  python -m benchmarks.cqrrpt error_analysis    <dir> <cqrrpt|geqp3> <num_runs> <m> <n1> [n2 ...]
        (CQRRPT_error_analysis.cc) -> _CQRRPT_error_analysis_num_info_lines_4.txt: per column size, one row per matrix type
        (polynomial, staircase, spiked, Kahan): avg ||AP - QR|| / ||A||, max - avg, avg ||Q'Q - I|| / sqrt(n), max - avg
  python -m benchmarks.cqrrpt pivot_quality     <dir> <m> <n> [mat_type]
        (CQRRPT_pivot_quality.cc): metric 1 = trailing-norm ratios vs GEQP3, metric 2 = |R_ii| / sigma_i (GEQP3 line, CQRRPT line)
"""
from __future__ import annotations

import sys
import time

import numpy as np
import torch

from randlapack_amd import device as d

from . import _common as c

D_FACTOR, NNZ = 1.25, 4            # benchmark settings, CQRRPT_speed_comparisons.cc:80


def _scholqr3(ctx, A, m, n):
    """the reference's inline shifted CholQR3 baseline (CQRRPT_speed_comparisons.cc:160-178): syrk, shift, potrf, trsm, twice more"""
    lib, h = ctx.lib, ctx.h
    R = d.cm_zeros(n, n, device=A.device)
    nrm = torch.zeros(1).double()
    import ctypes as C
    fro = C.c_double(0)
    lib.rlhip_lange_fro_f64(h, m, n, A.data_ptr(), m, C.byref(fro))
    shift = 11 * np.finfo(np.float64).eps * n * fro.value ** 2
    for it in range(3):
        lib.rlhip_syrk_f64(h, b"U", b"T", n, m, 1.0, A.data_ptr(), m, 0.0, R.data_ptr(), n)
        if it == 0:
            lib.rlhip_add_diag_f64(h, n, shift, R.data_ptr(), n)
        rc = lib.rlhip_potrf_f64(h, b"U", n, R.data_ptr(), n)
        assert rc == 0, rc
        lib.rlhip_trsm_f64(h, b"R", b"U", b"N", b"N", m, n, 1.0, R.data_ptr(), n, A.data_ptr(), m)


def speed(argv):
    directory, numruns, m = argv[0], int(argv[1]), int(argv[2])
    n_sz = [int(x) for x in argv[3:]]
    ctx = d.Context(0)
    path = c.out_path(directory, "_CQRRPT_speed_comparisons_num_info_lines_7.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the CQRRPT speed comparison benchmark, recording the time it takes to perform CQRRPT and alternative QR and QRCP factorizations."
                "\nFile format: 8 columns, containing time for each algorithm: GEQP3, GEQRF, CQRRPT(geqp3), CQRRPT(hqrrp), CQRRPT(bqrrp), sCholQR3, GEQR, GEQPT;"
                "               rows correspond to runs with varying column counts, with numruns repititions of each size."
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput size:{m} by {', '.join(map(str, n_sz))}"
                f"\nAdditional parameters: num runs per size {numruns} CQRRPT d factor: {D_FACTOR:f} nnz {NNZ}\n")
    t_all = time.perf_counter()
    for n in n_sz:
        for _ in range(numruns):
            row = []
            for what in ("geqp3", "geqrf", "cq_geqp3", "cq_hqrrp", "cq_bqrrp", "scholqr3", "geqr", "geqpt"):
                A = c.regen(ctx, "gaussian", m, n)

                def geqpt():                      # tall QR then QRCP of the small R (:199-203)
                    c.geqrf(ctx, A, m, n)
                    R = A[:, :n].contiguous()     # n x n leading block (column-major (n, n) tensor)
                    lib = ctx.lib
                    lib.rlhip_laset_f64(ctx.h, b"L", n - 1, n - 1, 0.0, 0.0, R.data_ptr() + 8, n)
                    c.geqp3(ctx, R, n, n)

                fn = {"geqp3": lambda: c.geqp3(ctx, A, m, n), "geqrf": lambda: c.geqrf(ctx, A, m, n),
                      "cq_geqp3": lambda: d.drv_cqrrpt(ctx, A, m, n, D_FACTOR, NNZ, qrcp=2),
                      "cq_hqrrp": lambda: d.drv_cqrrpt(ctx, A, m, n, D_FACTOR, NNZ, qrcp=0),
                      "cq_bqrrp": lambda: d.drv_cqrrpt(ctx, A, m, n, D_FACTOR, NNZ, qrcp=1),
                      "scholqr3": lambda: _scholqr3(ctx, A, m, n),
                      "geqr": lambda: c.geqrf(ctx, A, m, n),      # lapack::geqr picks TSQR or geqrf; the device has one tall QR
                      "geqpt": geqpt}[what]
                row.append(c.timed_us(fn))
                del A
            with open(path, "a") as f:
                f.write(",  ".join(map(str, row)) + ",\n")
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


def runtime_breakdown(argv):
    directory, numruns, m = argv[0], int(argv[1]), int(argv[2])
    n_sz = [int(x) for x in argv[3:]]
    ctx = d.Context(0)
    path = c.out_path(directory, "_CQRRPT_runtime_breakdown_num_info_lines_7.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the CQRRPT runtime breakdown benchmark, recording the time it takes to perform every subroutine in CQRRPT."
                "\nFile format: 8 data columns, each corresponding to a given CQRRPT subroutine: saso_t_dur, qrcp_t_dur, rank_reveal_t_dur, cholqr_t_dur, a_mod_piv_t_dur, a_mod_trsm_t_dur, t_rest, total_t_dur"
                "               rows correspond to CQRRPT runs with column counts varying as specified, with numruns repititions of each size"
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput size:{m} by {', '.join(map(str, n_sz))}"
                f"\nAdditional parameters: num runs per size {numruns} CQRRPT d factor: {D_FACTOR:f} nnz {NNZ}\n")
    t_all = time.perf_counter()
    for n in n_sz:
        for _ in range(numruns):
            A = c.regen(ctx, "gaussian", m, n)
            t = d.drv_cqrrpt(ctx, A, m, n, D_FACTOR, NNZ, timing=True)["times_us"]
            with open(path, "a") as f:
                f.write(",  ".join(map(str, t)) + ",\n")
            del A
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


def error_analysis(argv):
    directory, alg, num_runs, m = argv[0], argv[1], int(argv[2]), int(argv[3])
    col_sz = [int(x) for x in argv[4:]]
    ctx = d.Context(0)
    atol = float(np.finfo(np.float64).eps ** 0.75)                     # CQRRPT_error_analysis.cc:206
    tests = [("polynomial", dict(cond_num=1e10, exponent=2.0)), ("step", dict(cond_num=1e10)), ("spiked", dict(scaling=1e10)),
             ("kahan", dict(theta=1.2, perturb=1e3))]
    path = c.out_path(directory, "_CQRRPT_error_analysis_num_info_lines_4.txt")
    with open(path, "a") as f:
        f.write(f"Description: Results from the {alg} error analysis; putput rows capture results per given matrix type, columns capture results per error type."
                "\nAt the moment, i test polynomial, staircase and spiked matrices with reconstructiuon error, max column norm error and orthogonality loss."
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput size:{m} by {''.join(str(n) + ', ' for n in col_sz)}\n")
    for n in col_sz:
        for m_type, kw in tests:
            if m_type == "kahan" and m != n:
                continue                                               # the Kahan matrix is square (rl_gen.hh:408-434)
            rec, orth = [], []
            for run in range(num_runs):
                A0 = c.regen(ctx, m_type, m, n, key=(run, 0), **kw)
                A = A0.clone()
                if alg == "cqrrpt":
                    out = d.drv_cqrrpt(ctx, A, m, n, D_FACTOR, NNZ, eps=atol, key=(run, 1))
                    J, k = out["J"], out["rank"]
                    Q = A[:k]                                           # (k, m): the first k columns of the column-major m x n buffer
                    R = torch.triu(out["R"].T[:k])                      # (k, n)
                else:
                    J, tau = c.geqp3(ctx, A, m, n)
                    k = min(m, n)
                    R = torch.triu(A[:, :k].T)
                    Q = A[:k].clone()
                    ctx.lib.rlhip_ungqr_f64(ctx.h, m, k, k, Q.data_ptr(), m, tau.data_ptr())
                AP = A0[(J - 1).long()]                                 # permuted columns, (n, m)
                resid = AP - R.T @ Q                                    # (n, m) == (A P - Q R)^T   (error_check(), :83-106)
                rec.append(float(torch.linalg.norm(resid) / torch.linalg.norm(A0)))
                G = Q @ Q.T
                orth.append(float(torch.linalg.norm(G - torch.eye(k, dtype=G.dtype, device=G.device)) / np.sqrt(n)))
            ar, ao = float(np.mean(rec)), float(np.mean(orth))
            with open(path, "a") as f:
                f.write(f"{ar:.14e},  {max(rec) - ar:.14e},  {ao:.14e},  {max(orth) - ao:.14e},\n")
    return path


def pivot_quality(argv):
    directory, m, n = argv[0], int(argv[1]), int(argv[2])
    m_type = argv[3] if len(argv) > 3 else "polynomial"
    kw = dict(cond_num=1e10, exponent=2.0) if m_type in ("polynomial", "exponential", "step") else {}
    ctx = d.Context(0)
    hdr = (f"\nNum OMP threads:0 (device: MI355X)\nInput type:{c.MAT_TYPE_IDS[m_type]}\nInput size:{m} by {n}"
           f"\nAdditional parameters: CQRRPT d factor: {D_FACTOR:f}\n")
    A = c.regen(ctx, m_type, m, n, **kw)
    S = c.singular_values(ctx, A, m, n)
    c.geqp3(ctx, A, m, n)
    R_qp3 = c.upper_factor(A, m, n)
    A = c.regen(ctx, m_type, m, n, **kw)
    out = d.drv_cqrrpt(ctx, A, m, n, D_FACTOR, NNZ)
    k = out["rank"]
    R_cq = np.zeros((n, n))
    R_cq[:k, :] = np.triu(d.cm_to_numpy(out["R"]))[:k, :]
    p1 = c.out_path(directory, "_CQRRPT_pivot_quality_metric_1_num_info_lines_6.txt")
    p2 = c.out_path(directory, "_CQRRPT_pivot_quality_metric_2_num_info_lines_6.txt")
    with np.errstate(divide="ignore", invalid="ignore"):
        with open(p1, "a") as f:
            f.write("Description: Results of the CQRRPT pivot quality benchmark for the metric of ratios of the norms of R factors output by QP3 and CQRRPT."
                    "\nFile format: File output is one-line." + hdr)
            f.write("".join(f"{x:g},  " for x in c.trailing_norms(R_qp3) / c.trailing_norms(R_cq)) + "\n")
        with open(p2, "a") as f:
            f.write("Description: Results of the CQRRPT pivot quality benchmark for the metric of ratios of the diagonal R entries to true singular values."
                    "\nFile format: Line one contains GEQP3 retults, line 2 contains CQRRPT retults." + hdr)
            f.write("".join(f"{x:g},  " for x in np.abs(np.diag(R_qp3)) / S) + "\n")
            f.write("".join(f"{x:g},  " for x in np.abs(np.diag(R_cq)) / S) + "\n")
    return p1, p2


MAINS = {"speed": speed, "runtime_breakdown": runtime_breakdown, "pivot_quality": pivot_quality, "error_analysis": error_analysis}

if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in MAINS:
        print(__doc__)
        sys.exit(1)
    print(MAINS[sys.argv[1]](sys.argv[2:]))
