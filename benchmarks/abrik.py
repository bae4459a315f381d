"""ABRIK benchmark mains on the device library (benchmark/bench_ABRIK/ABRIK_speed_comparisons.cc, ABRIK_runtime_breakdown.cc and their
_sparse twins).

  python -m benchmarks.abrik runtime_breakdown <dir> <mat_type | sparse:<density>> <num_runs> <m> <n> <custom_rank> <num_block_sizes> <num_matmul_sizes> <block sizes...> <matmul counts...>
        -> _ABRIK_runtime_breakdown_num_info_lines_6.txt: block size, matmuls, then the 13 entries of ABRIK::times (microseconds):
           allocation, get_factors, ungqr, reorth, qr, gemm_A, main_loop, sketching, r_cpy, s_cpy, norm, rest, total
  python -m benchmarks.abrik speed <dir> <mat_type> <num_runs> <m> <n> <target_rank> <num_block_sizes> <num_matmul_sizes> <block sizes...> <matmul counts...>
  python -m benchmarks.abrik speed_sparse <dir> <path.mtx | sparse:<density>:<m>:<n>> <num_runs> <target_rank> <num_block_sizes> <num_matmul_sizes> <block sizes...> <matmul counts...>
        -> _ABRIK_speed_comparisons_sparse_num_info_lines_6.txt (ABRIK on a CSR operator in HBM vs a host SVDS)

The reference reads its input matrix from a file; here it is generated in HBM (gen::mat_gen types: polynomial, exponential, step,
gaussian).  Output `_ABRIK_speed_comparisons_num_info_lines_6.txt`: 15 columns -- block size, matmuls, target rank, then (residual
error, low-rank error, time in us) for ABRIK, RSVD, SVDS, SVD.  The SVDS columns (Spectra, a host Eigen solver in the reference)
are written as 0; the low-rank error is measured against the device SVD truncated to target_rank."""
from __future__ import annotations

import sys
import time

import numpy as np
import torch

from randlapack_amd import device as d

from . import _common as c


def _residual(A, U, S, V, k):
    """sqrt(||A V - U S||_F^2 + ||A^T U - V S||_F^2) on the leading k triplets (residual_error_comp, :134-160); tensors are column-major"""
    Uk, Vk, Sk = U[:k], V[:k], S[:k]                     # (k, m), (k, n) row-major == column-major m x k, n x k
    AV = Vk @ A                                           # (k, m) = (A V)^T  since A is stored as (n, m)
    ATU = Uk @ A.T                                        # (k, n) = (A^T U)^T
    return float(torch.sqrt(torch.linalg.norm(AV - Sk[:, None] * Uk) ** 2 + torch.linalg.norm(ATU - Sk[:, None] * Vk) ** 2))


def _lowrank_err(Aref, nref, U, S, V, k):
    """||U_k S_k V_k^T - A_svd_k||_F / ||A_svd_k||_F (approx_error_comp, :162-181)"""
    approx = (V[:k].T * S[:k]) @ U[:k]                    # (n, m) == column-major m x n
    return float(torch.linalg.norm(approx - Aref) / nref)


def speed(argv):
    directory, m_type, num_runs, m, n, target_rank = argv[0], argv[1], int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5])
    nb, nm = int(argv[6]), int(argv[7])
    b_sz = [int(x) for x in argv[8:8 + nb]]
    matmuls = [int(x) for x in argv[8 + nb:8 + nb + nm]]
    # strictly decaying spectrum (no plateau of ones): the rank-k truncation the low-rank error is measured against must be unique
    kw = dict(cond_num=1e8, exponent=2.0, frac_spectrum_one=0.0) if m_type == "polynomial" else (dict(cond_num=1e8) if m_type in ("exponential", "step") else {})
    ctx = d.Context(0)
    tol = float(np.finfo(np.float64).eps ** 0.85)
    path = c.out_path(directory, "_ABRIK_speed_comparisons_num_info_lines_6.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the ABRIK speed comparison benchmark, recording the time it takes to perform ABRIK and alternative methods for low-rank SVD."
                "\nFile format: 15 columns, showing krylov block size, nummber of matmuls permitted, and num svals and svecs to approximate, followed by the residual error, standard lowrank error and execution time for all algorithms (ABRIK, RSVD, SVDS, SVD)"
                "\n Rows correspond to algorithm runs with Krylov block sizes varying as specified, and numbers of matmuls varying as specified per eah block size, with num_runs repititions of each number of matmuls."
                "\nInput type:" + f"{m_type} (generated in HBM)"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: Krylov block sizes {', '.join(map(str, b_sz))}, matmuls: {', '.join(map(str, matmuls))}, num runs per size {num_runs} num singular values and vectors approximated {target_rank}\n")
    A = c.regen(ctx, m_type, m, n, **kw)
    # the dense SVD baseline (and the rank-target_rank reference for the low-rank error), once
    import ctypes as C
    Acpy = A.clone()
    Sd = torch.zeros(n, dtype=A.dtype, device=A.device)
    Ud = torch.zeros((n, m), dtype=A.dtype, device=A.device)
    VTd = torch.zeros((n, n), dtype=A.dtype, device=A.device)
    sw = C.c_int(0)
    dur_svd = c.timed_us(lambda: ctx.lib.rlhip_gesdd_f64(ctx.h, m, n, Acpy.data_ptr(), m, Sd.data_ptr(), Ud.data_ptr(), m, VTd.data_ptr(), n, C.byref(sw)))
    Vd = VTd.T.contiguous()                               # (n_triplets, n): row i = v_i  (VT column-major n x n -> tensor (n, n)[j, i] = VT[i, j])
    k = target_rank
    Aref = (Vd[:k].T * Sd[:k]) @ Ud[:k]
    nref = float(torch.linalg.norm(Aref))
    res_svd = _residual(A, Ud, Sd, Vd, k)
    t_all = time.perf_counter()
    for b in b_sz:
        for mm in matmuls:
            for _ in range(num_runs):
                holder = {}
                dur_abrik = c.timed_us(lambda: holder.update(o=d.drv_abrik(ctx, A, m, n, b, tol, max_krylov_iters=mm)))
                o = holder["o"]
                ka = min(k, o["triplets"])
                res_a = _residual(A, o["U"], o["S"], o["V"], ka)
                lr_a = _lowrank_err(Aref, nref, o["U"], o["S"], o["V"], ka)
                kr = max(1, b * mm // 2)                  # the reference gives RSVD the same matmul budget (:246)
                dur_rsvd = c.timed_us(lambda: holder.update(r=d.drv_rsvd(ctx, A, m, n, kr, kr, tol, 0, 1)))
                r = holder["r"]
                kk = min(k, r["k"])
                res_r = _residual(A, r["U"], r["S"], r["V"], kk)
                lr_r = _lowrank_err(Aref, nref, r["U"], r["S"], r["V"], kk)
                with open(path, "a") as f:
                    f.write(f"{b},  {mm},  {k},  {res_a:.16e},  {lr_a:.16e},  {dur_abrik},  {res_r:.16e},  {lr_r:.16e},  {dur_rsvd},  "
                            f"0,  0,  0,  {res_svd:.16e},  0.0,  {dur_svd},\n")
    with open(path, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return path


def runtime_breakdown(argv):
    directory, m_type, num_runs, m, n, custom_rank = argv[0], argv[1], int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5])
    nb, nm = int(argv[6]), int(argv[7])
    b_sz = [int(x) for x in argv[8:8 + nb]]
    matmuls = [int(x) for x in argv[8 + nb:8 + nb + nm]]
    ctx = d.Context(0)
    tol = float(np.finfo(np.float64).eps ** 0.85)
    if m_type.startswith("sparse"):                        # ABRIK_runtime_breakdown_sparse.cc reads a Matrix Market file; here: random CSR
        import scipy.sparse as sp
        density = float(m_type.split(":")[1]) if ":" in m_type else 1e-3
        nnz = int(density * m * n)
        if nnz > 2 * 10**8:
            raise SystemExit(f"{nnz} nonzeros requested: host-side generation is capped at 2e8")
        # positions drawn with replacement (duplicates are summed): scipy.sparse.random permutes all m*n cells for a legacy RandomState
        rng = np.random.default_rng(0)
        M = sp.coo_matrix((rng.standard_normal(nnz), (rng.integers(0, m, nnz), rng.integers(0, n, nnz))), shape=(m, n)).tocsr()
        op = d.CsrOperator.from_scipy(M)
    else:
        kw = dict(cond_num=1e8, exponent=2.0) if m_type in ("polynomial", "exponential") else (dict(cond_num=1e8) if m_type == "step" else {})
        A = c.regen(ctx, m_type, m, n, **kw)
        op = d.DenseOperator(A, m, n)
    path = c.out_path(directory, "_ABRIK_runtime_breakdown_num_info_lines_6.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the ABRIK runtime breakdown benchmark, recording the time it takes to perform every subroutine in ABRIK."
                "\nFile format: 13 data columns, each corresponding to a given ABRIK subroutine: allocation_t_dur, get_factors_t_dur, ungqr_t_dur, reorth_t_dur, qr_t_dur, gemm_A_t_dur, main_loop_t_dur, sketching_t_dur, r_cpy_t_dur, s_cpy_t_dur, norm_t_dur, t_rest, total_t_dur"
                "               rows correspond to ABRIK runs with block sizes varying as specified, with numruns repititions of each block size"
                f"\nInput type:{m_type} (generated in HBM)"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: Krylov block sizes {''.join(str(b) + ', ' for b in b_sz)} matmuls: {''.join(str(x) + ', ' for x in matmuls)}"
                f" num runs per size {num_runs} num singular values and vectors approximated {custom_rank}\n")
    for b in b_sz:
        for mm in matmuls:
            for _ in range(num_runs):
                o = d.drv_abrik_linop(ctx, op, b, tol, max_krylov_iters=mm, timing=True)
                with open(path, "a") as f:
                    f.write("".join(f"{x}, " for x in [b, mm] + o["times_us"]) + "\n")
    return path


def speed_sparse(argv):
    """<dir> <input: path.mtx | sparse:<density>:<m>:<n>> <num_runs> <target_rank> <num_block_sizes> <num_matmul_sizes> <block sizes...> <matmuls...>
    (ABRIK_speed_comparisons_sparse.cc:354-436).  ABRIK (qr_exp = cqrrt, as the reference sets it, :271) on a CSR operator in HBM
    against SVDS.  The reference's SVDS is Spectra (host, Eigen); here it is scipy.sparse.linalg.svds (host ARPACK) with the
    reference's triplet count min(b * matmuls / 2, n - 2).  Columns: block size, matmuls, target rank, ABRIK residual, ABRIK time
    (us), SVDS residual, SVDS time (us); residual = sqrt(||A V - U S||_F^2 + ||A^T U - V S||_F^2) on min(target, found) triplets."""
    import scipy.io
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl

    directory, src, num_runs, target_rank = argv[0], argv[1], int(argv[2]), int(argv[3])
    nb, nm = int(argv[4]), int(argv[5])
    b_sz = [int(x) for x in argv[6:6 + nb]]
    matmuls = [int(x) for x in argv[6 + nb:6 + nb + nm]]
    if src.startswith("sparse:"):
        _, dens, m, n = src.split(":")
        m, n, nnz = int(m), int(n), int(float(dens) * int(m) * int(n))
        rng = np.random.default_rng(0)
        # graded rows and columns: a spectrum that decays, so that a low-rank SVD is meaningful
        r_, c_ = rng.integers(0, m, nnz), rng.integers(0, n, nnz)
        M = sp.coo_matrix((rng.standard_normal(nnz) / ((1.0 + r_) ** 0.5 * (1.0 + c_) ** 0.5), (r_, c_)), shape=(m, n)).tocsr()
    else:
        M = sp.csr_matrix(scipy.io.mmread(src))
        m, n = M.shape
    M.sum_duplicates()
    ctx = d.Context(0)
    op = d.CsrOperator.from_scipy(M)
    tol = float(np.finfo(np.float64).eps ** 0.85)
    path = c.out_path(directory, "_ABRIK_speed_comparisons_sparse_num_info_lines_6.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the ABRIK speed comparison benchmark, recording the time it takes to perform ABRIK and alternative methods for low-rank SVD, specifically on sparse matrices."
                "\nFile format: 15 columns, showing krylov block size, nummber of matmuls permitted, and num svals and svecs to approximate, followed by the residual error, standard lowrank error and execution time for all algorithms (ABRIK, SVDS)"
                "\n Rows correspond to algorithm runs with Krylov block sizes varying as specified, and numbers of matmuls varying as specified per each block size, with num_runs repititions of each number of matmuls."
                f"\nInput type:{src}"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: Krylov block sizes {''.join(str(b) + ', ' for b in b_sz)} matmuls: {''.join(str(x) + ', ' for x in matmuls)}"
                f" num runs per size {num_runs} num singular values and vectors approximated {target_rank}\n")

    def resid(U, S, V, k):                                   # host: U (m, k'), V (n, k') numpy
        return float(np.hypot(np.linalg.norm(M @ V[:, :k] - U[:, :k] * S[:k]), np.linalg.norm(M.T @ U[:, :k] - V[:, :k] * S[:k])))

    for b in b_sz:
        for mm in matmuls:
            for _ in range(num_runs):
                hold = {}
                dur_abrik = c.timed_us(lambda: hold.update(o=d.drv_abrik_linop(ctx, op, b, tol, max_krylov_iters=mm, qr_exp=1)))
                o = hold["o"]
                ka = min(target_rank, o["triplets"])
                res_a = resid(d.cm_to_numpy(o["U"]), o["S"].cpu().numpy(), d.cm_to_numpy(o["V"]), ka)
                ks = max(1, min(b * mm // 2, n - 2, m - 2))
                t0 = time.perf_counter()
                Us, Ss, Vts = spl.svds(M, k=ks, ncv=min(2 * ks, n - 1, m - 1) if 2 * ks > ks + 1 else None)
                dur_svds = int((time.perf_counter() - t0) * 1e6)
                order = np.argsort(-Ss)
                res_s = resid(Us[:, order], Ss[order], Vts[order].T, min(target_rank, ks))
                with open(path, "a") as f:
                    f.write(f"{b},  {mm},  {target_rank},  {res_a:.16e},  {dur_abrik},  {res_s:.16e},  {dur_svds},\n")
    return path


MAINS = {"speed_sparse": speed_sparse, "speed": speed, "runtime_breakdown": runtime_breakdown}

if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in MAINS:
        print(__doc__)
        sys.exit(1)
    print(MAINS[sys.argv[1]](sys.argv[2:]))
