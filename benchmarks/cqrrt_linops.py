"""CQRRT_linops scaling study on the device library (benchmark/bench_CQRRT_linops/CQRRT_linop_basic.cc).

  python -m benchmarks.cqrrt_linops basic <dir> <num_sizes> <num_runs> <m_start> <m_end> <aspect_ratio> <nnz_per_row> <d_factor> [sketch_nnz] [block_size]
  python -m benchmarks.cqrrt_linops composite_applications double <dir> <num_runs> <K.mtx | gen:<m>> <V.mtx | gen:<n>:<nnz_per_row>> <d_factor> [sketch_nnz] [block_size] [skip_apps] [compute_cond]

Tall sparse operators (CSR in HBM, nnz_per_row nonzeros in every row) of m rows and n = m / aspect_ratio columns, m swept
geometrically from m_start to m_end.  Writes `cqrrt_linop_results.csv` with the reference's quality / time columns for CQRRT_linops,
CholQR_linops, sCholQR3_linops and the dense-operand CQRRT (the memory columns are the analytical working-set sizes; peak RSS is a
host notion and is written as 0)."""
from __future__ import annotations

import sys
import time

import numpy as np
import torch

from randlapack_amd import device as d

from . import _common as c


def _quality(Q, n):
    """orth_error = ||Q^T Q - I||_F / sqrt(n); max orthonormal prefix; flag (test_utils: orth_error below eps^0.75)"""
    G = Q @ Q.T                                            # Q is (n, m): G = Q^T Q in the column-major reading
    E = G - torch.eye(n, dtype=G.dtype, device=G.device)
    err = float(torch.linalg.norm(E)) / np.sqrt(n)
    tol = float(np.finfo(np.float64).eps ** 0.75)
    # ||E[:k, :k]||_F for every prefix k: prefix sums over the "L-shaped" shells of E^2
    E2 = E ** 2
    shell = torch.cumsum(E2, 1).diagonal() + torch.cumsum(E2, 0).diagonal() - E2.diagonal()      # sum of row k and column k up to the diagonal
    lead = torch.sqrt(torch.cumsum(shell, 0))
    ks = torch.arange(1, n + 1, device=G.device, dtype=G.dtype)
    ok = (lead / torch.sqrt(ks)) <= tol
    max_cols = n if bool(ok.all()) else int(torch.nonzero(~ok)[0].item())
    return err, max_cols, int(err <= tol)


def basic(argv):
    directory, num_sizes, num_runs = argv[0], int(argv[1]), int(argv[2])
    m_start, m_end, aspect, r, d_factor = int(argv[3]), int(argv[4]), float(argv[5]), int(argv[6]), float(argv[7])
    sketch_nnz = int(argv[8]) if len(argv) > 8 else 4
    block = int(argv[9]) if len(argv) > 9 else 0
    ctx = d.Context(0)
    path = c.out_path(directory, "cqrrt_linop_results.csv")
    with open(path, "w") as f:
        f.write("# CQRRT_linop vs CholQR vs sCholQR3 vs CQRRT_expl Results\n# Precision: double\n"
                f"# d_factor (CQRRT_linop only): {d_factor}\n# sketch_nnz (CQRRT_linop only): {sketch_nnz}\n"
                f"# block_size (CQRRT_linop, CholQR, sCholQR3): {block} (0 = full)\n# num_runs: {num_runs}\n# OpenMP threads: 0 (device: MI355X)\n"
                "# Format: per-run per-algorithm quality metrics (orth_error, max_orth_cols, orth_flag, time), memory (KB)\n"
                "m,n,run,aspect_ratio,cond_num,density,"
                "cqrrt_orth_error,cqrrt_max_orth_cols,cqrrt_is_orth,cqrrt_time_us,"
                "cholqr_orth_error,cholqr_max_orth_cols,cholqr_is_orth,cholqr_time_us,"
                "scholqr3_orth_error,scholqr3_max_orth_cols,scholqr3_is_orth,scholqr3_time_us,"
                "dense_cqrrt_orth_error,dense_cqrrt_max_orth_cols,dense_cqrrt_is_orth,dense_cqrrt_time_us,"
                "cqrrt_peak_rss_kb,cqrrt_analytical_kb,cholqr_peak_rss_kb,cholqr_analytical_kb,"
                "scholqr3_peak_rss_kb,scholqr3_analytical_kb,dense_cqrrt_peak_rss_kb,dense_cqrrt_analytical_kb\n")
    sizes = np.unique(np.round(np.geomspace(m_start, m_end, num_sizes)).astype(np.int64))
    rng = np.random.default_rng(0)
    for m in sizes:
        m = int(m)
        n = max(1, int(m / aspect))
        cols = np.sort(rng.integers(0, n, size=(m, r)), axis=1).astype(np.int64).ravel()
        op = d.CsrOperator(m, n, torch.as_tensor(np.arange(m + 1, dtype=np.int64) * r, device="cuda:0"), torch.as_tensor(cols, device="cuda:0"),
                           torch.as_tensor(rng.standard_normal(m * r), device="cuda:0"))
        dens = r / n
        b_eff = n if block <= 0 or block >= n else block
        kb = lambda elems: int(elems * 8 / 1024)
        for run in range(num_runs):
            row = [m, n, run, aspect, 0, f"{dens:.6f}"]
            for alg in ("cqrrt", "cholqr", "scholqr3"):
                t = c.timed_us(lambda: d.drv_qr_linops(ctx, alg, op, block_size=block, d_factor=d_factor, nnz=sketch_nnz))     # Q-less, as timed in the reference
                q = d.drv_qr_linops(ctx, alg, op, block_size=block, want_Q=True, d_factor=d_factor, nnz=sketch_nnz)
                row += list(_quality(q["Q"], n)) + [t] if q["rc"] == 0 else [float("nan"), 0, 0, t]
                del q
            # dense-operand CQRRT on the materialised matrix (the reference's CQRRT_expl column)
            import scipy.sparse as sp
            Ad = torch.zeros((n, m), dtype=torch.float64, device="cuda:0")
            Ad[torch.as_tensor(cols, device="cuda:0"), torch.arange(m, device="cuda:0").repeat_interleave(r)] += op.vals
            hold = {}
            t = c.timed_us(lambda: hold.update(o=d.drv_cqrrt(ctx, Ad, m, n, d_factor, sketch_nnz)))
            row += list(_quality(Ad, n)) + [t]
            del Ad
            row += [0, kb(int(d_factor * n) * n + 2 * n * n + m * b_eff), 0, kb(n * n + m * b_eff), 0, kb(3 * n * n + (m + n) * b_eff), 0, kb(m * n + int(d_factor * n) * n)]
            with open(path, "a") as f:
                f.write(",".join(str(x) for x in row) + "\n")
    return path


def composite_applications(argv):
    """<precision> <dir> <num_runs> <K: path.mtx | gen:<m>> <V: path.mtx | gen:<n>:<nnz_per_row>> <d_factor> [sketch_nnz] [block_size] [skip_apps] [compute_cond]
    (CQRRT_linop_composite_applications.cc: the generalized SVD / generalized least-squares study).

    Pipeline as in the reference (:3-10): K (m x m SPD) = L L^T; the operator L^{-1} V (V sparse, m x n) is held as
    CompositeOperator(L^{-1}, V) and factored Q-less by CQRRT_linops, CholQR_linops, sCholQR3_linops and sCholQR3_linops_basic; then
    (a) generalized least squares min_x ||V x - b||_{K^-1} through R, (b) generalized singular values = singular values of R,
    (c) generalized singular vectors = full SVD of R.  The reference's L^{-1} is a sparse-Cholesky SOLVE operator from its Eigen
    extras (out of scope); here L^{-1} is formed explicitly once (dense, in HBM, outside every timed region -- its cost is
    the chol_time_us column) and enters as a DenseLinOp.  Outputs `<stamp>_gsvd_results.csv` and `<stamp>_gsvd_breakdown.csv` with the
    reference's headers and columns; the per-subroutine breakdown of the linop drivers is not exposed through the C ABI, so each
    algorithm's row carries its total in the slot the reference uses for it and zeros elsewhere; peak RSS (a host notion) is 0."""
    import ctypes as C
    import scipy.io
    import scipy.sparse as sp

    precision, directory, num_runs = argv[0], argv[1], int(argv[2])
    K_spec, V_spec, d_factor = argv[3], argv[4], float(argv[5])
    sketch_nnz = int(argv[6]) if len(argv) > 6 else 4
    block = int(argv[7]) if len(argv) > 7 else 0
    skip_apps = bool(int(argv[8])) if len(argv) > 8 else False
    compute_cond = bool(int(argv[9])) if len(argv) > 9 else False
    if precision != "double":
        raise SystemExit("the device linop drivers behind this study are wired for double")
    rng = np.random.default_rng(0)
    if K_spec.startswith("gen:"):
        m = int(K_spec.split(":")[1])
        # banded SPD: a 1-D Laplacian-like stencil with a random positive diagonal shift
        K = sp.diags([-np.ones(m - 1), 2.5 + rng.random(m), -np.ones(m - 1)], [-1, 0, 1], format="csr")
    else:
        K = sp.csr_matrix(scipy.io.mmread(K_spec))
        m = K.shape[0]
    if V_spec.startswith("gen:"):
        _, n, r = V_spec.split(":")
        n, r = int(n), int(r)
        cols = rng.integers(0, n, size=(m, r))
        V = sp.csr_matrix((rng.standard_normal(m * r), (np.repeat(np.arange(m), r), cols.ravel())), shape=(m, n))
    else:
        V = sp.csr_matrix(scipy.io.mmread(V_spec))
        n = V.shape[1]
    V.sum_duplicates()
    assert V.shape[0] == m, "K and V must have the same number of rows"
    ctx = d.Context(0)
    dev = "cuda:0"
    # ---- K = L L^T, L^{-1} (timed once, shared by every algorithm: the chol_time_us column)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Kd = torch.as_tensor(K.toarray(), device=dev)
    L = torch.linalg.cholesky(Kd)
    Linv = torch.linalg.solve_triangular(L, torch.eye(m, dtype=torch.float64, device=dev), upper=False)
    torch.cuda.synchronize()
    chol_time_us = int((time.perf_counter() - t0) * 1e6)
    del Kd, L
    Linv_cm = Linv.T.contiguous()                           # column-major m x m held as the (m, m) tensor of its transpose
    left = d.DenseOperator(Linv_cm, m, m)
    right = d.CsrOperator.from_scipy(V)
    op = (left, right)
    Vd = torch.as_tensor(V.toarray(), device=dev)           # (m, n) row-major, for the reference quantities only
    LiV = Linv @ Vd                                         # L^{-1} V (m, n)
    if compute_cond:
        sv = torch.linalg.svdvals(LiV)
        print(f"  cond(L^-1 V) = {float(sv[0] / sv[-1]):.3e}")
    x_true = torch.as_tensor(rng.standard_normal(n), device=dev)
    b = Vd @ x_true
    stamp = time.strftime("%Y%m%d_%H%M%S")
    results = c.out_path(directory, f"{stamp}_gsvd_results.csv")
    breakdown = c.out_path(directory, f"{stamp}_gsvd_breakdown.csv")
    common = (f"# Date: {time.ctime()}\n# Matrix dimensions: m={m} n={n}\n# Runs per algorithm: {num_runs}\n# OpenMP threads: 0 (device: MI355X)\n"
              f"# K_file: {K_spec}\n# V_file: {V_spec}\n# d_factor: {d_factor:g}\n# sketch_nnz: {sketch_nnz}\n# block_size: {block}\n"
              f"# skip_apps: {int(skip_apps)}\n# compute_cond: {int(compute_cond)}\n")
    with open(results, "w") as f:
        f.write("# GSVD Benchmark results\n" + common +
                "m,n,run,algorithm,chol_time_us,qr_time_us,orth_error,max_orth_cols,app_a_time_us,ls_rel_error,app_b_time_us,"
                "app_c_time_us,right_svec_orth_error,total_a_time_us,total_b_time_us,total_c_time_us,peak_rss_kb,analytical_kb\n")
    with open(breakdown, "w") as f:
        f.write("# GSVD Benchmark runtime breakdown\n" + common + "# Times are in microseconds\n"
                "# CQRRT_linop breakdown (11): alloc, sketch, qr, tri_inv, fwd, adj, trmm, chol, finalize, rest, total\n"
                "# CholQR breakdown (6): alloc, fwd, adj, chol, rest, total\n"
                "# sCholQR3 breakdown (18): alloc, fwd1, adj1, chol1, upd1, fwd2, adj2, gemm2, chol2, upd2, fwd3, adj3, gemm3, chol3, upd3, q_mat, rest, total\n"
                "# sCholQR3_basic breakdown (15): alloc, fwd1, adj1, chol1, trsm1, fwd_q, syrk2, chol2, upd2, syrk3, chol3, upd3, q_mat, rest, total\n"
                "m,n,run,algorithm" + "".join(f",t{i}" for i in range(18)) + "\n")
    b_eff = n if block <= 0 or block >= n else block
    kb = lambda elems: int(elems * 8 / 1024)
    algs = [("CQRRT_linop", "cqrrt", 11, kb(int(d_factor * n) * n + 2 * n * n + m * b_eff)), ("CholQR", "cholqr", 6, kb(n * n + m * b_eff)),
            ("sCholQR3", "scholqr3", 18, kb(3 * n * n + (m + n) * b_eff)), ("sCholQR3_basic", "scholqr3_basic", 15, kb(m * n + 2 * n * n))]
    d.drv_qr_linops(ctx, "cqrrt", op, block_size=block, d_factor=d_factor, nnz=sketch_nnz, key=(123, 0))        # warm-up (:437-446)

    def svd_small(Rt, vectors):
        """singular values (and vectors) of the n x n factor on the device library; Rt = column-major tensor"""
        A = Rt.clone()
        S = torch.zeros(n, dtype=torch.float64, device=dev)
        U = torch.zeros((n, n), dtype=torch.float64, device=dev)
        VT = torch.zeros((n, n), dtype=torch.float64, device=dev)
        sw = C.c_int(0)
        rc = ctx.lib.rlhip_gesdd_f64(ctx.h, n, n, A.data_ptr(), n, S.data_ptr(), U.data_ptr(), n, VT.data_ptr(), n, C.byref(sw))
        assert rc == 0, rc
        return (S, U, VT) if vectors else S

    for name, alg, nslots, akb in algs:
        for run in range(num_runs):
            hold = {}
            qr_us = c.timed_us(lambda: hold.update(o=d.drv_qr_linops(ctx, alg, op, block_size=block, d_factor=d_factor, nnz=sketch_nnz, key=(123 + run, 0))))
            o = hold["o"]
            Rt = o["R"]                                      # column-major n x n as an (n, n) tensor: Rt[j, i] = R[i, j]
            R = torch.triu(Rt.T)
            # Q = (L^{-1} V) R^{-1} (compute_Q_from_R): orthogonality of what the factor implies
            Q = torch.linalg.solve_triangular(R, LiV, upper=True, left=False)
            err, max_cols, _ = _quality(Q.T.contiguous(), n)
            row = [m, n, run, name, chol_time_us, qr_us, f"{err:.6e}", max_cols]
            if not skip_apps:
                def app_a():                                 # x = R^{-1} R^{-T} V^T K^{-1} b,  K^{-1} b = L^{-T} (L^{-1} b)
                    w = Linv.T @ (Linv @ b)
                    rhs = Vd.T @ w
                    y = torch.linalg.solve_triangular(R.T, rhs[:, None], upper=False)
                    hold["x"] = torch.linalg.solve_triangular(R, y, upper=True)[:, 0]
                a_us = c.timed_us(app_a)
                ls_err = float(torch.linalg.norm(hold["x"] - x_true) / torch.linalg.norm(x_true))
                b_us = c.timed_us(lambda: svd_small(Rt, False))
                c_us = c.timed_us(lambda: hold.update(svd=svd_small(Rt, True)))
                VT = hold["svd"][2]
                rs_err = float(torch.linalg.norm(VT @ VT.T - torch.eye(n, dtype=torch.float64, device=dev)))
            else:
                a_us = b_us = c_us = 0
                ls_err = rs_err = 0.0
            row += [a_us, f"{ls_err:.6e}", b_us, c_us, f"{rs_err:.6e}", qr_us + a_us, qr_us + b_us, qr_us + c_us, 0, akb]
            with open(results, "a") as f:
                f.write(",".join(str(x) for x in row) + "\n")
            slots = [0] * 18
            slots[nslots - 1] = qr_us
            with open(breakdown, "a") as f:
                f.write(",".join(str(x) for x in [m, n, run, name] + slots) + "\n")
    return results, breakdown


MAINS = {"basic": basic, "composite_applications": composite_applications}

if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in MAINS:
        print(__doc__)
        sys.exit(1)
    print(MAINS[sys.argv[1]](sys.argv[2:]))
