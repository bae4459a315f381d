"""CQRRT_linops scaling study on the device library (benchmark/bench_CQRRT_linops/CQRRT_linop_basic.cc).

  python -m benchmarks.cqrrt_linops basic <dir> <num_sizes> <num_runs> <m_start> <m_end> <aspect_ratio> <nnz_per_row> <d_factor> [sketch_nnz] [block_size]

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


MAINS = {"basic": basic}

if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in MAINS:
        print(__doc__)
        sys.exit(1)
    print(MAINS[sys.argv[1]](sys.argv[2:]))
