"""The small mains of benchmark/bench_general and the two side checks of benchmark/bench_BQRRP, on the device library.

  python -m benchmarks.general chol_check                                   Chol_check.cc:8-58
        ten seeds: a 1000 x 1000 polynomial matrix (cond 1e8) whose leading 500 x 500 block is replaced by a Gram matrix; potrf on the whole
        (indefinite) matrix stops early "as expected" and the leading block of the factor still satisfies R'R = A[:k, :k]
  python -m benchmarks.general gemm_vs_ormqr                                Gemm_vs_ormqr.cc:15-79
        Q^T B through the implicit factor (larft + compact-WY apply: what the C++ layer's ormqr composes) against ungqr + GEMM; 2^10 x 2^5 ... 2^15 x 2^10
  python -m benchmarks.general basic_blas_speed [n_start = 1024] [n_stop = 16384] [numruns = 5]      basic_blas_speed.cc:69-145
        BLAS-1 / -2 / -3 microseconds -> BLAS_performance_comp_col_start_<n_start>_col_stop_<n_stop>.dat
  python -m benchmarks.general convert_time <file>                          convert_time.cc:18-57   (microseconds -> seconds, in place)
  python -m benchmarks.general hqrrp_sanity_check <dir> <num_runs> <m1> [m2 ...]                    HQRRP_sanity_check.cc:61-185
        HQRRP (block 128, d factor 1, no panel pivoting) against an m x m x m GEMM -> _HQRRP_GEMM_speed_comparisons_mat_size_num_info_lines_7.txt
  python -m benchmarks.general find_test_mat_spectrum <dir> <num_rows> <num_cols>                   find_test_mat_spectrum.cc:60-180
        singular values (one-sided Jacobi SVD on the device; the reference calls gejsv) of the polynomial / staircase / spiked / Kahan test
        matrices -> _{poly,stair,spike,kahan}_spectrum_num_info_lines_4.txt

The device library has no separate gemv / axpy entry: a matrix-vector product is the n = 1 GEMM and y <- y - x the k = 1 GEMM with a 1 x 1 unit
operand, which is what these two lines time."""
from __future__ import annotations

import sys
import time

import numpy as np
import torch

from randlapack_amd import device as d

from . import _common as c


def _lib(ctx, name, A, *args):
    rc = getattr(ctx.lib, f"rlhip_{name}_{d._suffix(A)[0]}")(ctx.h, *args)
    return rc


def chol_check(m=1000, k=500, seeds=10):
    ctx = d.Context(0)
    out = []
    for i in range(seeds):
        A = c.regen(ctx, "polynomial", m, m, key=(i, 0), cond_num=1e8)
        G = d.cm_zeros(k, k)
        ctx.syrk("U", "T", k, k, 1.0, A, k, 0.0, G, k)                   # (the reference reads A with leading dimension k here: Chol_check.cc:27)
        Gs = torch.triu(G.T)                                              # numpy-style view: G.T[i, j] = entry (i, j)
        Gs = Gs + torch.triu(Gs, 1).T                                    # full symmetric k x k block
        A.T[:k, :k] = Gs                                                 # leading block of A symmetric, the rest random
        rc = ctx.lib.rlhip_potrf_f64(ctx.h, b"U", m, A.data_ptr(), m)
        if rc != 0:
            print("Cholesky failed as expected.")
        R = torch.triu(A.T[:k, :k]).contiguous()
        Rd = d.cm_from_numpy(R.cpu().numpy())
        Gd = d.cm_from_numpy(Gs.cpu().numpy())
        ctx.gemm("T", "N", k, k, k, 1.0, Rd, k, Rd, k, -1.0, Gd, k)
        nrm = ctx.lange_fro(k, k, Gd, k)
        print(f"||R[:k, :k]'*R[:k, :k] - A[:k, :k]||_F:  {nrm:e}")
        out.append((rc, nrm))
    return out


def gemm_vs_ormqr(sizes=((2**10, 2**5), (2**11, 2**6), (2**12, 2**7), (2**13, 2**8), (2**14, 2**9), (2**15, 2**10)), runs=10):
    ctx = d.Context(0)
    rows = []
    for (m, n) in sizes:
        g_rate = o_rate = 0.0
        for i in range(runs):
            A = c.regen(ctx, "gaussian", m, n)
            B1 = c.regen(ctx, "gaussian", m, n, key=(1, 0))
            B2 = B1.clone()
            tau = c.geqrf(ctx, A, m, n)
            Tm = d.cm_zeros(n, n)
            P = d.cm_zeros(n, n)

            def ormqr():
                assert _lib(ctx, "larft", A, m, n, A.data_ptr(), m, tau.data_ptr(), Tm.data_ptr(), n) == 0
                assert _lib(ctx, "gemqrt", A, b"L", b"T", m, n, n, n, A.data_ptr(), m, Tm.data_ptr(), n, B1.data_ptr(), m) == 0
            dur_ormqr = c.timed_us(ormqr)

            def gemm():
                assert _lib(ctx, "ungqr", A, m, n, n, A.data_ptr(), m, tau.data_ptr()) == 0
                ctx.gemm("T", "N", n, n, m, 1.0, A, m, B2, m, 0.0, P, n)
            dur_gemm = c.timed_us(gemm)
            gflop = 2.0 * n * n * m / 1e9
            if i != 0:                                                    # (the reference's "rate" is GFLOP per MICROsecond: Gemm_vs_ormqr.cc:60-63)
                g_rate += gflop / dur_gemm
                o_rate += gflop / dur_ormqr
        print(f"{g_rate / (runs - 1):g}  {o_rate / (runs - 1):g}")
        rows.append((m, n, g_rate / (runs - 1), o_rate / (runs - 1)))
    return rows


def basic_blas_speed(n_start=1024, n_stop=16384, numruns=5, directory="."):
    ctx = d.Context(0)
    path = c.out_path(directory, f"BLAS_performance_comp_col_start_{n_start}_col_stop_{n_stop}.dat")
    one = d.cm_from_numpy(np.ones((1, 1)))
    n = n_start
    while n <= n_stop:
        A = c.regen(ctx, "gaussian", n, n)
        B = c.regen(ctx, "gaussian", n, n)
        Cm = d.cm_zeros(n, n)
        a = d.cm_empty(n, 1); ctx.fill_dense(a, n, 1, key=(3, 0))
        b = d.cm_empty(n, 1); ctx.fill_dense(b, n, 1, key=(4, 0))
        for i in range(numruns):
            print(f"ITERATION {i}, DIM {n}")
            dur3 = c.timed_us(lambda: ctx.gemm("N", "N", n, n, n, 1.0, A, n, B, n, 0.0, Cm, n))
            dur2 = c.timed_us(lambda: ctx.gemm("N", "N", n, 1, n, 1.0, A, n, a, n, 1.0, b, n))          # gemv: b <- A a + b
            dur1 = c.timed_us(lambda: ctx.gemm("N", "N", n, 1, 1, -1.0, a, n, one, 1, 1.0, b, n))       # axpy: b <- b - a
            with open(path, "a") as f:
                f.write(f"{n},  {dur1},  {dur2},  {dur3},\n")
        del A, B, Cm
        n *= 2
    return path


def convert_time(filename):
    """every whitespace-separated entry of every line divided by 1e6, written back in the reference's layout (a blank first line, two spaces)"""
    with open(filename) as f:
        lines = [ln.split() for ln in f if ln.split()]
    with open(filename, "w") as f:
        for ln in lines:
            f.write("\n" + "".join(f"{float(x) / 1e6:f}  " for x in ln))
    return filename


def hqrrp_sanity_check(argv):
    directory, numruns = argv[0], int(argv[1])
    m_sz = [int(x) for x in argv[2:]]
    ctx = d.Context(0)
    path = c.out_path(directory, "_HQRRP_GEMM_speed_comparisons_mat_size_num_info_lines_7.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the sanity check for the odd performance of HQRRP."
                "\nFile format: 2 columns, containing time for each algorithm: HQRRP, GEMM;"
                "\nrows correspond to BQRRP runs with varying mat sizes, with numruns repititions of each mat size."
                "\nNum OMP threads:0 (device: MI355X)"
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput row sizes:{''.join(str(x) + ', ' for x in m_sz)}"
                f"\nAdditional parameters: HQRRP columns/block size: 128 num runs per size {numruns} HQRRP d factor: 1.0\n")
    for m in m_sz:
        for i in range(numruns):
            print(f"ITERATION {i}, ROWS {m}")
            A = c.regen(ctx, "gaussian", m, m)
            dur_hqrrp = c.timed_us(lambda: d.drv_hqrrp(ctx, A, m, m, nb_alg=128, pp=0, panel_pivoting=0, qr_type=0))
            print(f"TOTAL TIME FOR HQRRP {dur_hqrrp}")
            A = c.regen(ctx, "gaussian", m, m)
            B = c.regen(ctx, "gaussian", m, m, key=(1, 0))
            Cm = d.cm_zeros(m, m)
            dur_gemm = c.timed_us(lambda: ctx.gemm("N", "N", m, m, m, 1.0, A, m, B, m, 0.0, Cm, m))
            print(f"TOTAL TIME FOR GEMM {dur_gemm}")
            with open(path, "a") as f:
                f.write(f"{dur_hqrrp},  {dur_gemm},\n")
            del A, B, Cm
    return path


def find_test_mat_spectrum(argv):
    directory, m, n = argv[0], int(argv[1]), int(argv[2])
    ctx = d.Context(0)
    cases = (("poly", "polynomial", dict(cond_num=1e10, exponent=2.0)), ("stair", "step", dict(cond_num=1e10)),
             ("spike", "spiked", dict(scaling=1e10)), ("kahan", "kahan", dict(theta=1.2, perturb=1e3)))
    paths = []
    for tag, mt, kw in cases:
        A = c.regen(ctx, mt, m, n, **kw)
        sv = c.singular_values(ctx, A, m, n)
        path = c.out_path(directory, f"_{tag}_spectrum_num_info_lines_4.txt")
        with open(path, "a") as f:
            f.write("Description: Spectrum of the matrix of a given type generated in RandLAPACK found via Jacobi SVD"
                    "\nNum OMP threads:0 (device: MI355X)"
                    f"\nInput type:{c.MAT_TYPE_IDS[mt]}"
                    f"\nInput size:{m} by {n}\n")
            f.write("".join(f"{x:g},  " for x in sv) + "\n")
        print(f"Done with the {mt} matrix")
        paths.append(path)
    return paths


if __name__ == "__main__":
    av = sys.argv[1:]
    if not av:
        print(__doc__); sys.exit(1)
    if av[0] == "chol_check": chol_check()
    elif av[0] == "gemm_vs_ormqr": gemm_vs_ormqr()
    elif av[0] == "basic_blas_speed": print(basic_blas_speed(*[int(x) for x in av[1:4]]))
    elif av[0] == "convert_time" and len(av) == 2: print(convert_time(av[1]))
    elif av[0] == "hqrrp_sanity_check" and len(av) >= 4: print(hqrrp_sanity_check(av[1:]))
    elif av[0] == "find_test_mat_spectrum" and len(av) == 4: print(find_test_mat_spectrum(av[1:]))
    else:
        print(__doc__); sys.exit(1)
