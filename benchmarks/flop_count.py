"""The reference's two "what can this system do" mains on the device library (benchmark/bench_general/GEMM_flop_count.cc,
LAPACK_flop_count.cc): same flop formulas (LAWN 41), same best-of-N protocol, same output lines.

  python -m benchmarks.flop_count gemm [k = 10000] [runs = 50]              GEMM_flop_count.cc:14-56     (2 k^3 flops, fp64)
  python -m benchmarks.flop_count geqrf|getrf|potrf <rows> <cols> <numruns>  LAPACK_flop_count.cc:16-168

The inputs are regenerated in HBM before every run (the factorizations overwrite them), Gaussian as in the reference; potrf factors
G^T G + n I of a Gaussian G (the reference hands potrf a Gaussian matrix, which is not positive definite: its dpotrf stops at the first
column -- SURVEY.md appendix B -- so the number it prints is not a Cholesky rate)."""
from __future__ import annotations

import sys

import torch

from randlapack_amd import device as d

from . import _common as c


def gemm_flops(k=10000, runs=50):
    ctx = d.Context(0)
    flop_cnt = 2.0 * k**3
    best = 0.0
    C = d.cm_empty(k, k)
    for i in range(runs):
        A = c.regen(ctx, "gaussian", k, k, key=(2 * i, 0))
        B = c.regen(ctx, "gaussian", k, k, key=(2 * i + 1, 0))
        us = c.timed_us(lambda: ctx.gemm("N", "N", k, k, k, 1.0, A, k, B, k, 0.0, C, k))
        best = max(best, flop_cnt / (us * 1e-6) / 1e9)
    print(f"THE SYSTEM IS CAPABLE OF {best:g} GFLOPs/sec.\n")
    return best


def geqrf_flops(rows, cols, numruns):
    ctx = d.Context(0)
    if rows >= cols:                                   # LAWN 41 (LAPACK_flop_count.cc:26-30)
        flop_count = 2.0 * rows * cols**2 - (2.0 / 3.0) * cols**3 + rows * cols + cols**2 + (14.0 / 3.0) * cols
    else:
        flop_count = 2.0 * cols * rows**2 - (2.0 / 3.0) * rows**3 + 3.0 * rows * cols - rows**2 + (14.0 / 3.0) * cols
    print(f"{flop_count:g}")
    best_us = None
    for i in range(numruns):
        A = c.regen(ctx, "gaussian", rows, cols, key=(i, 0))
        us = c.timed_us(lambda: c.geqrf(ctx, A, rows, cols))
        best_us = us if best_us is None else min(best_us, us)
    rate = flop_count / (best_us * 1e-6) / 1e9
    print(f"THE SYSTEM IS CAPABLE OF {rate:g} GFLOPs/sec RUNNING GEQRF.")
    return rate


def getrf_flops(rows, cols, numruns):
    ctx = d.Context(0)
    flop_count = rows * cols**2 - (1.0 / 3.0) * cols**3 - 0.5 * cols**2 + (5.0 / 6.0) * cols        # :72
    best_us = None
    ip = torch.zeros(min(rows, cols), dtype=torch.int64, device="cuda")
    for i in range(numruns):
        A = c.regen(ctx, "gaussian", rows, cols, key=(i, 0))
        us = c.timed_us(lambda: ctx.lib.rlhip_getrf_f64(ctx.h, rows, cols, A.data_ptr(), rows, ip.data_ptr()))
        best_us = us if best_us is None else min(best_us, us)
    rate = flop_count / (best_us * 1e-6) / 1e9
    print(f"THE SYSTEM IS CAPABLE OF {rate:g} GFLOPs/sec RUNNING GETRF.")
    return rate


def potrf_flops(dim, numruns):
    ctx = d.Context(0)
    flop_count = (1.0 / 3.0) * dim**3 + 0.5 * dim**2 + (1.0 / 6.0) * dim                              # :108
    best_us = None
    G = d.cm_empty(dim, dim)
    for i in range(numruns):
        A = c.regen(ctx, "gaussian", dim, dim, key=(i, 0))
        ctx.syrk("U", "T", dim, dim, 1.0, A, dim, 0.0, G, dim)
        ctx.lib.rlhip_add_diag_f64(ctx.h, dim, float(dim), G.data_ptr(), dim)
        us = c.timed_us(lambda: ctx.potrf(dim, G, dim))
        best_us = us if best_us is None else min(best_us, us)
    rate = flop_count / (best_us * 1e-6) / 1e9
    print(f"THE SYSTEM IS CAPABLE OF {rate:g} GFLOPs/sec RUNNING POTRF.")
    return rate


def main(argv):
    if not argv:
        print(__doc__)
        return 1
    name = argv[0].lower()
    if name == "gemm":
        gemm_flops(int(argv[1]) if len(argv) > 1 else 10000, int(argv[2]) if len(argv) > 2 else 50)
        return 0
    if len(argv) < 4:
        raise RuntimeError("Improper input provided.\n Please, specify the name of the function to be tested,the size of the input matrix and the number "
                           "of consecutive runs of the given algorithm to be performed.\nExample input: GEQRF 1000 1000 20\n")
    rows, cols, numruns = int(argv[1]), int(argv[2]), int(argv[3])
    if name in ("geqrf", "qrf"):
        geqrf_flops(rows, cols, numruns)
    elif name in ("getrf", "trf"):
        getrf_flops(rows, cols, numruns)
    elif name == "potrf":
        if rows != cols:
            print("Cholesky factorization required a square input. \n Using the smaller dimension provided.")
        potrf_flops(min(rows, cols), numruns)
    else:
        raise RuntimeError("Invalid LAPACK function name.")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
