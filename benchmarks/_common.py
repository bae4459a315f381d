"""Shared plumbing of the benchmark mains (reference: benchmark/bench_BQRRP/*.cc, benchmark/bench_CQRRPT/*.cc).

The mains reproduce the reference's text-file formats (same file names, same header block of `num_info_lines`, same column order,
microseconds) so that its plotting scripts keep working; what runs underneath is the device library.  Timing follows the
reference: wall clock around each call (steady_clock there, perf_counter + device sync here), every run recorded."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from randlapack_amd import device as d

MAT_TYPE_IDS = d.MAT_TYPES            # printed as std::to_string(m_info.m_type)


def out_path(directory: str, filename: str) -> str:
    return filename if directory == "." else os.path.join(directory, "") + filename


def timed_us(fn) -> int:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return int(round((time.perf_counter() - t0) * 1e6))


def regen(ctx, m_type, m, n, key=(0, 0), dtype=torch.float64, **kw):
    """data_regen(): the input is re-generated in HBM before every run (the factorizations overwrite it)"""
    return d.drv_mat_gen(ctx, m_type, m, n, key=key, dtype=dtype, **kw)["A"]


def geqrf(ctx, A, m, n):
    tau = torch.zeros(min(m, n), dtype=A.dtype, device=A.device)
    rc = getattr(ctx.lib, f"rlhip_geqrf_{d._suffix(A)[0]}")(ctx.h, m, n, A.data_ptr(), m, tau.data_ptr())
    assert rc == 0, rc
    return tau


def geqp3(ctx, A, m, n):
    tau = torch.zeros(min(m, n), dtype=A.dtype, device=A.device)
    J = torch.zeros(n, dtype=torch.int64, device=A.device)
    rc = getattr(ctx.lib, f"rlhip_geqp3_{d._suffix(A)[0]}")(ctx.h, m, n, A.data_ptr(), m, J.data_ptr(), tau.data_ptr())
    assert rc == 0, rc
    return J, tau


def upper_factor(A, m, n):
    """R (n x n numpy, upper triangle) out of a GEQP3-format column-major device tensor (n, m)"""
    k = min(m, n)
    return np.triu(A[:, :k].T.cpu().numpy())[:k, :]


def trailing_norms(R):
    """||R[i:, i:]||_F for every i (get_norms(), BQRRP_pivot_quality.cc:102-114: lantr on the trailing blocks)"""
    sq = np.triu(R) ** 2
    col = np.cumsum(sq[::-1, :], axis=0)[::-1, :]            # col[i, c] = sum_{r >= i} R[r, c]^2
    n = R.shape[1]
    return np.sqrt(np.array([col[i, i:].sum() for i in range(n)]))


def singular_values(ctx, A, m, n):
    """all singular values of the (m x n, m >= n) device matrix, on the device (the reference calls gejsv)"""
    cpy = A.clone()
    S = torch.zeros(n, dtype=A.dtype, device=A.device)
    VT = torch.zeros((n, n), dtype=A.dtype, device=A.device)
    import ctypes as C
    sw = C.c_int(0)
    rc = getattr(ctx.lib, f"rlhip_gesvdj_{d._suffix(A)[0]}")(ctx.h, m, n, cpy.data_ptr(), m, S.data_ptr(), VT.data_ptr(), n, C.byref(sw))
    assert rc == 0, rc
    return S.cpu().numpy()
