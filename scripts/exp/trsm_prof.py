"""Per-phase wall-clock profile of ONE workgroup of the fused in-place solve at C3's shape (needs a library built with -DRLHIP_TF_PROF)."""
import os, sys, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n = 1048576, 1024
A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(3, 0)); U = d.cm_empty(n, n); ctx.fill_dense(U, n, n, key=(2, 0))
ctx.lib.rlhip_add_diag_f64(ctx.h, n, 40.0, U.data_ptr(), n)
for _ in range(3):
    ctx.timer_start(); ctx.trsm(m, n, 1.0, U, n, A, m); print("ms", round(ctx.timer_stop_ms(), 3), flush=True)
