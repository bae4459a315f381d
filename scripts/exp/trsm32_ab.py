"""fp32 fused solve X U = B at BQRRP's panel shape (49152 x 2048, in place): ms per launch for a given library build.  usage: trsm32_ab.py <lib.so> [rows]"""
import os, sys, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n = (int(sys.argv[2]) if len(sys.argv) > 2 else 49152), 2048
A = d.cm_empty(m, n, dtype=torch.float32); ctx.fill_dense(A, m, n, key=(3, 0)); U = d.cm_empty(n, n, dtype=torch.float32); ctx.fill_dense(U, n, n, key=(2, 0))
import ctypes
ctx.lib.rlhip_add_diag_f32(ctx.h, n, ctypes.c_float(60.0), U.data_ptr(), n)
best = 1e9
ctx.trsm(m, n, 1.0, U, n, A, m); ctx.sync()
for _ in range(4):
    ctx.timer_start()
    for _ in range(5): ctx.trsm(m, n, 1.0, U, n, A, m)
    best = min(best, ctx.timer_stop_ms() / 5)
print(os.path.basename(sys.argv[1]), m, f"{best:.3f} ms = {m * n * n / best / 1e9:.1f} TFLOP/s", flush=True)
