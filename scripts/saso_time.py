"""Time S*A for the CQRRPT sketch of BASELINE configs[2] (S: 1280 x 1048576 SASO with 4 nonzeros per column, A: 1048576 x 1024 fp64)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd import _lib
import pathlib
if os.environ.get("RLHIP_EXP_LIB"): _lib.LIB_PATH = pathlib.Path(os.environ["RLHIP_EXP_LIB"]).resolve()
from randlapack_amd import device as d
ctx = d.Context(0)
m, n, dd, nnz = 1048576, 1024, 1280, 4
A = torch.randn((n, m), dtype=torch.float64, device="cuda")          # column-major m x n
B = torch.zeros((n, dd), dtype=torch.float64, device="cuda")
u32 = lambda t: (C.c_uint32 * len(t))(*t)
for mode in (1, 0):
    S = C.c_void_p(); nxt = (C.c_uint32 * 4)()
    assert ctx.lib.rlhip_saso_create_mode(ctx.h, dd, m, nnz, mode, u32((0, 0, 0, 0)), u32((7, 0)), nxt, C.byref(S)) == 0
    ts = []
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = ctx.lib.rlhip_saso_apply_f64(ctx.h, S, n, 1.0, A.data_ptr(), m, 0.0, B.data_ptr(), dd)
        ctx.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        assert rc == 0
    print(f"mode {mode} dbg {os.environ.get('RLHIP_SASO_DBG', '0')}: {min(ts)*1e3:.2f} ms = {8*m*n/min(ts)/1e12:.2f} TB/s", flush=True)
    ctx.lib.rlhip_saso_destroy(ctx.h, S)
