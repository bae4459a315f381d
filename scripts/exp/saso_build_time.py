"""Time to build the C3 sparse-sign operator (1280 x 1048576, 4 nonzeros per column) on the device."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import device as d
ctx = d.Context(0)
m, dd, nnz = 1048576, 1280, 4
u32 = lambda t: (C.c_uint32 * len(t))(*t)
for it in range(5):
    S = C.c_void_p(); nxt = (C.c_uint32 * 4)()
    ctx.sync(); t0 = time.perf_counter()
    assert ctx.lib.rlhip_saso_create_mode(ctx.h, dd, m, nnz, 1, u32((0, 0, 0, 0)), u32((7, 0)), nxt, C.byref(S)) == 0
    ctx.sync(); tb = time.perf_counter() - t0
    ctx.lib.rlhip_saso_destroy(ctx.h, S)
    print(f"operator build {tb * 1e3:.3f} ms", flush=True)
