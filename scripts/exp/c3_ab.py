"""C3 CQRRPT (1048576 x 1024 fp64) best-of-N ms for a given library build, same box A/B.  usage: c3_ab.py <lib.so> [steps]"""
import os, sys, pathlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n, nnz = 1048576, 1024, 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
A = d.cm_empty(m, n)
ts = []
for it in range(steps + 1):
    ctx.fill_dense(A, m, n, key=(0, 0)); ctx.sync()
    t0 = time.perf_counter(); r = d.drv_cqrrpt(ctx, A, m, n, 1.25, nnz); ctx.sync(); ts.append(time.perf_counter() - t0)
ts = sorted(ts[1:])
print(os.path.basename(sys.argv[1]), f"best {ts[0] * 1e3:.3f} ms  median {ts[len(ts) // 2] * 1e3:.3f} ms", flush=True)
