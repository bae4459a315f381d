"""Two-level Cholesky factorization of n x n Gram matrices on the device: ms per call (host-synchronous entry, as CQRRPT / BQRRP call it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from randlapack_amd import _lib
if len(sys.argv) > 1:
    import pathlib; _lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
from randlapack_amd import device as d
ctx = d.Context(0)
for dt, nm in ((torch.float64, "f64"), (torch.float32, "f32")):
    for n in (256, 448, 1024, 2048, 4096):
        X = torch.randn((n, 2 * n), dtype=dt, device="cuda")
        G = (X @ X.T).contiguous()
        best = 1e9
        for it in range(12):
            Gd = G.clone(); ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = ctx.potrf(n, Gd, n); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        R = torch.triu(Gd.T.clone())    # column-major storage viewed row-major: transpose
        err = float((R.T @ R - G).abs().max() / G.abs().max())
        print(f"{nm} n {n}: {best * 1e3:.3f} ms  info {rc}  |R^T R - G| / |G| {err:.2e}", flush=True)
