"""potrf of a 256 x 256 Gram matrix: ms per call for a given library build (ablation builds give wrong factors: timing only)."""
import os, sys, time, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
for dt, nm in ((torch.float64, "f64"), (torch.float32, "f32")):
    n = 256
    X = torch.randn((n, 2 * n), dtype=dt, device="cuda"); G = (X @ X.T).contiguous()
    best = 1e9
    for it in range(20):
        Gd = G.clone(); ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.potrf(n, Gd, n); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(os.path.basename(sys.argv[1]), nm, f"{best * 1e3:.3f} ms", flush=True)
