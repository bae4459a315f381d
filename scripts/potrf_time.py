import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
for n in (256, 1024, 2048, 4096):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((2*n, n)); G = X.T @ X
    for env in ("1", "0"):
        import os
        ts = []
        for it in range(4):
            Gd = d.cm_from_numpy(G)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.potrf(n, Gd, n)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(n, "ms", min(ts) * 1e3)
        break
