import sys
import numpy as np, torch
from randlapack_amd import device as d
m, n, r = 1_000_000, 1000, 10
ctx = d.Context(0)
rng = np.random.default_rng(0)
cols = np.sort(rng.integers(0, n, size=(m, r)), axis=1).astype(np.int64).ravel()
op = d.CsrOperator(m, n, torch.as_tensor(np.arange(m + 1, dtype=np.int64) * r, device="cuda:0"), torch.as_tensor(cols, device="cuda:0"),
                   torch.as_tensor(rng.standard_normal(m * r), device="cuda:0"))
alg = sys.argv[1] if len(sys.argv) > 1 else "cqrrt"
for _ in range(3):
    d.drv_qr_linops(ctx, alg, op, block_size=0, d_factor=2.0, nnz=4)
torch.cuda.synchronize()
