"""Time lapack::geqp3's device counterpart on the CQRRPT sketch shape (d x n = 1280 x 1024 fp64) and a few others."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n, dt) in ((1280, 1024, np.float64), (2560, 2048, np.float64), (1280, 1024, np.float32), (512, 256, np.float64)):
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m, n)).astype(dt)
    ts = []
    fn = ctx.lib.rlhip_geqp3_f64 if dt == np.float64 else ctx.lib.rlhip_geqp3_f32
    tdt = torch.float64 if dt == np.float64 else torch.float32
    for it in range(4):
        Ad = d.cm_from_numpy(A)
        J = torch.zeros(n, dtype=torch.int64, device="cuda"); tau = torch.zeros(n, dtype=tdt, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = fn(ctx.h, m, n, Ad.data_ptr(), m, J.data_ptr(), tau.data_ptr())
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        assert rc == 0, rc
    print(f"geqp3 {m}x{n} {np.dtype(dt).name}: {min(ts)*1e3:.2f} ms  ({min(ts)*1e6/min(m,n):.1f} us/column)", flush=True)
