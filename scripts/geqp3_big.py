"""One-off: lapack::geqp3's device counterpart on matrices that do not fit LDS (the QP3 baseline column of the BQRRP benchmark files)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n) in ((512, 16384), (16384, 16384)):
    A = torch.randn((n, m), dtype=torch.float64, device="cuda")
    J = torch.zeros(n, dtype=torch.int64, device="cuda"); tau = torch.zeros(min(m, n), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = ctx.lib.rlhip_geqp3_f64(ctx.h, m, n, A.data_ptr(), m, J.data_ptr(), tau.data_ptr())
    ctx.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"geqp3 {m}x{n} f64: {dt*1e3:.1f} ms ({dt*1e6/min(m,n):.1f} us/column) rc={rc}", flush=True)
