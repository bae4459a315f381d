"""Time the row-pivoted LU of BQRRP's transposed sketch shape (n x 2048 fp32, pivots only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n) in ((65536, 2048), (32768, 2048), (8192, 2048)):
    A = torch.randn((n, m), dtype=torch.float32, device="cuda")
    ip = torch.zeros(n, dtype=torch.int64, device="cuda")
    ts = []
    for it in range(3):
        B = A.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = ctx.lib.rlhip_getrf_piv_f32(ctx.h, m, n, B.data_ptr(), m, ip.data_ptr())
        ctx.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"getrf_piv {m}x{n} f32: {min(ts)*1e3:.1f} ms ({min(ts)*1e6/n:.1f} us/column all-in) rc={rc}", flush=True)
