"""Time lapack::geqrf's device counterpart on BQRRP's pivoted sketch shapes (d x cols, wide) in fp32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n) in ((2048, 65536), (2048, 32768), (2048, 2048), (512, 16384)):
    A = torch.randn((n, m), dtype=torch.float32, device="cuda")
    tau = torch.zeros(min(m, n), dtype=torch.float32, device="cuda")
    ts = []
    for it in range(3):
        B = A.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = ctx.lib.rlhip_geqrf_f32(ctx.h, m, n, B.data_ptr(), m, tau.data_ptr())
        ctx.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"geqrf {m}x{n} f32: {min(ts)*1e3:.2f} ms rc={rc}", flush=True)
