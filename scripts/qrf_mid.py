import sys, time, torch, numpy as np
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n) in [(2000, 1000), (1280, 1024), (2560, 1024), (4096, 2048), (512, 256), (1024, 512)]:
    A0 = torch.randn((n, m), dtype=torch.float64, device="cuda:0")
    tau = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    best = 1e9
    for it in range(4):
        A = A0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.lib.rlhip_geqrf_f64(ctx.h, m, n, A.data_ptr(), m, tau.data_ptr()); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(m, n, f"{best*1e3:.2f} ms")
