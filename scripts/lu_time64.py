"""Time the row-pivoted LU of BQRRP's transposed sketch in fp64 (n x 512, pivots only) and check the pivots against LAPACK."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.linalg.lapack as ll
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n) in ((16384, 512), (32768, 512), (8192, 512), (3000, 96)):
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m, n))
    ip = torch.zeros(n, dtype=torch.int64, device="cuda")
    ts = []
    for it in range(3):
        B = d.cm_from_numpy(A)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = ctx.lib.rlhip_getrf_f64(ctx.h, m, n, B.data_ptr(), m, ip.data_ptr())
        ctx.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lu_ref, piv_ref, info = ll.dgetrf(A)
    ok = np.array_equal(ip.cpu().numpy() - 1, piv_ref)
    err = np.abs(d.cm_to_numpy(B) - lu_ref).max() / np.abs(lu_ref).max()
    print(f"getrf {m}x{n} f64: {min(ts)*1e3:.2f} ms ({min(ts)*1e6/n:.1f} us/column all-in) rc={rc} pivots==LAPACK {ok} LU err {err:.1e}", flush=True)
