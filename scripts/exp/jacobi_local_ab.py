"""Device SVD of a 20000 x 256 well-conditioned factor (the RSVD tail): persistent Jacobi with the same-XCD hand-over (option 1) against the
uncached hand-over only (option 2): ms per call, bitwise comparison of U, S, V^T; both routes (Gram / classic)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n = 20000, 256
rng = np.random.default_rng(5)
A = rng.standard_normal((m, n))
for gram in (1, 0):
    ctx.set_option("gesdd_gram", gram)
    res = {}
    for mode in (1, 2, 1, 2):
        ctx.set_option("jacobi_persist", mode)
        best = 1e9
        for it in range(6):
            Ad = d.cm_from_numpy(A); S = torch.zeros(n, dtype=torch.float64, device="cuda"); U = d.cm_empty(m, n); VT = d.cm_empty(n, n); sw = C.c_int(0)
            ctx.sync(); t0 = time.perf_counter()
            assert ctx.lib.rlhip_gesdd_f64(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, C.byref(sw)) == 0
            ctx.sync(); best = min(best, time.perf_counter() - t0)
        res[mode] = (U.clone(), S.clone(), VT.clone())
        print(f"gram {gram} jacobi_persist {mode}: {best * 1e3:.3f} ms, sweeps {sw.value}", flush=True)
    print("  bitwise equal:", all(torch.equal(a, b) for a, b in zip(res[1], res[2])), flush=True)
