"""C = A^T B for tall narrow operands (gemm_tn_skinny_kernel) and the Gram matrix of a tall narrow block: us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n, k) in ((32, 32, 200000), (64, 32, 200000), (64, 64, 200000), (32, 32, 1048576)):
    A = d.cm_empty(k, m); ctx.fill_dense(A, k, m, key=(1, 0)); B = d.cm_empty(k, n); ctx.fill_dense(B, k, n, key=(2, 0))
    C = d.cm_zeros(m, n)
    for name, fn in (("A^T B", lambda: ctx.gemm("T", "N", m, n, k, 1.0, A, k, B, k, 0.0, C, m)), ("A^T A", lambda: ctx.syrk("U", "T", m, k, 1.0, A, k, 0.0, C, m))):
        fn(); ctx.sync(); best = 1e9
        for _ in range(3):
            ctx.timer_start()
            for _ in range(20): fn()
            best = min(best, ctx.timer_stop_ms() / 20)
        by = (k * (m + (n if name == "A^T B" else 0))) * 8
        print(f"{name} {m} x {n} over {k} rows: {best * 1e3:.1f} us = {by / best / 1e9:.2f} TB/s", flush=True)
