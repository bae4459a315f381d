"""A/B of the leading dimension at C3's shape: 1048576 rows with ld = 2^20 against a padded ld (column stride no longer a power of two)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n = 1048576, 1024
rng = np.random.default_rng(0)
U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
Ud = d.cm_from_numpy(U)
for pad in (0, 32, 64, 160):
    ld = m + pad
    A = torch.empty((n, ld), dtype=torch.float64, device="cuda"); ctx.fill_dense(A, ld, n, key=(3, 0))
    G = d.cm_zeros(n, n)
    ctx.syrk("U", "T", n, m, 1.0, A, ld, 0.0, G, n); ctx.sync()
    ctx.timer_start()
    for _ in range(3): ctx.syrk("U", "T", n, m, 1.0, A, ld, 0.0, G, n)
    t1 = ctx.timer_stop_ms() / 3
    ctx.trsm(m, n, 1.0, Ud, n, A, ld); ctx.sync()
    ctx.timer_start()
    for _ in range(3): ctx.trsm(m, n, 1.0, Ud, n, A, ld)
    t2 = ctx.timer_stop_ms() / 3
    print(f"pad {pad}: syrk {t1:.2f} ms  trsm {t2:.2f} ms", flush=True)
