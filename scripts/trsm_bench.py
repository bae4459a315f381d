"""time B <- B inv(U) at a CQRRPT-like shape (default 1048576 x 1024 fp64); env knobs select the kernel variant"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from randlapack_amd import device as d
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
ctx = d.Context(0)
B = d.cm_empty(m, n, dtype=dt); ctx.fill_dense(B, m, n, key=(1, 0))
rng = np.random.default_rng(0)
U = np.triu(rng.standard_normal((n, n))) / np.sqrt(n) + 2 * np.eye(n)
Ud = d.cm_from_numpy(U).to(dt)
ctx.trsm(m, n, 1.0, Ud, n, B, m); ctx.sync()
ctx.fill_dense(B, m, n, key=(1, 0))
ctx.timer_start()
reps = 3
for _ in range(reps): ctx.trsm(m, n, 1.0, Ud, n, B, m)
ms = ctx.timer_stop_ms() / reps
print(f"trsm {m}x{n} {dt}: {ms:.2f} ms  {1.0*m*n*n/ms/1e9:.1f} TFLOP/s (m n^2)  variant={os.environ.get('RLHIP_TRSM_FUSED_VARIANT','default')} fused={os.environ.get('RLHIP_TRSM_FUSED','1')}", flush=True)
