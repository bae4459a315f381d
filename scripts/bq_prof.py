import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randlapack_amd import device as d
m = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
ctx = d.Context(0)
A = d.cm_empty(m, m, dtype=dt)
for it in range(2):
    ctx.fill_dense(A, m, m, key=(4, 0)); ctx.sync()
    r = d.drv_bqrrp(ctx, A, m, m, b, 1.0, timing=True, qr_tall=1, apply_trans_q=1)
    print(r["times_us"])
ctx.close()
