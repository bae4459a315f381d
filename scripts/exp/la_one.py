"""One warm-up and one measured BQRRP call (look-ahead on/off by argv[4]) for a kernel trace.  usage: la_one.py m b {f32|f64} {0|1}"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from randlapack_amd import device as d
m = int(sys.argv[1]); b = int(sys.argv[2]); dt = torch.float32 if sys.argv[3] == "f32" else torch.float64
ctx = d.Context(0)
A = d.cm_empty(m, m, dtype=dt)
ctx.set_option("bqrrp_lookahead_min_elems", 0 if int(sys.argv[4]) else 1 << 62)
ctx.fill_dense(A, m, m, key=(4, 0)); ctx.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
r = d.drv_bqrrp(ctx, A, m, m, b, 1.0, timing=False, qr_tall=1, apply_trans_q=1)
torch.cuda.synchronize()
print(f"lookahead {sys.argv[4]}: {(time.perf_counter() - t0) * 1e3:.1f} ms rank {r['rank']}", flush=True)
