"""A/B of the stream-K lockstep group size (RLHIP_STREAMK_GROUP, read once per process): HIP-event time of the NN and TN passes of C2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n, k = 200000, 20000, 256
A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7, 0))
Om = d.cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(0, 0))
Q = d.cm_empty(m, k); ctx.fill_dense(Q, m, k, key=(1, 0))
Y = d.cm_empty(m, k); BT = d.cm_empty(n, k)
out = []
for name, fn in (("NN", lambda: ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)), ("TN", lambda: ctx.gemm("T", "N", n, k, m, 1.0, A, m, Q, m, 0.0, BT, n))):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(4): fn()
        best = min(best, ctx.timer_stop_ms() / 4)
    out.append(f"{name} {best:.3f} ms")
print("group", os.environ.get("RLHIP_STREAMK_GROUP", "default"), " ".join(out), flush=True)
