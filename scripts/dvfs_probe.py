"""How the MI355X's clock follows the load between the two tall products of a 1/8-shard RSVD step (RLHIP_SK_CLOCK=1 prints the
effective shader clock of every stream-K launch).  Pattern: {NN product, TN product, tail} x 6 for several kinds of tail."""
import os, sys, time
os.environ["RLHIP_SK_CLOCK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n, k = 25000, 20000, 256
A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7, 0))
Om = d.cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(0, 0))
Q = d.cm_empty(m, k); ctx.fill_dense(Q, m, k, key=(1, 0))
Y = d.cm_empty(m, k); BT = d.cm_empty(n, k)
nn = lambda: ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)
tn = lambda: ctx.gemm("T", "N", n, k, m, 1.0, A, m, Q, m, 0.0, BT, n)
burn = lambda blocks, mode, usec, side=0: ctx.lib.rlhip_dvfs_burn(ctx.h, blocks, mode, usec, side)
tails = {
    "none": lambda: None,
    "host sleep 4 ms": lambda: time.sleep(0.004),
    "8 WGs sleeping 4 ms": lambda: burn(8, 0, 4000),
    "8 WGs fma 4 ms": lambda: burn(8, 1, 4000),
    "256 WGs fma 4 ms": lambda: burn(256, 1, 4000),
    "1024 WGs fma 4 ms": lambda: burn(1024, 1, 4000),
    "1024 WGs mfma 4 ms": lambda: burn(1024, 2, 4000),
    "8 WGs fma 4 ms + 1024 WGs mfma beside": lambda: (burn(1024, 2, 3800, 1), burn(8, 1, 4000)),
    "8 WGs fma 4 ms + 248 WGs mfma beside": lambda: (burn(248, 2, 3800, 1), burn(8, 1, 4000)),
    "8 WGs fma 4 ms + 248 WGs fma beside": lambda: (burn(248, 1, 3800, 1), burn(8, 1, 4000)),
    "8 WGs fma 2 ms": lambda: burn(8, 1, 2000),
    "8 WGs fma 1 ms": lambda: burn(8, 1, 1000),
}
for name, tail in tails.items():
    for _ in range(6): nn()          # warm the clock
    ctx.sync(); torch.cuda.synchronize()
    print(f"==== tail: {name}", file=sys.stderr, flush=True)
    for _ in range(6):
        nn(); tn(); tail(); ctx.sync(); torch.cuda.synchronize()
