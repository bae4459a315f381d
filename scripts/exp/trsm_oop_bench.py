import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n = 1048576, 1024
A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(3, 0)); U = d.cm_empty(n, n); ctx.fill_dense(U, n, n, key=(2, 0))
ctx.lib.rlhip_add_diag_f64(ctx.h, n, 40.0, U.data_ptr(), n)
ldw = m + 32
W = torch.empty((n, ldw), dtype=torch.float64, device="cuda"); Jp = torch.arange(n, 0, -1, dtype=torch.int64, device="cuda")
B = A.clone()
res = []
for _ in range(3):
    ctx.trsm_gather(m, n, 1.0, U, n, A, m, Jp, W, ldw); ctx.sync(); ctx.timer_start()
    for _ in range(5): ctx.trsm_gather(m, n, 1.0, U, n, A, m, Jp, W, ldw)
    t1 = ctx.timer_stop_ms() / 5
    ctx.trsm(m, n, 1.0, U, n, B, m); ctx.sync(); ctx.timer_start()
    for _ in range(5): ctx.trsm(m, n, 1.0, U, n, B, m)
    t2 = ctx.timer_stop_ms() / 5
    res.append((round(t1, 3), round(t2, 3)))
print("HPR", os.environ.get("RLHIP_DBG_TF_HPR", "16"), "oop / in-place ms:", res, flush=True)
res = []
for _ in range(2):
    ctx.trsm_gather(m, n, 1.0, U, n, A, m, None, W, ldw); ctx.sync(); ctx.timer_start()
    for _ in range(5): ctx.trsm_gather(m, n, 1.0, U, n, A, m, None, W, ldw)
    res.append(round(ctx.timer_stop_ms() / 5, 3))
print("oop without a pivot vector ms:", res, flush=True)
