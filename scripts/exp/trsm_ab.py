"""A/B of two builds of librlhip.so on the fused solves of C3 (in place, out of place with / without a pivot vector).  usage: trsm_ab.py <lib.so>"""
import os, sys, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n = 1048576, 1024
A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(3, 0)); U = d.cm_empty(n, n); ctx.fill_dense(U, n, n, key=(2, 0))
ctx.lib.rlhip_add_diag_f64(ctx.h, n, 40.0, U.data_ptr(), n)
ldw = m + 32
W = torch.empty((n, ldw), dtype=torch.float64, device="cuda"); Jp = torch.arange(n, 0, -1, dtype=torch.int64, device="cuda")
B = A.clone()
def t(fn):
    fn(); ctx.sync(); best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(5): fn()
        best = min(best, ctx.timer_stop_ms() / 5)
    return round(best, 3)
print(os.path.basename(sys.argv[1]), "oop+perm", t(lambda: ctx.trsm_gather(m, n, 1.0, U, n, A, m, Jp, W, ldw)),
      "oop", t(lambda: ctx.trsm_gather(m, n, 1.0, U, n, A, m, None, W, ldw)), "in place", t(lambda: ctx.trsm(m, n, 1.0, U, n, B, m)), flush=True)
