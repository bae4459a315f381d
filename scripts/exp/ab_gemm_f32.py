"""A/B of two builds of librlhip.so on the fp32 products of C4's compact-WY apply: usage ab_gemm_f32.py <lib.so>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathlib
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
rows, b, rest = 65536, 2048, 16384
V = d.cm_empty(rows, b, dtype=torch.float32); ctx.fill_dense(V, rows, b, key=(1, 0))
Cm = d.cm_empty(rows, rest, dtype=torch.float32); ctx.fill_dense(Cm, rows, rest, key=(2, 0))
W = d.cm_zeros(b, rest, dtype=torch.float32)
out = {}
for name, fn in (("TN", lambda: ctx.gemm("T", "N", b, rest, 16384, 1.0, V, rows, Cm, rows, 0.0, W, b)),
                 ("NN", lambda: ctx.gemm("N", "N", rows, rest, b, -1e-6, V, rows, W, b, 1.0, Cm, rows))):
    fn(); ctx.sync(); best = 1e9
    for _ in range(4):
        ctx.timer_start()
        for _ in range(5): fn()
        best = min(best, ctx.timer_stop_ms() / 5)
    out[name] = round(best, 3)
print(os.path.basename(sys.argv[1]), out, flush=True)
del V, Cm, W
m, n, k = 200000, 20000, 256
A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7, 0))
Om = d.cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(8, 0))
Y = d.cm_zeros(m, k); Bt = d.cm_zeros(n, k)
out = {}
for name, fn in (("NN64", lambda: ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)),
                 ("TN64", lambda: ctx.gemm("T", "N", n, k, m, 1.0, A, m, Y, m, 0.0, Bt, n))):
    fn(); ctx.sync(); best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(3): fn()
        best = min(best, ctx.timer_stop_ms() / 3)
    out[name] = round(best, 3)
print(os.path.basename(sys.argv[1]), out, flush=True)
