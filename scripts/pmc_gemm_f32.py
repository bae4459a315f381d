"""The two products of BQRRP's compact-WY apply at 32768^2 fp32, b = 2048 (generic MFMA GEMM): W = V^T C (TN) and C -= V W (NN)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
m, n, b = 32768, 30720, 2048
f32 = torch.float32
V = d.cm_empty(m, b, dtype=f32); ctx.fill_dense(V, m, b, key=(1, 0))
Cm = d.cm_empty(m, n, dtype=f32); ctx.fill_dense(Cm, m, n, key=(2, 0))
W = d.cm_empty(b, n, dtype=f32)
for _ in range(2):
    ctx.gemm("T", "N", b, n, m, 1.0, V, m, Cm, m, 0.0, W, b)
    ctx.gemm("N", "N", m, n, b, -1e-6, V, m, W, b, 1.0, Cm, m)
ctx.sync()
