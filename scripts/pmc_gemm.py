import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from randlapack_amd.device import *
ctx = Context(0)
m, n, k = 200000, 20000, 256
A = cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7,0))
Om = cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(0,0))
Y = cm_empty(m, k)
for _ in range(3):
    ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)
ctx.sync()
