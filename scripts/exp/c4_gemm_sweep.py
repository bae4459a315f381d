"""The two products of BQRRP's compact-WY apply at C4 (65536^2 fp32, b = 2048) at the shapes of block iteration i: W = V^T C (TN, contraction over
the m_i rows) and C -= V W (NN, K = b): ms and TFLOP/s per iteration -- where does the apply's average (124.6) fall below the best case (137)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
N, b = 65536, 2048
V = d.cm_empty(N, b, dtype=torch.float32); ctx.fill_dense(V, N, b, key=(1, 0))
Cm = d.cm_empty(N, N - b, dtype=torch.float32); ctx.fill_dense(Cm, N, N - b, key=(2, 0))
W = d.cm_zeros(b, N - b, dtype=torch.float32)
tot = {"TN": [0.0, 0.0], "NN": [0.0, 0.0]}
for i in range(0, 31):
    mi = N - b * i - b      # rows below the panel's triangle (V2, C2)
    ni = N - b * (i + 1)    # trailing columns
    if ni <= 0 or mi <= 0: break
    res = {}
    for name, fn, fl in (("TN", lambda: ctx.gemm("T", "N", b, ni, mi, 1.0, V, N, Cm, N, 0.0, W, b), 2.0 * b * ni * mi),
                         ("NN", lambda: ctx.gemm("N", "N", mi, ni, b, -1e-6, V, N, W, b, 1.0, Cm, N), 2.0 * b * ni * mi)):
        fn(); ctx.sync(); best = 1e9
        for _ in range(2):
            ctx.timer_start(); fn(); best = min(best, ctx.timer_stop_ms())
        res[name] = (best, fl / best / 1e9)
        tot[name][0] += best; tot[name][1] += fl
    print(f"iter {i:2d} m {mi:6d} n {ni:6d}: TN {res['TN'][0]:8.3f} ms {res['TN'][1]:6.1f} TF   NN {res['NN'][0]:8.3f} ms {res['NN'][1]:6.1f} TF", flush=True)
for k, (ms, fl) in tot.items(): print(f"{k}: total {ms:.1f} ms, {fl / ms / 1e9:.1f} TFLOP/s")
