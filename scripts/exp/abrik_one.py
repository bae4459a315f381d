"""One warm-up and one traced ABRIK call on the C5 operator (200000^2 CSR, block 32) for a kernel timeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, scipy.sparse as sp
from randlapack_amd import device as d
ctx = d.Context(0)
m = n = 200000; k, target = 32, 128
rng = np.random.default_rng(77)
rows = np.repeat(np.arange(m), 10); colsi = (rows + np.tile(np.arange(-4, 6), m)) % n
vals = rng.standard_normal(m * 10)
d1 = np.exp(-np.arange(m) / 4.0) + 1e-13; d2 = np.exp(-np.arange(n) / 4.0) + 1e-13
G = sp.csr_matrix((vals * d1[rows] * d2[colsi], (rows, colsi)), shape=(m, n)); G.sum_duplicates()
op = d.CsrOperator.from_scipy(G)
eps = float(np.finfo(float).eps ** 0.85)
for it in range(3):
    ctx.sync(); X = torch.zeros(8, device="cuda"); X.fill_(float(it)); torch.cuda.synchronize()   # marker: a torch fill kernel before every call
    t0 = time.perf_counter(); r = d.drv_abrik_linop(ctx, op, k, eps, 2 * target // k, key=(2, 0), timing=False); ctx.sync()
    print(f"call {it}: {(time.perf_counter() - t0) * 1e3:.2f} ms iters {r['iters']} triplets {r['triplets']}", flush=True)
