"""ABRIK on the C5 operator: best / median ms per call for a given library build (same-box A/B).  usage: abrik_ab.py <lib.so>"""
import os, sys, time, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
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
ts = []
for it in range(14):
    ctx.sync(); t0 = time.perf_counter(); r = d.drv_abrik_linop(ctx, op, k, eps, 2 * target // k, key=(2, 0), timing=False); ctx.sync(); ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print(os.path.basename(sys.argv[1]), f"best {ts[0] * 1e3:.3f} ms  median {ts[len(ts) // 2] * 1e3:.3f} ms  iters {r['iters']}", flush=True)
