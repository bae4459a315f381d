"""Upper Gram matrix Y^T Y of tall 256- / 128- / 512-column factors (split-K tiled kernel): us per call for a given library build."""
import os, sys, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from randlapack_amd import _lib
_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
for (m, n) in ((200000, 256), (25000, 256), (20000, 256), (200000, 128), (100000, 512), (200000, 64)):
    Y = d.cm_empty(m, n); ctx.fill_dense(Y, m, n, key=(1, 0)); G = d.cm_zeros(n, n)
    ctx.syrk("U", "T", n, m, 1.0, Y, m, 0.0, G, n); ctx.sync()
    best = 1e9
    for _ in range(5):
        ctx.timer_start()
        for _ in range(10): ctx.syrk("U", "T", n, m, 1.0, Y, m, 0.0, G, n)
        best = min(best, ctx.timer_stop_ms() / 10)
    print(os.path.basename(sys.argv[1]), m, n, f"{best * 1e3:.1f} us", float(G.sum()), flush=True)
