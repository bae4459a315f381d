import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from randlapack_amd.device import *
ctx = Context(0)
m, n = 1048576, 1024
B = cm_empty(m, n); ctx.fill_dense(B, m, n, key=(1,0))
U = cm_empty(n, n); ctx.fill_dense(U, n, n, key=(2,0)); ctx.lib.rlhip_add_diag_f64(ctx.h, n, 40.0, U.data_ptr(), n)
G = cm_zeros(n, n)
for name, fn in [('trsm', lambda: ctx.trsm(m, n, 1.0, U, n, B, m)), ('syrk', lambda: ctx.syrk('U','T',n,m,1.0,B,m,0.0,G,n)), ('potrf', lambda: (ctx.add_diag if False else None, ctx.lib.rlhip_add_diag_f64(ctx.h, n, 1e9, G.data_ptr(), n), ctx.potrf(n, G, n)))]:
    fn(); ctx.sync(); ctx.timer_start(); fn(); ms = ctx.timer_stop_ms()
    print(f'{name}: {ms:.2f} ms', flush=True)
