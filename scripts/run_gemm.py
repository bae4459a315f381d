import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd.device import *
ctx = Context(0)
m,n,k = 200000, 20000, 256
A = cm_empty(m,n); ctx.fill_dense(A, m, n, key=(7,0))
Om = cm_empty(n,k); ctx.fill_dense(Om, n, k, key=(8,0))
Y = cm_empty(m,k); BT = cm_empty(n,k)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, fn in [('A*Om', lambda: ctx.gemm('N','N',m,k,n,1.0,A,m,Om,n,0.0,Y,m)), ('At*Y', lambda: ctx.gemm('T','N',n,k,m,1.0,A,m,Y,m,0.0,BT,n))]:
    fn(); ctx.sync()
    ctx.timer_start()
    for _ in range(reps): fn()
    ms = ctx.timer_stop_ms()/reps
    print(f'{name}: {ms:.2f} ms  {2.0*m*n*k/ms/1e9:.1f} TFLOP/s', flush=True)
