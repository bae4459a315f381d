import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd.device import *
ctx = Context(0)
def bench(m,n,k,ta):
    if ta=='N':
        A = cm_empty(m,n); ctx.fill_dense(A, m, n, key=(7,0)); B = cm_empty(n,k); ctx.fill_dense(B, n, k, key=(8,0)); C = cm_empty(m,k)
        fn = lambda: ctx.gemm('N','N',m,k,n,1.0,A,m,B,n,0.0,C,m)
    else:
        A = cm_empty(m,n); ctx.fill_dense(A, m, n, key=(7,0)); B = cm_empty(m,k); ctx.fill_dense(B, m, k, key=(8,0)); C = cm_empty(n,k)
        fn = lambda: ctx.gemm('T','N',n,k,m,1.0,A,m,B,m,0.0,C,n)
    fn(); ctx.sync(); ctx.timer_start()
    for _ in range(3): fn()
    ms = ctx.timer_stop_ms()/3
    print(f'{ta} m={m} n={n} k={k}: {ms:.3f} ms {2.0*m*n*k/ms/1e9:.1f} TF', flush=True)
for (m,n) in [(32768,20000),(65536,20000),(32768*6,20000),(200000,20000),(32768,2000)]:
    bench(m,n,256,'N')
