import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from randlapack_amd.device import *
ctx = Context(0)
m = 1048576
B = cm_empty(m, 1024); ctx.fill_dense(B, m, 1024, key=(1,0))
U = cm_from_numpy(np.triu(np.random.default_rng(0).standard_normal((1024,1024))) + 40*np.eye(1024))
C2 = cm_empty(m, 64)
for (n, lab) in [(32,'trsm n=32'), (64,'trsm n=64'), (256, 'trsm n=256')]:
    ctx.trsm(m, n, 1.0, U, 1024, B, m); ctx.sync()
    ctx.timer_start()
    for _ in range(5): ctx.trsm(m, n, 1.0, U, 1024, B, m)
    print(lab, ctx.timer_stop_ms()/5*1e3, 'us')
ctx.lacpy('A', m, 32, B, m, C2, m); ctx.sync()
ctx.timer_start()
for _ in range(5): ctx.lacpy('A', m, 32, B, m, C2, m)
t = ctx.timer_stop_ms()/5*1e3
print('lacpy m x 32', t, 'us ->', 2*m*32*8/t/1e6, 'TB/s')
x = torch.empty(m*32, dtype=torch.float64, device='cuda'); y = torch.empty_like(x)
torch.cuda.synchronize(); t0=time.time()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); t=(time.time()-t0)/10*1e6
print('torch copy same bytes', t, 'us ->', 2*m*32*8/t/1e6, 'TB/s')
