import sys, torch, time
from randlapack_amd import device as d
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = d.Context(0)
A = d.cm_empty(m, m)
for it in range(2):
    ctx.fill_dense(A, m, m, key=(4, 0)); ctx.sync()
    t0 = time.perf_counter(); r = d.drv_hqrrp(ctx, A, m, m, nb_alg=nb, pp=10, qr_type=0); ctx.sync()
    print("hqrrp ms", (time.perf_counter() - t0) * 1e3)
