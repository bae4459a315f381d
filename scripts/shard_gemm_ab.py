"""Per-launch HIP-event times of the two tall products at the 1/8-shard shape against the full shape (is the shard's lower rate structural
or a clock ramp?) and A/B of RLHIP_SK_TUNE=gs_nn,gs_tn,workgroups.  usage: python scripts/shard_gemm_ab.py [m ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
n, k = 20000, 256
ms = [int(x) for x in sys.argv[1:]] or [25000, 200000]
for m in ms:
    A = d.cm_empty(m, n); ctx.fill_dense(A, m, n, key=(7, 0))
    Om = d.cm_empty(n, k); ctx.fill_dense(Om, n, k, key=(0, 0))
    Q = d.cm_empty(m, k); ctx.fill_dense(Q, m, k, key=(1, 0))
    Y = d.cm_empty(m, k); BT = d.cm_empty(n, k)
    for name, fn in (("NN", lambda: ctx.gemm("N", "N", m, k, n, 1.0, A, m, Om, n, 0.0, Y, m)), ("TN", lambda: ctx.gemm("T", "N", n, k, m, 1.0, A, m, Q, m, 0.0, BT, n))):
        fn(); ctx.sync()
        reps = max(4, min(40, int(2.0e6 / m)))
        times = []
        for _ in range(reps):
            ctx.timer_start(); fn(); times.append(ctx.timer_stop_ms())
        torch.cuda.synchronize()
        import time; time.sleep(0.05)
        ctx.timer_start()
        for _ in range(reps): fn()
        bb = ctx.timer_stop_ms() / reps
        tf = 2.0 * m * n * k / 1e9
        print(f"tune {os.environ.get('RLHIP_SK_TUNE', 'default')} m {m} {name}: single launches min {min(times):.3f} med {sorted(times)[len(times)//2]:.3f} first {times[0]:.3f} ms "
              f"({tf / min(times):.1f} TF) ; {reps} back to back {bb:.3f} ms ({tf / bb:.1f} TF)", flush=True)
    del A, Q, Y
