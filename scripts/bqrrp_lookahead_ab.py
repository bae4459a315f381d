"""BQRRP with and without the look-ahead (RLHIP_BQRRP_LOOKAHEAD is read once per process: run this script once per setting and compare the
checksums): wall time per factorization and a checksum of (A_out, tau, J).  usage: bqrrp_lookahead_ab.py m b {f32|f64} [reps]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd import device as d
m = int(sys.argv[1]); b = int(sys.argv[2]); dt = torch.float32 if sys.argv[3] == "f32" else torch.float64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ctx = d.Context(0)
A = d.cm_empty(m, m, dtype=dt)
best = 1e9
for it in range(reps):
    ctx.fill_dense(A, m, m, key=(4, 0)); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = d.drv_bqrrp(ctx, A, m, m, b, 1.0, timing=False, qr_tall=1, apply_trans_q=1)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
h = hashlib.sha256()
h.update(A.cpu().numpy().tobytes()); h.update(r["tau"].cpu().numpy().tobytes()); h.update(r["J"].cpu().numpy().tobytes())
print(f"lookahead {os.environ.get('RLHIP_BQRRP_LOOKAHEAD', 'default')}: m = n = {m} b = {b} {sys.argv[3]}: {best * 1e3:.1f} ms, rank {r['rank']}, checksum {h.hexdigest()[:16]}", flush=True)
