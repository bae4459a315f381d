"""BQRRP with and without the look-ahead (BQRRP::lookahead_min_elems through the context option): wall time per factorization, separate
checksums of J and of (A_out, tau), and -- the question VERDICT r4 asked -- whether the PIVOTS moved between the two orders.
usage: bqrrp_lookahead_ab.py m b {f32|f64} [reps]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd import device as d
m = int(sys.argv[1]); b = int(sys.argv[2]); dt = torch.float32 if sys.argv[3] == "f32" else torch.float64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ctx = d.Context(0)
A = d.cm_empty(m, m, dtype=dt)
out = {}
for name, thresh in (("serial", 1 << 62), ("lookahead", 0)):
    ctx.set_option("bqrrp_lookahead_min_elems", thresh)
    best = 1e9
    for it in range(reps):
        ctx.fill_dense(A, m, m, key=(4, 0)); ctx.sync(); torch.cuda.synchronize()
        n0 = ctx.path_count(12)
        t0 = time.perf_counter()
        r = d.drv_bqrrp(ctx, A, m, m, b, 1.0, timing=False, qr_tall=1, apply_trans_q=1)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    J = r["J"].cpu().numpy()
    hj = hashlib.sha256(J.tobytes()).hexdigest()[:16]
    hf = hashlib.sha256(); hf.update(A.cpu().numpy().tobytes()); hf.update(r["tau"].cpu().numpy().tobytes())
    out[name] = (J, A.diagonal().clone() if False else torch.diagonal(A).abs().cpu().numpy().copy())
    print(f"{name}: m = n = {m} b = {b} {sys.argv[3]}: {best * 1e3:.1f} ms, rank {r['rank']}, side-queue iterations {ctx.path_count(12) - n0}, "
          f"J sha {hj}, (A, tau) sha {hf.hexdigest()[:16]}", flush=True)
Js, Jl = out["serial"][0], out["lookahead"][0]
same = Js == Jl
first = int((~same).argmax()) if not same.all() else -1
print(f"pivots identical: {bool(same.all())}; positions that differ: {int((~same).sum())} of {m}" + (f", first at {first} (block {first // b})" if first >= 0 else ""))
ov = [len(set(Js[i:i + b].tolist()) & set(Jl[i:i + b].tolist())) / b for i in range(0, m, b)]
print(f"block-wise overlap of the pivot sets: min {min(ov):.4f} mean {sum(ov) / len(ov):.4f}")
ds, dl = out["serial"][1], out["lookahead"][1]
print(f"|diag R| serial vs lookahead: max rel diff {float(abs(ds - dl).max() / ds.max()):.3e}")
