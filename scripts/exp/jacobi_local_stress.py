"""Same-XCD hand-over stress: ONE context, many consecutive device SVDs of different factors (sizes, spectra, sweep counts), every result
compared bitwise with the uncached hand-over (option 2) run right after it on the same input; both Gram and classic routes."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from randlapack_amd import device as d
ctx = d.Context(0)
rng = np.random.default_rng(11)
bad = 0; runs = 0; sweeps_seen = set()
for trial in range(60):
    n = int(rng.choice([256, 256, 224, 192, 130, 96, 64]))
    m = int(rng.choice([3000, 5000, 20000]))
    kind = trial % 4
    if kind == 0: A = rng.standard_normal((m, n))
    else:
        s = [np.logspace(0, -1, n), 1.0 - 1e-7 * rng.random(n), np.linspace(1, 2, n)][kind - 1]
        A = (np.linalg.qr(rng.standard_normal((m, n)))[0] * s) @ np.linalg.qr(rng.standard_normal((n, n)))[0].T
    ctx.set_option("gesdd_gram", trial % 2)
    out = {}
    for mode in (1, 2):
        ctx.set_option("jacobi_persist", mode)
        Ad = d.cm_from_numpy(A); S = torch.zeros(n, dtype=torch.float64, device="cuda"); U = d.cm_empty(m, n); VT = d.cm_empty(n, n); sw = C.c_int(0)
        rc = ctx.lib.rlhip_gesdd_f64(ctx.h, m, n, Ad.data_ptr(), m, S.data_ptr(), U.data_ptr(), m, VT.data_ptr(), n, C.byref(sw)); ctx.sync()
        assert rc == 0
        out[mode] = (U.clone(), S.clone(), VT.clone(), sw.value)
    same = all(torch.equal(a, b) for a, b in zip(out[1][:3], out[2][:3])) and out[1][3] == out[2][3]
    Sh = out[1][1].cpu().numpy(); ref = np.linalg.svd(A, compute_uv=False)
    acc = np.abs(np.sort(Sh)[::-1] - ref).max() / ref[0]
    runs += 1; sweeps_seen.add(out[1][3])
    if not same or acc > 1e-12: bad += 1; print("MISMATCH", trial, m, n, kind, same, acc, flush=True)
print(f"{runs} factors, {bad} mismatches, sweep counts seen {sorted(sweeps_seen)}", flush=True)
