"""C3's Gram matrix (1048576 x 1024 fp64, ld m + 32, triangular stream-K map) with P workgroups instead of one per CU: with P = 18 * s the 18
tiles are cut into s equal K-shares each, so the 18 workgroups of one K-share walk the SAME rows of A at the same time (Infinity-Cache / L2
sharing instead of 18 unrelated streams).  One process per setting (the tuning variable is read once).  usage: tri_gram_ab.py [P ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from randlapack_amd import device as d
    ctx = d.Context(0)
    m, n = 1048576, 1024
    ld = m + 32
    A = torch.empty((n, ld), dtype=torch.float64, device="cuda"); ctx.fill_dense(A, ld, n, key=(3, 0))
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    ctx.syrk("U", "T", n, m, 1.0, A, ld, 0.0, G, n); ctx.sync()
    best = 1e9
    for _ in range(4):
        ctx.timer_start()
        for _ in range(5): ctx.syrk("U", "T", n, m, 1.0, A, ld, 0.0, G, n)
        best = min(best, ctx.timer_stop_ms() / 5)
    print(f"P={os.environ.get('RLHIP_SK_TUNE', 'default'):>14s}  {best:7.3f} ms  {1.0995e12 / best / 1e9:6.1f} TFLOP/s  checksum {float(G.sum()):.12e}", flush=True)
    sys.exit(0)
for P in (sys.argv[1:] or ["0", "252", "234", "216", "198", "180", "144"]):
    env = dict(os.environ, RLHIP_SK_CLOCK="1")
    if P != "0": env["RLHIP_SK_TUNE"] = f"0,0,{P},-1"
    r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True, timeout=300)
    clk = [ln for ln in r.stderr.splitlines() if "[sk clock]" in ln]
    print(r.stdout.strip(), "|", clk[-1] if clk else r.stderr[-300:], flush=True)
