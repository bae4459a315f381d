import sys, torch
sys.path.insert(0, "/root/repo")
from randlapack_amd import device as d
rows = int(sys.argv[1]); n = 2048
ctx = d.Context(0)
A = torch.randn((n, rows), dtype=torch.float32, device="cuda")
ip = torch.zeros(n, dtype=torch.int64, device="cuda")
for _ in range(2):
    B = A.clone(); ctx.sync()
    ctx.timer_start(); rc = ctx.lib.rlhip_getrf_piv_f32(ctx.h, rows, n, B.data_ptr(), rows, ip.data_ptr()); ms = ctx.timer_stop_ms()
    print(rows, rc, ms)
