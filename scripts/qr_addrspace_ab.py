#!/usr/bin/env python3
"""A/B of the cooperative Householder kernels (qr_pipe_kernel: unpivoted, flag-pipelined; qrcp_kernel: pivoted, rendezvous) before and after their
LDS-or-global column pointers became template parameters (no flat_ instructions).  Usage: qr_addrspace_ab.py [path of the library to load].
Every shape is factored twice; the second time is printed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import randlapack_amd._lib as L
if len(sys.argv) > 1:
    import pathlib
    L.LIB_PATH = pathlib.Path(sys.argv[1])
import torch
import randlapack_amd.device as d
ctx = d.Context(0)
os.environ["RLHIP_QR_BLK"] = "0"         # keep geqrf on the pipelined kernel for the shapes the blocked route would take
os.environ["RLHIP_QRCP_TAG"] = "0"       # and geqp3 on the rendezvous kernel
def t_geqrf(m, n, dt=torch.float64):
    fn = ctx.lib.rlhip_geqrf_f64 if dt == torch.float64 else ctx.lib.rlhip_geqrf_f32
    best = None
    for rep in range(2):
        A = torch.randn((n, m), dtype=dt, device="cuda:0"); tau = torch.zeros(n, dtype=dt, device="cuda:0")
        ctx.sync(); t0 = time.perf_counter(); rc = fn(ctx.h, m, n, A.data_ptr(), m, tau.data_ptr()); ctx.sync(); best = time.perf_counter() - t0
        assert rc == 0
    return best
def t_geqp3(m, n, dt=torch.float64):
    fn = ctx.lib.rlhip_geqp3_f64 if dt == torch.float64 else ctx.lib.rlhip_geqp3_f32
    best = None
    for rep in range(2):
        A = torch.randn((n, m), dtype=dt, device="cuda:0"); tau = torch.zeros(min(m, n), dtype=dt, device="cuda:0"); jp = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        ctx.sync(); t0 = time.perf_counter(); rc = fn(ctx.h, m, n, A.data_ptr(), m, jp.data_ptr(), tau.data_ptr()); ctx.sync(); best = time.perf_counter() - t0
        assert rc == 0
    return best
print("library:", L.LIB_PATH)
for (m, n) in ((2000, 1000), (1280, 512), (2560, 1280), (8192, 1024), (200000, 32), (100000, 128)):
    t = t_geqrf(m, n); print(f"geqrf  {m:7d} x {n:5d} fp64: {t * 1e3:9.3f} ms = {t * 1e6 / min(m, n):7.2f} us per column")
for (m, n) in ((1280, 1024), (2560, 2048), (512, 8192), (4096, 4096)):
    t = t_geqp3(m, n); print(f"geqp3  {m:7d} x {n:5d} fp64: {t * 1e3:9.3f} ms = {t * 1e6 / min(m, n):7.2f} us per column")
t = t_geqrf(2000, 1000, torch.float32); print(f"geqrf     2000 x  1000 fp32: {t * 1e3:9.3f} ms")
t = t_geqp3(2560, 2048, torch.float32); print(f"geqp3     2560 x  2048 fp32: {t * 1e3:9.3f} ms")
