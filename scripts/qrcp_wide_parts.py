"""Per-part timing of one BQRRP iteration's qrcp_wide = luqr (rl_bqrrp.hh:337-357 of the reference) on a d x cols fp32 sketch:
transpose, row-pivoted LU of the transposed sketch (pivots only), pivot conversion, column permutation of the sketch, wide geqrf.
  python scripts/qrcp_wide_parts.py [d=2048] [cols ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from randlapack_amd import device as d

dd = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cols_list = [int(x) for x in sys.argv[2:]] or [65536, 49152, 32768, 16384, 4096]
ctx = d.Context(0)
L = ctx.lib
for cols in cols_list:
    S = d.cm_empty(dd, cols, dtype=torch.float32); ST = d.cm_empty(cols, dd, dtype=torch.float32)
    ip = torch.zeros(dd, dtype=torch.int64, device="cuda"); J = torch.zeros(cols, dtype=torch.int64, device="cuda")
    tau = torch.zeros(dd, dtype=torch.float32, device="cuda")
    res = {}
    for rep in range(2):
        ctx.fill_dense(S, dd, cols, key=(9, 0)); ctx.sync()
        def t(name, f):
            ctx.timer_start(); rc = f(); ms = ctx.timer_stop_ms(); assert rc == 0, (name, rc); res[name] = ms
        t("transpose", lambda: L.rlhip_transpose_f32(ctx.h, dd, cols, S.data_ptr(), dd, ST.data_ptr(), cols, 0))
        t("getrf_piv", lambda: L.rlhip_getrf_piv_f32(ctx.h, cols, dd, ST.data_ptr(), cols, ip.data_ptr()))
        t("luqrcp_piv", lambda: L.rlhip_luqrcp_piv(ctx.h, dd, cols, ip.data_ptr(), J.data_ptr()))
        t("col_swap", lambda: L.rlhip_col_swap_f32(ctx.h, dd, cols, cols, S.data_ptr(), dd, J.data_ptr()))
        t("geqrf_wide", lambda: L.rlhip_geqrf_f32(ctx.h, dd, cols, S.data_ptr(), dd, tau.data_ptr()))
    print(cols, {k: round(v, 2) for k, v in res.items()}, "sum", round(sum(res.values()), 2), flush=True)
