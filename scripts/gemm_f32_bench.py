"""fp32 GEMM shapes of BQRRP's compact-WY apply at BASELINE configs[3]: stream-K twin vs generic kernel (RLHIP_STREAMK_F32=0)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from randlapack_amd import device as d
ctx = d.Context(0)
f32 = torch.float32
for (ta, M, N, K) in [("T", 2048, 63488, 63488), ("N", 63488, 63488, 2048), ("T", 2048, 32768, 32768), ("N", 32768, 32768, 2048), ("N", 65536, 2048, 2048), ("T", 2048, 2048, 65536)]:
    A = d.cm_empty(K if ta == "T" else M, M if ta == "T" else K, dtype=f32); ctx.fill_dense(A, A.shape[1], A.shape[0], key=(1, 0))
    B = d.cm_empty(K, N, dtype=f32); ctx.fill_dense(B, K, N, key=(2, 0))
    C = d.cm_zeros(M, N, dtype=f32)
    lda = A.shape[1]
    before = ctx.path_count(1)
    ctx.gemm(ta, "N", M, N, K, 1.0, A, lda, B, K, 1.0, C, M); ctx.sync()
    used = ctx.path_count(1) - before
    ctx.timer_start()
    reps = 3
    for _ in range(reps): ctx.gemm(ta, "N", M, N, K, 1.0, A, lda, B, K, 1.0, C, M)
    ms = ctx.timer_stop_ms() / reps
    print(f"{ta}N {M}x{N}x{K}: {ms:.2f} ms {2.0*M*N*K/ms/1e9:.1f} TFLOP/s streamk={used}", flush=True)
    del A, B, C
