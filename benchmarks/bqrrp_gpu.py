"""BQRRP_GPU benchmark main on the device library (reference: benchmark/bench_BQRRP/BQRRP_GPU_benchmark.cu, the only GPU benchmark the
reference ships), built on the BQRRP_GPU class (include/RandLAPACK_amd/rl_bqrrp_gpu.hh) through `rlhip_drv_bqrrp_gpu_*`.

  python -m benchmarks.bqrrp_gpu block_size [matrix_size = 16384] [profile_runtime = 0] [run_qrf = 1] [dir = .] [b1 b2 ...]
        (:162-232, :313-327) -> _BQRRP_GPU_speed_comparisons_block_size_num_info_lines_6.txt and, with profile_runtime,
           _BQRRP_GPU_runtime_breakdown_qrf_num_info_lines_6.txt / _BQRRP_GPU_runtime_breakdown_cholqr_num_info_lines_6.txt
        default block sizes: the reference's 32 ... 2048 (:313)
  python -m benchmarks.bqrrp_gpu mat_size [profile_runtime = 0] [run_qrf = 1] [dir = .] [m1 m2 ...]
        (:234-270, :329-343) -> BQRRP_GPU_speed_comparisons_mat_size_num_info_lines_6.txt; block size m / 32, sizes 512 ... 32768 (:339)

Same file names, header blocks (6 info lines), column order and units (microseconds) as the reference, so its plotting scripts read them:
speed files hold `BQRRP+QRF  BQRRP+CholQR  QRF` per line (the order the reference WRITES, :172; its header text names them in another
order), breakdown files the 15 entries of BQRRP_GPU::times (rl_bqrrp_gpu.hh:829-834).

What differs from the reference main: (1) the Gaussian input is generated in HBM and RE-generated before every factorization -- the
reference regenerates the host copy only (:61-73) and never uploads it again, so from the second call on it factors its own previous
output; (2) the sketch S A is formed on the device (Gaussian S from the counter stream + one MFMA GEMM) instead of on the host; it is
produced BEFORE the timed region in both, so the timed quantity is the same: BQRRP_GPU::call alone."""
from __future__ import annotations

import sys
import time

import torch

from randlapack_amd import device as d

from . import _common as c

BLOCK_SIZES = [32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448, 480, 512, 640, 768, 896, 1024, 1152, 1280, 1408, 1536, 1664, 1792,
               1920, 2048]
MAT_SIZES = [512, 1024, 2048, 4096, 8192, 16384, 32768]
TIMES_HEADER = ("preallocation_t_dur, qrcp_main_t_dur, copy_A_sk_t_dur, qrcp_piv_t_dur, copy_A_t_dur, piv_A_t_dur, copy_J_t_dur, updating_J_t_dur, "
                "preconditioning_t_dur, qr_tall_t_dur, q_reconstruction_t_dur, apply_transq_t_dur, sample_update_t_dur, t_rest, total_t_dur")
CHOLQR, GEQRF = 0, 1


def _sketch(ctx, A, m, n, dd):
    S = d.cm_empty(dd, m, dtype=A.dtype, device=A.device)
    ctx.fill_dense(S, dd, m, key=(1, 0))
    A_sk = d.cm_empty(dd, n, dtype=A.dtype, device=A.device)
    ctx.gemm("N", "N", dd, n, m, 1.0, S, dd, A, m, 0.0, A_sk, dd)
    return A_sk


def bench_bqrrp(ctx, profile_runtime, run_qrf, m, n, block_size, file_qrf, file_cholqr, file_speed):
    """bench_BQRRP (:76-160): BQRRP_GPU with geqrf panels, with Cholesky-QR panels, optionally the plain device geqrf"""
    dd = block_size                                            # d_factor = 1 (:89)
    out = {}
    for name, which in (("qrf", GEQRF), ("cholqr", CHOLQR)):
        A = c.regen(ctx, "gaussian", m, n)
        A_sk = _sketch(ctx, A, m, n, dd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = d.drv_bqrrp_gpu(ctx, A, m, n, A_sk, dd, block_size, qr_tall=which, timing=profile_runtime)
        torch.cuda.synchronize()
        out[name] = int(round((time.perf_counter() - t0) * 1e6))
        assert r["rc"] == 0
        if profile_runtime:
            with open(file_qrf if name == "qrf" else file_cholqr, "a") as f:
                f.write("".join(f"{t}, " for t in r["times_us"]) + "\n")
        del A, A_sk
    diff_qrf = 0
    if run_qrf:
        A = c.regen(ctx, "gaussian", m, n)
        diff_qrf = c.timed_us(lambda: c.geqrf(ctx, A, m, n))
        print(f" QRF TIME (MS) = {diff_qrf}")
        del A
    print(f"  BLOCK SIZE = {block_size} BQRRP+QRF TIME (MS) = {out['qrf']} BQRRP+CholQR TIME (MS) = {out['cholqr']}")
    with open(file_speed, "a") as f:
        f.write(f"{out['qrf']}  {out['cholqr']}  {diff_qrf}\n")
    return out["qrf"], out["cholqr"], diff_qrf


def run_block_size_sweep(directory, m, n, b_sz, profile_runtime, run_qrf):
    ctx = d.Context(0)
    b_str = ",".join(map(str, b_sz))
    mt = c.MAT_TYPE_IDS["gaussian"]
    f1 = f2 = None
    if profile_runtime:
        f1 = c.out_path(directory, "_BQRRP_GPU_runtime_breakdown_qrf_num_info_lines_6.txt")
        f2 = c.out_path(directory, "_BQRRP_GPU_runtime_breakdown_cholqr_num_info_lines_6.txt")
        for path, sub in ((f1, "geqrf"), (f2, "cholqr")):      # (the reference prints the two subroutine names swapped, :195,205)
            with open(path, "a") as f:
                f.write("Description: Results from the BQRRP GPU runtime breakdown benchmark, recording the time it takes to perform every subroutine in BQRRP."
                        f"\nFile format: 15 data columns, each corresponding to a given BQRRP subroutine: {TIMES_HEADER}"
                        "               rows correspond to BQRRP runs with block sizes varying in a way unique for a particular run."
                        f"\nInput type:{mt}"
                        f"\nInput size:{m} by {n}"
                        f"\nAdditional parameters: Tall QR subroutine {sub} BQRRP block sizes: {b_str}"
                        "\n")
    f3 = c.out_path(directory, "_BQRRP_GPU_speed_comparisons_block_size_num_info_lines_6.txt")
    with open(f3, "a") as f:
        f.write("Description: Results from the BQRRP GPU speed comparison benchmark, recording the time it takes to perform BQRRP and alternative QR and QRCP factorizations."
                "\nFile format: 3 columns, containing time for each algorithm: BQRRP+CholQR, BQRRP+QRF, QRF;"
                "               rows correspond to BQRRP runs with block sizes varying in powers of 2 or multiples of 10"
                f"\nInput type:{mt}"
                f"\nInput size:{m} by {n}"
                f"\nAdditional parameters: BQRRP block sizes: {b_str}"
                "\n")
    t_all = time.perf_counter()
    rows = [bench_bqrrp(ctx, profile_runtime, run_qrf, m, n, b, f1, f2, f3) for b in b_sz]
    with open(f3, "a") as f:
        f.write(f"Total benchmark execution time:{int((time.perf_counter() - t_all) * 1e6)}\n")
    return f3, f1, f2, rows


def run_mat_size_sweep(directory, m_sz, profile_runtime, run_qrf):
    ctx = d.Context(0)
    path = c.out_path(directory, "BQRRP_GPU_speed_comparisons_mat_size_num_info_lines_6.txt")
    with open(path, "a") as f:
        f.write("Description: Results from the BQRRP GPU speed comparison benchmark, recording the time it takes to perform BQRRP and alternative QR and QRCP factorizations."
                "\nFile format: 3 columns, containing time for each algorithm: BQRRP+CholQR, BQRRP+QRF, QRF;"
                "               rows correspond to BQRRP runs with varying mat sizes, with numruns repititions of each mat size."
                f"\nInput type:{c.MAT_TYPE_IDS['gaussian']}"
                f"\nInput size: dim start: {','.join(map(str, m_sz))}"
                "\nAdditional parameters: BQRRP block size: 0"
                "\n")
    # the reference hands nullptr file names for the breakdowns here (:269): profile_runtime only arms the timers
    rows = [bench_bqrrp(ctx, False, run_qrf, m, m, max(1, m // 32), None, None, path) for m in m_sz]
    return path, rows


def main(argv):
    if not argv or argv[0] not in ("block_size", "mat_size"):
        print(__doc__)
        return 1
    mode, rest = argv[0], argv[1:]
    if mode == "block_size":
        m = int(rest[0]) if len(rest) > 0 else 16384
        prof = bool(int(rest[1])) if len(rest) > 1 else False
        qrf = bool(int(rest[2])) if len(rest) > 2 else True
        directory = rest[3] if len(rest) > 3 else "."
        b_sz = [int(x) for x in rest[4:]] or [b for b in BLOCK_SIZES if b <= m]
        print(f"Running block size sweep benchmark\nMatrix size: {m} x {m}\nProfile runtime: {'yes' if prof else 'no'}\nRun QRF: {'yes' if qrf else 'no'}\n")
        run_block_size_sweep(directory, m, m, b_sz, prof, qrf)
    else:
        prof = bool(int(rest[0])) if len(rest) > 0 else False
        qrf = bool(int(rest[1])) if len(rest) > 1 else True
        directory = rest[2] if len(rest) > 2 else "."
        m_sz = [int(x) for x in rest[3:]] or MAT_SIZES
        print(f"Running matrix size sweep benchmark\nProfile runtime: {'yes' if prof else 'no'}\nRun QRF: {'yes' if qrf else 'no'}\n")
        run_mat_size_sweep(directory, m_sz, prof, qrf)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
