"""ctypes binding of librlhip.so (the C ABI declared in include/rlhip.h).

Plumbing only: PyTorch supplies device memory and the HIP stream; every FLOP on the path is executed by
the hand-written HIP kernels inside librlhip.so.  There is deliberately NO fallback: if the shared
library is missing or a symbol cannot be resolved this module raises at import/load time.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "librlhip.so"

c_i64 = C.c_int64
c_int = C.c_int
c_char = C.c_char
c_dbl = C.c_double
c_flt = C.c_float
c_vp = C.c_void_p
c_sz = C.c_size_t
u32p = C.POINTER(C.c_uint32)


def _blas3(T):
    return {
        "gemm": [c_vp, c_char, c_char, c_i64, c_i64, c_i64, T, c_vp, c_i64, c_vp, c_i64, T, c_vp, c_i64],
        "syrk": [c_vp, c_char, c_char, c_i64, c_i64, T, c_vp, c_i64, T, c_vp, c_i64],
        "trsm": [c_vp, c_char, c_char, c_char, c_char, c_i64, c_i64, T, c_vp, c_i64, c_vp, c_i64],
        "trmm": [c_vp, c_char, c_char, c_char, c_char, c_i64, c_i64, T, c_vp, c_i64, c_vp, c_i64],
        "trsm_gather": [c_vp, c_char, c_i64, c_i64, T, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64],
        "trsm_gather_range": [c_vp, c_char, c_i64, c_i64, T, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64],
        "potrf": [c_vp, c_char, c_i64, c_vp, c_i64],
        "lange_fro": [c_vp, c_i64, c_i64, c_vp, c_i64, C.POINTER(T)],
        "lacpy": [c_vp, c_char, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64],
        "laset": [c_vp, c_char, c_i64, c_i64, T, T, c_vp, c_i64],
        "saso_apply": [c_vp, c_vp, c_i64, T, c_vp, c_i64, T, c_vp, c_i64],
        "saso_apply_rows": [c_vp, c_vp, c_i64, T, c_vp, c_i64, c_i64, c_i64, T, c_vp, c_i64],
        "saso_dense": [c_vp, c_vp, c_vp],
        "saso_apply_csr": [c_vp, c_vp, c_i64, T, c_vp, c_vp, c_vp, T, c_vp, c_i64, c_i64],
        "col_swap": [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp],
        "geqp3": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp],
        "get_diag": [c_vp, c_i64, c_vp, c_i64, C.POINTER(T)],
        "orhr_col": [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp],
        "gemqrt": [c_vp, c_char, c_char, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64],
        "larft": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64],
        "row_sign": [c_vp, c_i64, c_vp, c_i64, c_vp],
        "tau_from_t": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
        "any_abs_gt": [c_vp, c_i64, c_vp, T, C.POINTER(c_int)],
        "geqrf": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
        "geqrf_q": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64],
        "vrows_explicit": [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64],
        "qrp_partial": [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp],
        "geqp3_steps": [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp],
        "ungqr": [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp],
        "laswp": [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp],
        "getrf": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
        "getrf_piv": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
        "scal_cols": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
        "scal_rows_idx": [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, T],
        "gen_kahan": [c_vp, c_i64, c_i64, c_vp, c_i64, T, T],
        "symmetrize": [c_vp, c_char, c_i64, c_vp, c_i64, c_vp, c_i64],
        "axpby": [c_vp, c_i64, T, c_vp, T, c_vp],
        "csr_spmm": [c_vp, c_char, c_i64, c_i64, c_i64, T, c_vp, c_vp, c_vp, c_vp, c_i64, T, c_vp, c_i64],
        "csr_transpose": [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
        "csr_densify_cols": [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64],
        "add_diag": [c_vp, c_i64, T, c_vp, c_i64],
        "gesdd": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, C.POINTER(c_int)],
        "transpose": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_int],
        "gesvdj": [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, C.POINTER(c_int)],
        "fill_dense": [c_vp, c_int, c_i64, c_i64, c_vp, u32p, u32p, u32p],
        "fill_dense_rows": [c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, u32p, u32p, u32p],
    }


# name -> (restype, argtypes); every symbol include/rlhip.h declares must appear here
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    "rlhip_version": (C.c_char_p, []),
    "rlhip_create": (c_int, [C.POINTER(c_vp), c_int, c_vp, c_int]),
    "rlhip_destroy": (c_int, [c_vp]),
    "rlhip_sync": (c_int, [c_vp]),
    "rlhip_stream": (c_vp, [c_vp]),
    "rlhip_malloc": (c_int, [c_vp, C.POINTER(c_vp), c_sz]),
    "rlhip_free": (c_int, [c_vp, c_vp]),
    "rlhip_memcpy_h2d": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "rlhip_memcpy_d2h": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "rlhip_trim": (c_int, [c_vp]),
    "rlhip_malloc_host": (c_int, [c_vp, C.POINTER(c_vp), c_sz]),
    "rlhip_free_host": (c_int, [c_vp, c_vp]),
    "rlhip_memcpy_d2d": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "rlhip_memset": (c_int, [c_vp, c_vp, c_int, c_sz]),
    "rlhip_reserve_workspace": (c_int, [c_vp, c_sz]),
    "rlhip_workspace_highwater": (c_sz, [c_vp]),
    "rlhip_scratch_mark": (c_sz, [c_vp]),
    "rlhip_scratch_alloc": (c_int, [c_vp, C.POINTER(c_vp), c_sz]),
    "rlhip_scratch_release": (c_int, [c_vp, c_sz]),
    "rlhip_timer_start": (c_int, [c_vp]),
    "rlhip_timer_stop_ms": (c_int, [c_vp, C.POINTER(c_flt)]),
    "rlhip_philox4x32_10": (c_int, [c_vp, c_i64, c_vp, u32p, u32p]),
    "rlhip_gemm_norma_f64": (c_int, [c_vp, c_char, c_char, c_i64, c_i64, c_i64, c_dbl, c_vp, c_i64, c_vp, c_i64, c_dbl,
                                     c_vp, c_i64, C.POINTER(c_dbl), C.POINTER(c_int)]),
    "rlhip_create_side": (c_int, [c_vp, C.POINTER(c_vp)]),
    "rlhip_order_after": (c_int, [c_vp, c_vp]),
    "rlhip_set_qrcp_cols": (c_int, [c_vp, c_int]),
    "rlhip_side_of": (c_int, [c_vp, C.POINTER(c_vp)]),
    "rlhip_gemqrt_head_f64": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "rlhip_gemqrt_head_f32": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "rlhip_gemqrt_tail_f64": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64]),
    "rlhip_gemqrt_tail_f32": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64]),
    "rlhip_norma_collect_f64": (c_int, [c_vp, c_int, C.POINTER(c_dbl)]),
    "rlhip_cholqrq_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_int, C.POINTER(c_int)]),
    "rlhip_cholqrq_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_int, C.POINTER(c_int)]),
    "rlhip_saso_create": (c_int, [c_vp, c_i64, c_i64, c_int, u32p, u32p, u32p, C.POINTER(c_vp)]),
    "rlhip_saso_create_mode": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, u32p, u32p, u32p, C.POINTER(c_vp)]),
    "rlhip_saso_destroy": (c_int, [c_vp, c_vp]),
    "rlhip_col_swap_i64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "rlhip_luqrcp_piv": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "rlhip_path_count": (c_i64, [c_vp, c_int]),
    "rlhip_path_note": (c_int, [c_vp, c_int, c_i64]),
    "rlhip_range_push": (c_int, [C.c_char_p]),
    "rlhip_range_pop": (c_int, []),
    "rlhip_avoid_persistent": (c_int, [c_vp, c_int]),
    "rlhip_set_option": (c_int, [c_vp, c_int, c_i64]),
    "rlhip_get_option": (c_i64, [c_vp, c_int]),
    "rlhip_mfma_peak": (c_int, [c_vp, c_int, c_int, C.POINTER(c_dbl)]),
    "rlhip_dvfs_burn": (c_int, [c_vp, c_int, c_int, c_int, c_int]),
    "rlhip_hbm_read_peak": (c_int, [c_vp, c_vp, c_sz, C.POINTER(c_dbl)]),
}
HOOK = C.CFUNCTYPE(c_int, c_vp, c_vp, c_i64, c_int)
SIGNATURES.update({
    "rlhip_comm_can_load": (c_int, []),
    "rlhip_comm_rccl_origin": (C.c_char_p, []),
    "rlhip_comm_rccl_version": (c_int, []),
    "rlhip_comm_kind": (c_int, [c_vp]),
    "rlhip_comm_unique_id": (c_int, [c_vp]),
    "rlhip_comm_init": (c_int, [c_vp, c_int, c_int, c_vp]),
    "rlhip_comm_set_hook": (c_int, [c_vp, HOOK, c_vp, c_int, c_int]),
    "rlhip_comm_destroy": (c_int, [c_vp]),
    "rlhip_comm_size": (c_int, [c_vp]),
    "rlhip_comm_rank": (c_int, [c_vp]),
    "rlhip_allreduce_sum_f64": (c_int, [c_vp, c_vp, c_i64]),
    "rlhip_allreduce_sum_f32": (c_int, [c_vp, c_vp, c_i64]),
    "rlhip_allreduce_sum_host_f64": (c_int, [c_vp, C.POINTER(c_dbl), c_i64]),
})
dpp = C.POINTER(c_vp)
SIGNATURES.update({
    "rlhip_last_error": (C.c_char_p, []),
    "rlhip_drv_stab_f64": (c_int, [c_vp, c_int, c_int, c_i64, c_i64, c_vp, C.POINTER(c_int)]),
    "rlhip_drv_rs_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, u32p]),
    "rlhip_drv_rf_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp, u32p]),
    "rlhip_drv_qb_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, C.POINTER(c_i64), c_i64, c_dbl, c_i64, c_i64, c_int, c_int,
                                 c_int, c_int, dpp, dpp, u32p]),
    "rlhip_drv_cqrrpt_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_dbl, c_i64, c_dbl, u32p, c_vp,
                                     c_vp, C.POINTER(c_i64), C.POINTER(C.c_long), c_int]),
    "rlhip_drv_hqrrp_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, u32p, c_vp]),
    "rlhip_drv_hqrrp_timed_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, u32p, C.POINTER(c_dbl)]),
    "rlhip_drv_bqrrp_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_dbl, c_i64, c_i64, c_dbl, c_vp, c_vp, u32p, c_vp, c_vp,
                                    C.POINTER(c_i64), C.POINTER(C.c_long), c_int, c_int, c_int]),
    "rlhip_drv_abrik_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_dbl, c_i64, dpp, dpp, dpp, u32p, C.POINTER(c_i64),
                                    C.POINTER(c_i64), C.POINTER(c_dbl), c_int]),
    "rlhip_drv_cqrrt_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_dbl, c_i64, c_dbl, u32p, c_vp, c_vp]),
    "rlhip_drv_bqrrp_gpu_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_dbl, c_vp, c_vp, C.POINTER(c_i64),
                                        C.POINTER(C.c_long)]),
    "rlhip_drv_bqrrp_gpu_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_flt, c_vp, c_vp, C.POINTER(c_i64),
                                        C.POINTER(C.c_long)]),
    "rlhip_drv_cqrrpt_gpu_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_dbl, c_i64, c_dbl, c_int, u32p, c_vp,
                                         C.POINTER(c_i64), C.POINTER(C.c_long)]),
    "rlhip_drv_cqrrpt_gpu_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_flt, c_i64, c_flt, c_int, u32p, c_vp,
                                         C.POINTER(c_i64), C.POINTER(C.c_long)]),
    "rlhip_drv_rsvd_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, C.POINTER(c_i64), c_i64, c_dbl, c_i64, c_i64, c_int,
                                   c_int, c_int, c_int, dpp, dpp, dpp, u32p, C.POINTER(c_int)]),
})


class LinOpDesc(C.Structure):
    """rlhip_linop_desc (include/rlhip_drivers.h)"""
    _fields_ = [("kind", c_int), ("rows", c_i64), ("cols", c_i64), ("dense", c_vp), ("ld", c_i64), ("nnz", c_i64),
                ("rowptr", c_vp), ("colidx", c_vp), ("vals", c_vp)]


_ldp = C.POINTER(LinOpDesc)
SIGNATURES.update({
    "rlhip_drv_qr_linops_f64": (c_int, [c_vp, c_int, _ldp, _ldp, c_vp, c_i64, c_i64, dpp, c_dbl, c_i64, c_int, u32p, c_vp, c_vp]),
    "rlhip_drv_abrik_linop_f64": (c_int, [c_vp, _ldp, _ldp, c_i64, c_dbl, c_i64, dpp, dpp, dpp, u32p, C.POINTER(c_i64),
                                          C.POINTER(c_i64), C.POINTER(c_dbl), c_int]),
    "rlhip_drv_abrik_linop_timed_f64": (c_int, [c_vp, _ldp, c_i64, c_dbl, c_i64, dpp, dpp, dpp, u32p, C.POINTER(c_i64), C.POINTER(c_i64),
                                                C.POINTER(c_dbl), c_int, C.POINTER(C.c_long)]),
    "rlhip_linop_apply_f64": (c_int, [c_vp, _ldp, _ldp, c_char, c_char, c_i64, c_i64, c_i64, c_dbl, c_vp, c_i64, c_dbl, c_vp, c_i64]),
    "rlhip_linop_apply_view_f64": (c_int, [c_vp, _ldp, _ldp, c_int, C.POINTER(c_i64), c_char, c_char, c_i64, c_i64, c_i64, c_dbl, c_vp, c_i64, c_dbl, c_vp, c_i64]),
    "rlhip_regsym_apply_f64": (c_int, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_dbl), c_i64, c_int, c_i64, c_dbl, c_vp, c_i64, c_dbl, c_vp, c_i64]),
})
SIGNATURES["rlhip_drv_revd2_f64"] = (c_int, [c_vp, c_char, c_i64, c_vp, C.POINTER(c_i64), c_dbl, c_i64, c_i64, c_int, c_int, dpp, dpp, u32p,
                                               C.POINTER(c_dbl)])
SIGNATURES["rlhip_drv_syrf_f64"] = (c_int, [c_vp, c_char, c_i64, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, u32p])
SIGNATURES["rlhip_drv_mat_gen_f64"] = (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_dbl, c_dbl, c_dbl, c_int, c_dbl, c_dbl, c_dbl, c_int, c_vp, u32p,
                                                 C.POINTER(c_i64)])
for _name in ("stab", "rsvd", "cqrrpt", "hqrrp", "bqrrp", "qr_linops", "mat_gen"):      # fp32 instantiations: same shapes, float scalars
    _rt, _args = SIGNATURES[f"rlhip_drv_{_name}_f64"]
    SIGNATURES[f"rlhip_drv_{_name}_f32"] = (_rt, [c_flt if a is c_dbl else a for a in _args])
for _suf, _T in (("f64", c_dbl), ("f32", c_flt)):
    for _name, _args in _blas3(_T).items():
        SIGNATURES[f"rlhip_{_name}_{_suf}"] = (c_int, _args)

_lib = None


def load():
    """Load librlhip.so, building nothing.  Raises if it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). randlapack_amd has no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else C.DEFAULT_MODE)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the binding drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RlhipError(RuntimeError):
    pass


def check(rc: int, what: str) -> int:
    """Negative return codes are argument / runtime errors -> raise.  Positive codes (LAPACK info) pass through."""
    if rc < 0:
        raise RlhipError(f"{what} failed with code {rc}")
    return rc


def lib_sha256() -> str:
    """fingerprint of the librlhip.so this module loads: counter files (scripts/pmc_all.py) record it, the bench lines compare it -- a counter
    file taken on THIS build is quoted even when the profiler's own overhead moved the launch time by more than 5 %"""
    import hashlib

    h = hashlib.sha256()
    with open(LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()
