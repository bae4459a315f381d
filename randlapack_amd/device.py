"""Thin Python handles over the rlhip C ABI for tests / bench / smoke.

A column-major m x n matrix is held as a torch tensor `t` of shape (n, m), contiguous, so that
element (i, j) is `t[j, i]` and the leading dimension is m.  `cm_from_numpy` / `cm_to_numpy` convert
from/to ordinary (m, n) numpy arrays.  Nothing in here computes: it forwards pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_TORCH_DT = None


def _torch():
    import torch

    return torch


def _suffix(t):
    torch = _torch()
    if t.dtype == torch.float64:
        return "f64", C.c_double
    if t.dtype == torch.float32:
        return "f32", C.c_float
    raise TypeError(f"unsupported dtype {t.dtype}")


def cm_from_numpy(a: np.ndarray, device="cuda:0"):
    """(m, n) numpy -> column-major device tensor of shape (n, m)."""
    torch = _torch()
    a = np.asarray(a)
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(device)


def cm_to_numpy(t) -> np.ndarray:
    """column-major device tensor (n, m) -> (m, n) numpy."""
    return t.detach().cpu().numpy().T.copy()


def cm_empty(m: int, n: int, dtype=None, device="cuda:0"):
    torch = _torch()
    return torch.empty((n, m), dtype=dtype or torch.float64, device=device)


def cm_zeros(m: int, n: int, dtype=None, device="cuda:0"):
    torch = _torch()
    return torch.zeros((n, m), dtype=dtype or torch.float64, device=device)


class Context:
    """One rlhip context bound to torch's CURRENT HIP stream on `device` (so torch.cuda.Event timing and
    torch allocations are ordered with the kernels)."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("randlapack_amd needs a HIP device (torch.cuda.is_available() is False)")
        self.lib = _lib.load()
        self.device = device
        torch.cuda.set_device(device)
        stream = torch.cuda.current_stream(device).cuda_stream if use_torch_stream else 0
        h = C.c_void_p()
        _lib.check(self.lib.rlhip_create(C.byref(h), device, C.c_void_p(stream), 0 if use_torch_stream else 1),
                   "rlhip_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.rlhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(self.lib.rlhip_sync(self.h), "rlhip_sync")

    # ---- timing on the context stream
    def timer_start(self):
        _lib.check(self.lib.rlhip_timer_start(self.h), "timer_start")

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        _lib.check(self.lib.rlhip_timer_stop_ms(self.h, C.byref(ms)), "timer_stop")
        return float(ms.value)

    def reserve_workspace(self, nbytes: int):
        _lib.check(self.lib.rlhip_reserve_workspace(self.h, nbytes), "reserve_workspace")

    # ---- RNG
    @staticmethod
    def _u32(vals):
        arr = (C.c_uint32 * len(vals))(*[int(v) & 0xFFFFFFFF for v in vals])
        return arr

    def philox(self, nblocks: int, ctr, key):
        torch = _torch()
        out = torch.empty(4 * nblocks, dtype=torch.int32, device=f"cuda:{self.device}")
        _lib.check(
            self.lib.rlhip_philox4x32_10(self.h, nblocks, out.data_ptr(), self._u32(ctr), self._u32(key)), "philox"
        )
        return out.cpu().numpy().view(np.uint32)

    def fill_dense(self, buf, rows: int, cols: int, ctr=(0, 0, 0, 0), key=(0, 0), dist: int = 0):
        suf, _ = _suffix(buf)
        nxt = (C.c_uint32 * 4)()
        fn = getattr(self.lib, f"rlhip_fill_dense_{suf}")
        _lib.check(fn(self.h, dist, rows, cols, buf.data_ptr(), self._u32(ctr), self._u32(key), nxt), "fill_dense")
        return tuple(int(x) for x in nxt)

    # ---- BLAS-3 (column-major tensors, see module docstring)
    def gemm(self, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, Cm, ldc):
        suf, T = _suffix(Cm)
        fn = getattr(self.lib, f"rlhip_gemm_{suf}")
        return _lib.check(
            fn(self.h, ta.encode(), tb.encode(), m, n, k, T(alpha), A.data_ptr(), lda, B.data_ptr(), ldb, T(beta),
               Cm.data_ptr(), ldc),
            "gemm",
        )

    def syrk(self, uplo, trans, n, k, alpha, A, lda, beta, Cm, ldc):
        suf, T = _suffix(Cm)
        fn = getattr(self.lib, f"rlhip_syrk_{suf}")
        return _lib.check(
            fn(self.h, uplo.encode(), trans.encode(), n, k, T(alpha), A.data_ptr(), lda, T(beta), Cm.data_ptr(), ldc),
            "syrk",
        )

    def trsm(self, m, n, alpha, A, lda, B, ldb, diag="N"):
        suf, T = _suffix(B)
        fn = getattr(self.lib, f"rlhip_trsm_{suf}")
        return _lib.check(
            fn(self.h, b"R", b"U", b"N", diag.encode(), m, n, T(alpha), A.data_ptr(), lda, B.data_ptr(), ldb), "trsm"
        )

    def trsm_gather(self, m, n, alpha, A, lda, Bsrc, ldsrc, jpvt, B, ldb, diag="N"):
        """B = alpha * Bsrc[:, jpvt - 1] * inv(A), out of place (rlhip_trsm_gather_*); jpvt: device int64 tensor (1-based) or None"""
        suf, T = _suffix(B)
        fn = getattr(self.lib, f"rlhip_trsm_gather_{suf}")
        return _lib.check(
            fn(self.h, diag.encode(), m, n, T(alpha), A.data_ptr(), lda, Bsrc.data_ptr(), ldsrc, jpvt.data_ptr() if jpvt is not None else None,
               B.data_ptr(), ldb), "trsm_gather")

    def trmm(self, m, n, alpha, A, lda, B, ldb, diag="N"):
        suf, T = _suffix(B)
        fn = getattr(self.lib, f"rlhip_trmm_{suf}")
        return _lib.check(
            fn(self.h, b"R", b"U", b"N", diag.encode(), m, n, T(alpha), A.data_ptr(), lda, B.data_ptr(), ldb), "trmm"
        )

    def potrf(self, n, A, lda) -> int:
        suf, _ = _suffix(A)
        fn = getattr(self.lib, f"rlhip_potrf_{suf}")
        return _lib.check(fn(self.h, b"U", n, A.data_ptr(), lda), "potrf")

    def lange_fro(self, m, n, A, lda) -> float:
        suf, T = _suffix(A)
        res = T()
        fn = getattr(self.lib, f"rlhip_lange_fro_{suf}")
        _lib.check(fn(self.h, m, n, A.data_ptr(), lda, C.byref(res)), "lange")
        return float(res.value)

    def lacpy(self, uplo, m, n, A, lda, B, ldb):
        suf, _ = _suffix(A)
        fn = getattr(self.lib, f"rlhip_lacpy_{suf}")
        return _lib.check(fn(self.h, uplo.encode(), m, n, A.data_ptr(), lda, B.data_ptr(), ldb), "lacpy")

    def laset(self, uplo, m, n, offd, diag, A, lda):
        suf, T = _suffix(A)
        fn = getattr(self.lib, f"rlhip_laset_{suf}")
        return _lib.check(fn(self.h, uplo.encode(), m, n, T(offd), T(diag), A.data_ptr(), lda), "laset")

    def gesvdj(self, m, n, A, lda, S, VT, ldvt):
        suf, _ = _suffix(A)
        sweeps = C.c_int()
        fn = getattr(self.lib, f"rlhip_gesvdj_{suf}")
        info = _lib.check(fn(self.h, m, n, A.data_ptr(), lda, S.data_ptr(), VT.data_ptr(), ldvt, C.byref(sweeps)),
                          "gesvdj")
        return info, int(sweeps.value)

    # ---- diagnostics
    def path_count(self, which: int) -> int:
        return int(self.lib.rlhip_path_count(self.h, which))

    # ---- options (include/rlhip.h: enum rlhip_option; -1 restores the default)
    OPT = dict(cholqrq_one_stream=0, gesdd_gram=1, jacobi_persist=2, trsm_xasm=3, saso_mode=4, hqrrp_tall_panel=5,
               bqrrp_lookahead_min_elems=6, bqrrp_cholqr_fallback=7, cqrrpt_fold_pivoting=8, cqrrpt_split_qrcp=9, sparse_sketch_densify=10, jacobi_clock_holders=11)

    def set_option(self, name: str, value: int) -> None:
        _lib.check(self.lib.rlhip_set_option(self.h, self.OPT[name], int(value)), "set_option")

    def get_option(self, name: str) -> int:
        return int(self.lib.rlhip_get_option(self.h, self.OPT[name]))

    def options(self, **kw):
        """context manager: set options for the duration of a block, then restore what was there"""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return _cm()

    def gemm_norma(self, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, Cm, ldc):
        """gemm + ||A||_F in one pass (rlhip_gemm_norma_f64); returns (norm, fused flag)"""
        nrm, fused = C.c_double(), C.c_int()
        _lib.check(self.lib.rlhip_gemm_norma_f64(self.h, ta.encode(), tb.encode(), m, n, k, alpha, A.data_ptr(), lda, B.data_ptr(), ldb,
                                                 beta, Cm.data_ptr(), ldc, C.byref(nrm), C.byref(fused)), "gemm_norma")
        return float(nrm.value), int(fused.value)

    def mfma_peak(self, is_f64=True, iters=20000) -> float:
        tf = C.c_double()
        _lib.check(self.lib.rlhip_mfma_peak(self.h, 1 if is_f64 else 0, iters, C.byref(tf)), "mfma_peak")
        return float(tf.value)

    def hbm_read_peak(self, buf) -> float:
        g = C.c_double()
        nbytes = buf.numel() * buf.element_size()
        _lib.check(self.lib.rlhip_hbm_read_peak(self.h, buf.data_ptr(), nbytes, C.byref(g)), "hbm_read_peak")
        return float(g.value)


# ------------------------------------------------------------------------------------------------------
# driver-level entry points (include/rlhip_drivers.h): the C++ RS/RF/QB/RSVD objects behind a C ABI
# ------------------------------------------------------------------------------------------------------
def _state_arr(ctr, key):
    return (C.c_uint32 * 6)(*[int(v) & 0xFFFFFFFF for v in list(ctr) + list(key)])


class _Owned:
    """a callee-allocated device block (rlhip_malloc) exposed through __cuda_array_interface__; returned to the library's pool when the
    last tensor viewing it goes away (the reference's caller free()s these arrays: rl_rsvd.hh:139-143)"""

    def __init__(self, ctx: "Context", ptr: int, shape, typestr: str):
        self._ctx, self._ptr = ctx, ptr
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            if self._ptr and getattr(self._ctx, "h", None):
                self._ctx.lib.rlhip_free(self._ctx.h, C.c_void_p(self._ptr))
        except Exception:
            pass
        self._ptr = 0


def _adopt(ctx: "Context", ptr: C.c_void_p, rows: int, cols: int, dtype=None):
    """a callee-allocated column-major device block as a torch tensor of shape (cols, rows) WITHOUT a copy: the tensor views the block and
    the block goes back to the library's pool when the tensor dies"""
    torch = _torch()
    dtype = dtype or torch.float64
    if rows * cols <= 0 or not ptr.value:
        if ptr.value:
            _lib.check(ctx.lib.rlhip_free(ctx.h, ptr), "rlhip_free")
        return torch.empty((cols, rows), dtype=dtype, device=f"cuda:{ctx.device}")
    owner = _Owned(ctx, int(ptr.value), (cols, rows), "<f8" if dtype == torch.float64 else "<f4")
    return torch.as_tensor(owner, device=f"cuda:{ctx.device}")


def _drv_check(ctx, rc, what):
    if rc <= -100:
        raise _lib.RlhipError(f"{what}: {ctx.lib.rlhip_last_error().decode()} (code {rc})")
    return rc


def drv_stab(ctx: Context, kind: int, A, m: int, k: int, cond_check: bool = False):
    cf = C.c_int(0)
    rc = getattr(ctx.lib, f"rlhip_drv_stab_{_suffix(A)[0]}")(ctx.h, kind, int(cond_check), m, k, A.data_ptr(), C.byref(cf))
    return _drv_check(ctx, rc, "stab"), bool(cf.value)


def drv_rs(ctx: Context, A, m, n, k, p, q, stab_kind=0, ctr=(0, 0, 0, 0), key=(0, 0)):
    Om = cm_empty(n, k, device=f"cuda:{ctx.device}")
    st = _state_arr(ctr, key)
    rc = ctx.lib.rlhip_drv_rs_f64(ctx.h, m, n, A.data_ptr(), k, p, q, stab_kind, Om.data_ptr(), st)
    return _drv_check(ctx, rc, "rs"), Om, tuple(int(x) for x in st[:4])


def drv_rf(ctx: Context, A, m, n, k, p, q, rs_stab=0, orth_kind=0, ctr=(0, 0, 0, 0), key=(0, 0)):
    Q = cm_empty(m, k, device=f"cuda:{ctx.device}")
    st = _state_arr(ctr, key)
    rc = ctx.lib.rlhip_drv_rf_f64(ctx.h, m, n, A.data_ptr(), k, p, q, rs_stab, orth_kind, Q.data_ptr(), st)
    return _drv_check(ctx, rc, "rf"), Q, tuple(int(x) for x in st[:4])


def drv_qb(ctx: Context, A, m, n, k, b_sz, tol, p, q, rs_stab=0, rf_orth=0, qb_orth=0, orth_check=False,
           ctr=(0, 0, 0, 0), key=(0, 0)):
    kk = C.c_int64(k)
    Qp, Bp = C.c_void_p(), C.c_void_p()
    st = _state_arr(ctr, key)
    rc = ctx.lib.rlhip_drv_qb_f64(ctx.h, m, n, A.data_ptr(), C.byref(kk), b_sz, tol, p, q, rs_stab, rf_orth, qb_orth,
                                  int(orth_check), C.byref(Qp), C.byref(Bp), st)
    _drv_check(ctx, rc, "qb")
    Q = _adopt(ctx, Qp, m, k)
    BT = _adopt(ctx, Bp, n, k)
    kf = int(kk.value)
    return rc, kf, Q[:kf], BT[:kf], tuple(int(x) for x in st[:4])


def drv_rsvd(ctx: Context, A, m, n, k, b_sz, tol, p, q, rs_stab=0, rf_orth=0, qb_orth=0, orth_check=False,
             ctr=(0, 0, 0, 0), key=(0, 0), keep_on_device=True):
    """RSVD::call.  Returns dict(rc, qb_rc, k, U, S, V, next_ctr); U/V are column-major tensors (k, m)/(k, n)."""
    kk = C.c_int64(k)
    Up, Sp, Vp = C.c_void_p(), C.c_void_p(), C.c_void_p()
    qrc = C.c_int(0)
    st = _state_arr(ctr, key)
    suf, _ = _suffix(A)
    rc = getattr(ctx.lib, f"rlhip_drv_rsvd_{suf}")(ctx.h, m, n, A.data_ptr(), C.byref(kk), b_sz, tol, p, q, rs_stab, rf_orth, qb_orth,
                                                   int(orth_check), C.byref(Up), C.byref(Sp), C.byref(Vp), st, C.byref(qrc))
    _drv_check(ctx, rc, "rsvd")
    kf = int(kk.value)
    kal = max(kf, 1)
    U = _adopt(ctx, Up, m, kal, A.dtype)
    S = _adopt(ctx, Sp, kal, 1, A.dtype).reshape(-1)
    V = _adopt(ctx, Vp, n, kal, A.dtype)
    return dict(rc=rc, qb_rc=int(qrc.value), k=kf, U=U[:kf], S=S[:kf], V=V[:kf], next_ctr=tuple(int(x) for x in st[:4]))


def drv_cqrrt(ctx: Context, A, m, n, d_factor=1.25, nnz=2, eps=None, ctr=(0, 0, 0, 0), key=(0, 0), sketch_in=None, want_sketch=False):
    """CQRRT::call.  A (column-major tensor (n, m)) is overwritten by Q.  Returns dict(rc, R, next_ctr[, sketch])."""
    dev = f"cuda:{ctx.device}"
    if eps is None:
        eps = float(np.finfo(np.float64).eps ** 0.85)
    d = int(d_factor * n)
    R = cm_zeros(n, n, device=dev)
    sk_out = cm_empty(d, n, device=dev) if want_sketch else None
    st = _state_arr(ctr, key)
    rc = ctx.lib.rlhip_drv_cqrrt_f64(ctx.h, m, n, A.data_ptr(), m, R.data_ptr(), n, d_factor, nnz, eps, st,
                                     sketch_in.data_ptr() if sketch_in is not None else None,
                                     sk_out.data_ptr() if sk_out is not None else None)
    _drv_check(ctx, rc, "cqrrt")
    out = dict(rc=rc, R=R, next_ctr=tuple(int(x) for x in st[:4]))
    if want_sketch:
        out["sketch"] = sk_out
    return out


def drv_abrik(ctx: Context, A, m, n, k, tol, max_krylov_iters=0, ctr=(0, 0, 0, 0), key=(0, 0), qr_exp=-1):
    """ABRIK::call on the dense operator A (column-major tensor (n, m), not modified).  Returns dict(rc, U, S, V, triplets, iters,
    norm_R_end, next_ctr); U/V are column-major tensors (triplets, m)/(triplets, n)."""
    Up, Sp, Vp = C.c_void_p(), C.c_void_p(), C.c_void_p()
    trip, iters = C.c_int64(0), C.c_int64(0)
    nre = C.c_double(0)
    st = _state_arr(ctr, key)
    rc = ctx.lib.rlhip_drv_abrik_f64(ctx.h, m, n, A.data_ptr(), m, k, tol, max_krylov_iters, C.byref(Up), C.byref(Sp), C.byref(Vp), st,
                                     C.byref(trip), C.byref(iters), C.byref(nre), qr_exp)
    _drv_check(ctx, rc, "abrik")
    t = int(trip.value)
    U = _adopt(ctx, Up, m, t)
    S = _adopt(ctx, Sp, t, 1).reshape(-1)
    V = _adopt(ctx, Vp, n, t)
    return dict(rc=rc, U=U, S=S, V=V, triplets=t, iters=int(iters.value), norm_R_end=float(nre.value),
                next_ctr=tuple(int(x) for x in st[:4]))


def drv_revd2(ctx: Context, A, m, k, tol, uplo="U", syps_passes=2, passes_per_stab=1, error_est_p=10, orth_kind=1, ctr=(0, 0, 0, 0),
              key=(0, 0)):
    """REVD2::call on the symmetric matrix whose `uplo` triangle is stored in A (column-major tensor (m, m)).
    Returns dict(rc, k, V, eigvals, err, next_ctr); V is a column-major tensor (k, m)."""
    Vp, Ep = C.c_void_p(), C.c_void_p()
    kk = C.c_int64(k)
    err = C.c_double(0)
    st = _state_arr(ctr, key)
    rc = ctx.lib.rlhip_drv_revd2_f64(ctx.h, uplo.encode(), m, A.data_ptr(), C.byref(kk), tol, syps_passes, passes_per_stab, error_est_p,
                                     orth_kind, C.byref(Vp), C.byref(Ep), st, C.byref(err))
    _drv_check(ctx, rc, "revd2")
    kf = int(kk.value)
    return dict(rc=rc, k=kf, V=_adopt(ctx, Vp, m, kf), eigvals=_adopt(ctx, Ep, kf, 1).reshape(-1), err=float(err.value),
                next_ctr=tuple(int(x) for x in st[:4]))


def drv_syrf(ctx: Context, A, m, k, uplo="U", syps_passes=2, passes_per_stab=1, orth_kind=1, ctr=(0, 0, 0, 0), key=(0, 0)):
    """SYRF::call.  Returns dict(rc, Q, next_ctr) with Q a column-major tensor (k, m)."""
    Q = cm_zeros(m, k, device=f"cuda:{ctx.device}")
    st = _state_arr(ctr, key)
    rc = ctx.lib.rlhip_drv_syrf_f64(ctx.h, uplo.encode(), m, A.data_ptr(), k, syps_passes, passes_per_stab, orth_kind, Q.data_ptr(), st)
    _drv_check(ctx, rc, "syrf")
    return dict(rc=rc, Q=Q, next_ctr=tuple(int(x) for x in st[:4]))


MAT_TYPES = {"polynomial": 0, "exponential": 1, "gaussian": 2, "step": 3, "spiked": 4, "adverserial": 5, "bad_cholqr": 6, "kahan": 7}


def drv_mat_gen(ctx: Context, m_type, m, n, rank=None, cond_num=1.0, scaling=1.0, exponent=1.0, diag=False, theta=1.0, perturb=1.0,
                frac_spectrum_one=0.1, check_true_rank=False, ctr=(0, 0, 0, 0), key=(0, 0), dtype=None):
    """gen::mat_gen into HBM.  Returns dict(A, rank, next_ctr); A is a column-major tensor (n, m) ((rank, rank) when diag)."""
    torch = _torch()
    dev = f"cuda:{ctx.device}"
    dtype = dtype or torch.float64
    rank = n if rank is None else rank
    A = cm_zeros(rank, rank, dtype=dtype, device=dev) if diag else cm_zeros(m, n, dtype=dtype, device=dev)
    st = _state_arr(ctr, key)
    r_out = C.c_int64(rank)
    suf = "f64" if dtype == torch.float64 else "f32"
    rc = getattr(ctx.lib, f"rlhip_drv_mat_gen_{suf}")(ctx.h, MAT_TYPES[m_type] if isinstance(m_type, str) else m_type, m, n, rank, cond_num,
                                                      scaling, exponent, 1 if diag else 0, theta, perturb, frac_spectrum_one,
                                                      1 if check_true_rank else 0, A.data_ptr(), st, C.byref(r_out))
    _drv_check(ctx, rc, "mat_gen")
    return dict(A=A, rank=int(r_out.value), next_ctr=tuple(int(x) for x in st[:4]))


class DenseOperator:
    """linops::DenseLinOp over a column-major device tensor (n, m) (rows m, cols n)."""

    def __init__(self, A, m, n):
        self.A, self.rows, self.cols = A, m, n
        self.dtype = A.dtype

    def desc(self):
        from ._lib import LinOpDesc
        return LinOpDesc(0, self.rows, self.cols, self.A.data_ptr(), self.rows, 0, None, None, None)


class CsrOperator:
    """linops::SparseLinOp over device CSR arrays (int64 rowptr / colidx).  Build from scipy with `from_scipy`."""

    def __init__(self, rows, cols, rowptr, colidx, vals):
        self.rows, self.cols = rows, cols
        self.rowptr, self.colidx, self.vals = rowptr, colidx, vals
        self.dtype = vals.dtype

    @classmethod
    def from_scipy(cls, sp, device="cuda:0", dtype=None):
        torch = _torch()
        csr = sp.tocsr()
        dt = dtype or torch.float64
        return cls(csr.shape[0], csr.shape[1], torch.as_tensor(csr.indptr.astype(np.int64), device=device),
                   torch.as_tensor(csr.indices.astype(np.int64), device=device),
                   torch.as_tensor(csr.data, device=device).to(dt))

    def desc(self):
        from ._lib import LinOpDesc
        return LinOpDesc(1, self.rows, self.cols, None, 0, int(self.vals.numel()), self.rowptr.data_ptr(), self.colidx.data_ptr(),
                         self.vals.data_ptr())


class CscOperator(CsrOperator):
    """linops::SparseLinOp::from_csc: the operator given by its columns (colptr, row indices, values)"""

    @classmethod
    def from_scipy(cls, sp, device="cuda:0", dtype=None):
        torch = _torch()
        csc = sp.tocsc()
        dt = dtype or torch.float64
        return cls(csc.shape[0], csc.shape[1], torch.as_tensor(csc.indptr.astype(np.int64), device=device),
                   torch.as_tensor(csc.indices.astype(np.int64), device=device), torch.as_tensor(csc.data, device=device).to(dt))

    def desc(self):
        from ._lib import LinOpDesc
        return LinOpDesc(2, self.rows, self.cols, None, 0, int(self.vals.numel()), self.rowptr.data_ptr(), self.colidx.data_ptr(), self.vals.data_ptr())


class CooOperator(CsrOperator):
    """linops::SparseLinOp::from_coo: (row, col, value) triplets in any order, duplicates summed"""

    @classmethod
    def from_triplets(cls, rows, cols, ri, ci, v, device="cuda:0", dtype=None):
        torch = _torch()
        dt = dtype or torch.float64
        return cls(rows, cols, torch.as_tensor(np.asarray(ri, dtype=np.int64), device=device), torch.as_tensor(np.asarray(ci, dtype=np.int64), device=device),
                   torch.as_tensor(np.asarray(v), device=device).to(dt))

    def desc(self):
        from ._lib import LinOpDesc
        return LinOpDesc(3, self.rows, self.cols, None, 0, int(self.vals.numel()), self.rowptr.data_ptr(), self.colidx.data_ptr(), self.vals.data_ptr())


def _op_descs(op):
    """op: DenseOperator | CsrOperator | (left, right) -> (left desc, right desc or None, rows, cols, dtype)"""
    if isinstance(op, tuple):
        left, right = op
        return left.desc(), right.desc(), left.rows, right.cols, left.dtype
    return op.desc(), None, op.rows, op.cols, op.dtype


QR_LINOPS_ALGS = {"cholqr": 0, "scholqr3": 1, "scholqr3_basic": 2, "cqrrt": 3}


def drv_qr_linops(ctx: Context, alg, op, block_size=0, want_Q=False, d_factor=2.0, nnz=2, use_dense_sketch=False, ctr=(0, 0, 0, 0),
                  key=(0, 0), sketch_in=None, want_sketch=False):
    """CholQR_linops / sCholQR3_linops / sCholQR3_linops_basic / CQRRT_linops ::call on a dense, CSR or composite (tuple) operator.
    Returns dict(rc, R[, Q][, sketch], next_ctr); R and Q are column-major tensors (n, n) / (n, m)."""
    torch = _torch()
    dev = f"cuda:{ctx.device}"
    ld, rd, m, n, dtype = _op_descs(op)
    suf = "f64" if dtype == torch.float64 else "f32"
    R = cm_zeros(n, n, dtype=dtype, device=dev)
    d = int(d_factor * n)
    sk_out = cm_empty(d, n, dtype=dtype, device=dev) if want_sketch else None
    Qp = C.c_void_p()
    st = _state_arr(ctr, key)
    rc = getattr(ctx.lib, f"rlhip_drv_qr_linops_{suf}")(
        ctx.h, QR_LINOPS_ALGS[alg] if isinstance(alg, str) else alg, C.byref(ld), C.byref(rd) if rd is not None else None, R.data_ptr(), n,
        block_size, C.byref(Qp) if want_Q else None, d_factor, nnz, 1 if use_dense_sketch else 0, st,
        sketch_in.data_ptr() if sketch_in is not None else None, sk_out.data_ptr() if sk_out is not None else None)
    _drv_check(ctx, rc, "qr_linops")
    out = dict(rc=rc, R=R, next_ctr=tuple(int(x) for x in st[:4]))
    if want_Q:
        out["Q"] = _adopt(ctx, Qp, m, n, dtype=dtype) if Qp.value else None
    if want_sketch:
        out["sketch"] = sk_out
    return out


ABRIK_TIMES = ("allocation", "get_factors", "ungqr", "reorth", "qr", "gemm_A", "main_loop", "sketching", "r_cpy", "s_cpy", "norm", "rest", "total")


def drv_abrik_linop(ctx: Context, op, k, tol, max_krylov_iters=0, ctr=(0, 0, 0, 0), key=(0, 0), qr_exp=-1, timing=False):
    """ABRIK::call on a DenseOperator / CsrOperator.  Same outputs as drv_abrik; timing=True arms ABRIK's subroutine timers and adds
    times_us (the 13 entries of ABRIK::times, rl_abrik.hh:733-734, names in ABRIK_TIMES)."""
    ld, rd, m, n, _ = _op_descs(op)
    Up, Sp, Vp = C.c_void_p(), C.c_void_p(), C.c_void_p()
    trip, iters = C.c_int64(0), C.c_int64(0)
    nre = C.c_double(0)
    st = _state_arr(ctr, key)
    times = (C.c_long * 13)() if timing else None
    if timing:
        if rd is not None:
            raise ValueError("ABRIK runs on single operators only")
        rc = ctx.lib.rlhip_drv_abrik_linop_timed_f64(ctx.h, C.byref(ld), k, tol, max_krylov_iters, C.byref(Up), C.byref(Sp), C.byref(Vp), st,
                                                     C.byref(trip), C.byref(iters), C.byref(nre), qr_exp, times)
    else:
        rc = ctx.lib.rlhip_drv_abrik_linop_f64(ctx.h, C.byref(ld), C.byref(rd) if rd is not None else None, k, tol, max_krylov_iters,
                                               C.byref(Up), C.byref(Sp), C.byref(Vp), st, C.byref(trip), C.byref(iters), C.byref(nre), qr_exp)
    _drv_check(ctx, rc, "abrik_linop")
    t = int(trip.value)
    out = dict(rc=rc, U=_adopt(ctx, Up, m, t), S=_adopt(ctx, Sp, t, 1).reshape(-1), V=_adopt(ctx, Vp, n, t), triplets=t,
               iters=int(iters.value), norm_R_end=float(nre.value), next_ctr=tuple(int(x) for x in st[:4]))
    if timing:
        out["times_us"] = [int(x) for x in times]
    return out


def linop_apply(ctx: Context, op, side, trans, B, m, n, k, alpha=1.0, beta=0.0, C_in=None):
    """C (m x n) = alpha * op(A) * B + beta * C (side 'L') or alpha * B * op(A) + beta * C (side 'R'); column-major tensors."""
    dev = f"cuda:{ctx.device}"
    ld, rd, _, _, _ = _op_descs(op)
    Cout = C_in if C_in is not None else cm_zeros(m, n, device=dev)
    ldb = B.shape[1]
    rc = ctx.lib.rlhip_linop_apply_f64(ctx.h, C.byref(ld), C.byref(rd) if rd is not None else None, side.encode(), trans.encode(), m, n, k,
                                       alpha, B.data_ptr(), ldb, beta, Cout.data_ptr(), m)
    _drv_check(ctx, rc, "linop_apply")
    return Cout


VIEW_HOW = {"row_block": 0, "col_block": 1, "submatrix": 2}


def linop_apply_view(ctx: Context, op, how, view, side, trans, B, m, n, k, alpha=1.0, beta=0.0, C_in=None):
    """the same product with a block view of the operator: how in VIEW_HOW, view = (row_start, col_start, row_count, col_count)"""
    dev = f"cuda:{ctx.device}"
    ld, rd, _, _, _ = _op_descs(op)
    Cout = C_in if C_in is not None else cm_zeros(m, n, device=dev)
    v = (C.c_int64 * 4)(*[int(x) for x in view])
    rc = ctx.lib.rlhip_linop_apply_view_f64(ctx.h, C.byref(ld), C.byref(rd) if rd is not None else None, VIEW_HOW[how], v, side.encode(), trans.encode(),
                                            m, n, k, alpha, B.data_ptr(), B.shape[1], beta, Cout.data_ptr(), m)
    _drv_check(ctx, rc, "linop_apply_view")
    return Cout


def regsym_apply(ctx: Context, A, dim, regs, eval_includes_reg, B, n, alpha=1.0, beta=0.0, C_in=None, lda=None):
    """linops::RegExplicitSymLinOp: C = alpha (A + mu_i I) B + beta C, A by its upper triangle (column-major tensor (dim, lda))"""
    dev = f"cuda:{ctx.device}"
    Cout = C_in if C_in is not None else cm_zeros(dim, n, device=dev)
    rg = (C.c_double * max(len(regs), 1))(*[float(x) for x in regs])
    rc = ctx.lib.rlhip_regsym_apply_f64(ctx.h, dim, A.data_ptr(), lda or dim, rg, len(regs), int(bool(eval_includes_reg)), n, alpha, B.data_ptr(),
                                        B.shape[1], beta, Cout.data_ptr(), dim)
    _drv_check(ctx, rc, "regsym_apply")
    return Cout


def drv_hqrrp(ctx: Context, A, m, n, nb_alg=64, pp=10, panel_pivoting=1, qr_type=0, ctr=(0, 0, 0, 0), key=(0, 0), want_G=False, m_global=None):
    """hqrrp: A (column-major tensor (n, m)) is overwritten in GEQP3 format.  Returns dict(rc, tau, J, next_ctr[, G]).
    Row-sharded context: m = this rank's rows, m_global = the matrix's (tau has min(m_global, n) entries)."""
    torch = _torch()
    dev = f"cuda:{ctx.device}"
    tau = torch.zeros(min(m_global or m, n), dtype=A.dtype, device=dev)
    J = torch.zeros(n, dtype=torch.int64, device=dev)
    G = cm_empty(nb_alg + pp, m, dtype=A.dtype, device=dev) if want_G else None
    st = _state_arr(ctr, key)
    rc = getattr(ctx.lib, f"rlhip_drv_hqrrp_{_suffix(A)[0]}")(ctx.h, m, n, A.data_ptr(), m, J.data_ptr(), tau.data_ptr(), nb_alg, pp, panel_pivoting, qr_type,
                                     st, G.data_ptr() if G is not None else None)
    _drv_check(ctx, rc, "hqrrp")
    out = dict(rc=rc, tau=tau, J=J, next_ctr=tuple(int(x) for x in st[:4]))
    if want_G:
        out["G"] = G
    return out


def drv_hqrrp_timed(ctx: Context, A, m, n, nb_alg=64, pp=10, panel_pivoting=0, qr_type=0, ctr=(0, 0, 0, 0), key=(0, 0)):
    """hqrrp with the reference's timing argument armed: dict(rc, tau, J, times_us (27 entries, rl_hqrrp.hh:1144-1164))"""
    torch = _torch()
    dev = f"cuda:{ctx.device}"
    tau = torch.zeros(min(m, n), dtype=A.dtype, device=dev)
    J = torch.zeros(n, dtype=torch.int64, device=dev)
    st = _state_arr(ctr, key)
    times = (C.c_double * 27)()
    rc = ctx.lib.rlhip_drv_hqrrp_timed_f64(ctx.h, m, n, A.data_ptr(), m, J.data_ptr(), tau.data_ptr(), nb_alg, pp, panel_pivoting, qr_type, st, times)
    _drv_check(ctx, rc, "hqrrp_timed")
    return dict(rc=rc, tau=tau, J=J, times_us=[float(x) for x in times])


def drv_cqrrpt(ctx: Context, A, m, n, d_factor=1.25, nnz=4, eps=None, ctr=(0, 0, 0, 0), key=(0, 0), sketch_in=None,
               want_sketch=False, timing=False, qrcp=-1):
    """CQRRPT::call (qrcp = geqp3).  A (column-major tensor (n, m)) is overwritten by Q.  Returns dict(rc, rank, R, J,
    next_ctr[, sketch][, times_us])."""
    torch = _torch()
    dev = f"cuda:{ctx.device}"
    npdt = np.float64 if A.dtype == torch.float64 else np.float32
    if eps is None:
        eps = float(np.finfo(npdt).eps ** 0.85)
    d = int(d_factor * n)
    R = cm_zeros(n, n, dtype=A.dtype, device=dev)
    J = torch.zeros(n, dtype=torch.int64, device=dev)
    sk_out = cm_empty(d, n, dtype=A.dtype, device=dev) if want_sketch else None
    rank = C.c_int64(0)
    st = _state_arr(ctr, key)
    times = (C.c_long * 8)() if timing else None
    rc = getattr(ctx.lib, f"rlhip_drv_cqrrpt_{_suffix(A)[0]}")(ctx.h, m, n, A.data_ptr(), m, R.data_ptr(), n, J.data_ptr(), d_factor, nnz, eps, st,
                                      sketch_in.data_ptr() if sketch_in is not None else None,
                                      sk_out.data_ptr() if sk_out is not None else None, C.byref(rank), times, qrcp)
    _drv_check(ctx, rc, "cqrrpt")
    out = dict(rc=rc, rank=int(rank.value), R=R, J=J, next_ctr=tuple(int(x) for x in st[:4]))
    if want_sketch:
        out["sketch"] = sk_out
    if timing:
        out["times_us"] = [int(t) for t in times]
    return out


def drv_bqrrp(ctx: Context, A, m, n, b_sz, d_factor=1.0, internal_nb=0, tol=0.0, ctr=(0, 0, 0, 0), key=(0, 0), sketch_in=None,
              want_sketch=False, timing=False, qrcp_wide=-1, qr_tall=-1, apply_trans_q=-1, m_global=None, block_cyclic=False):
    """BQRRP::call; options as in the reference's enums (qrcp_wide 0 luqr | 1 geqp3; qr_tall 0 geqrt | 1 cholqr | 2 geqrf;
    apply_trans_q 0 ormqr | 1 gemqrt; -1 = object default).  A (column-major tensor (n, m)) is overwritten in GEQP3 format.
    Returns dict(rc, rank, tau, J, next_ctr[, sketch][, times_us])."""
    torch = _torch()
    dev = f"cuda:{ctx.device}"
    d = int(d_factor * b_sz)
    # row-sharded call (context joined to a communicator): m is the LOCAL row count, tau has min(global rows, n) entries
    tau = torch.zeros(min(m_global or m, n), dtype=A.dtype, device=dev)
    J = torch.zeros(n, dtype=torch.int64, device=dev)
    sk_out = cm_empty(d, n, dtype=A.dtype, device=dev) if want_sketch else None
    rank = C.c_int64(0)
    st = _state_arr(ctr, key)
    times = (C.c_long * 9)() if timing else None
    rc = getattr(ctx.lib, f"rlhip_drv_bqrrp_{_suffix(A)[0]}")(ctx.h, m, n, A.data_ptr(), m, d_factor, b_sz, internal_nb, tol, tau.data_ptr(), J.data_ptr(),
                                     st, sketch_in.data_ptr() if sketch_in is not None else None,
                                     sk_out.data_ptr() if sk_out is not None else None, C.byref(rank), times,
                                     qrcp_wide, (16 + (qr_tall if qr_tall >= 0 else 3)) if block_cyclic else qr_tall, apply_trans_q)
    _drv_check(ctx, rc, "bqrrp")
    out = dict(rc=rc, rank=int(rank.value), tau=tau, J=J, next_ctr=tuple(int(x) for x in st[:4]))
    if want_sketch:
        out["sketch"] = sk_out
    if timing:
        out["times_us"] = [int(t) for t in times]
    return out


def drv_bqrrp_gpu(ctx: Context, A, m, n, A_sk, d, b_sz, qr_tall=-1, tol=0.0, timing=False, lda=None):
    """BQRRP_GPU::call (the reference's device class, drivers/rl_bqrrp_gpu.hh:120-129): the d x n sketch A_sk (column-major tensor
    (n, d), OVERWRITTEN) is an input.  qr_tall 0 cholqr | 1 geqrf | -1 object default (geqrf).  A is overwritten in GEQP3 format.
    Returns dict(rc, rank, tau, J[, times_us (15 entries)])."""
    torch = _torch()
    dev = f"cuda:{ctx.device}"
    tau = torch.zeros(n, dtype=A.dtype, device=dev)
    J = torch.zeros(n, dtype=torch.int64, device=dev)
    rank = C.c_int64(0)
    times = (C.c_long * 15)() if timing else None
    rc = getattr(ctx.lib, f"rlhip_drv_bqrrp_gpu_{_suffix(A)[0]}")(ctx.h, m, n, A.data_ptr(), lda or m, A_sk.data_ptr(), d, b_sz, qr_tall, tol,
                                                                   tau.data_ptr(), J.data_ptr(), C.byref(rank), times)
    _drv_check(ctx, rc, "bqrrp_gpu")
    out = dict(rc=rc, rank=int(rank.value), tau=tau, J=J)
    if timing:
        out["times_us"] = [int(t) for t in times]
    return out


def drv_cqrrpt_gpu(ctx: Context, A_host, d_factor=1.25, nnz=4, eps=None, ctr=(0, 0, 0, 0), key=(0, 0), no_hqrrp=-1, want_sketch=False,
                   timing=False, lda=None, ldr=None):
    """CQRRPT_GPU::call (drivers/rl_cqrrpt_gpu.hh:117-127): HOST matrices, as in the reference.  A_host: numpy (m, n); it is NOT
    modified -- the call runs on a column-major copy with leading dimension lda (default m).  Returns dict(rc, rank, Q (m x n numpy),
    R (n x n numpy), J, next_ctr[, sketch (d x n numpy)][, times_us], A_buf / R_buf = the raw padded buffers)."""
    m, n = A_host.shape
    npdt = A_host.dtype.type
    lda = lda or m
    ldr = ldr or n
    if eps is None:
        eps = float(np.finfo(npdt).eps ** 0.85)
    d = int(npdt(d_factor) * n)
    A_buf = np.full(lda * n, np.nan, dtype=npdt)
    A_buf.reshape(n, lda)[:, :m] = A_host.T
    R_buf = np.full(ldr * n, 7.0, dtype=npdt)                       # padding rows keep their marker (checked by the tests)
    R_buf.reshape(n, ldr)[:, :n] = 0.0
    J = np.zeros(n, dtype=np.int64)
    sk = np.zeros(d * n, dtype=npdt) if want_sketch else None
    rank = C.c_int64(0)
    st = _state_arr(ctr, key)
    times = (C.c_long * 8)() if timing else None
    suffix = "f64" if npdt is np.float64 else "f32"
    rc = getattr(ctx.lib, f"rlhip_drv_cqrrpt_gpu_{suffix}")(ctx.h, m, n, A_buf.ctypes.data, lda, R_buf.ctypes.data, ldr, J.ctypes.data, d_factor, nnz, eps,
                                                           no_hqrrp, st, sk.ctypes.data if sk is not None else None, C.byref(rank), times)
    _drv_check(ctx, rc, "cqrrpt_gpu")
    out = dict(rc=rc, rank=int(rank.value), Q=A_buf.reshape(n, lda)[:, :m].T.copy(), R=R_buf.reshape(n, ldr)[:, :n].T.copy(), J=J,
               next_ctr=tuple(int(x) for x in st[:4]), A_buf=A_buf, R_buf=R_buf)
    if want_sketch:
        out["sketch"] = sk.reshape(n, d).T.copy()
    if timing:
        out["times_us"] = [int(t) for t in times]
    return out
