// lapack:: call surface of the hot path, device flavour (cf. RandLAPACK/rl_lapackpp.hh:5-9).
#pragma once
#include <cmath>
#include "rl_blaspp.hh"

namespace lapack {

using blas::Queue;
enum class Job : char { NoVec = 'N', Vec = 'V', SomeVec = 'S', AllVec = 'A' };
enum class Norm : char { One = '1', Inf = 'I', Fro = 'F', Max = 'M' };
enum class MatrixType : char { General = 'G', Lower = 'L', Upper = 'U' };
inline char to_char(Job v) { return (char)v; }
inline char to_char(Norm v) { return (char)v; }
inline char to_char(MatrixType v) { return (char)v; }

// returns LAPACK info (0 or the order of the failing minor), like lapack::potrf
inline int64_t potrf(blas::Uplo u, int64_t n, double* A, int64_t lda, Queue& q = blas::default_queue()) {
    int rc = rlhip_potrf_f64(q.ctx(), (char)u, n, A, lda); blas::check(rc, "potrf"); return rc;
}
inline int64_t potrf(blas::Uplo u, int64_t n, float* A, int64_t lda, Queue& q = blas::default_queue()) {
    int rc = rlhip_potrf_f32(q.ctx(), (char)u, n, A, lda); blas::check(rc, "potrf"); return rc;
}
// CholQRQ's syrk + potrf + trsm as one call (rlhip_cholqrq): 0 done (info = potrf's), 1 shape not served
inline int cholqrq(int64_t m, int64_t k, double* A, int64_t lda, double* R, bool reduce_gram, int& info, Queue& q = blas::default_queue()) {
    const int rc = rlhip_cholqrq_f64(q.ctx(), m, k, A, lda, R, reduce_gram ? 1 : 0, &info);
    blas::check(rc, "cholqrq");
    return rc;
}
inline int cholqrq(int64_t m, int64_t k, float* A, int64_t lda, float* R, bool reduce_gram, int& info, Queue& q = blas::default_queue()) {
    const int rc = rlhip_cholqrq_f32(q.ctx(), m, k, A, lda, R, reduce_gram ? 1 : 0, &info);
    blas::check(rc, "cholqrq");
    return rc;
}
inline double lange(Norm nt, int64_t m, int64_t n, double const* A, int64_t lda, Queue& q = blas::default_queue()) {
    if (nt != Norm::Fro) throw blas::Error("lange: only Norm::Fro is on the path");
    double r = 0; blas::check(rlhip_lange_fro_f64(q.ctx(), m, n, A, lda, &r), "lange"); return r;
}
inline float lange(Norm nt, int64_t m, int64_t n, float const* A, int64_t lda, Queue& q = blas::default_queue()) {
    if (nt != Norm::Fro) throw blas::Error("lange: only Norm::Fro is on the path");
    float r = 0; blas::check(rlhip_lange_fro_f32(q.ctx(), m, n, A, lda, &r), "lange"); return r;
}
inline void lacpy(MatrixType t, int64_t m, int64_t n, double const* A, int64_t lda, double* B, int64_t ldb, Queue& q = blas::default_queue()) {
    blas::check(rlhip_lacpy_f64(q.ctx(), (char)t, m, n, A, lda, B, ldb), "lacpy");
}
inline void lacpy(MatrixType t, int64_t m, int64_t n, float const* A, int64_t lda, float* B, int64_t ldb, Queue& q = blas::default_queue()) {
    blas::check(rlhip_lacpy_f32(q.ctx(), (char)t, m, n, A, lda, B, ldb), "lacpy");
}
inline void laset(MatrixType t, int64_t m, int64_t n, double offd, double diag, double* A, int64_t lda, Queue& q = blas::default_queue()) {
    blas::check(rlhip_laset_f64(q.ctx(), (char)t, m, n, offd, diag, A, lda), "laset");
}
inline void laset(MatrixType t, int64_t m, int64_t n, float offd, float diag, float* A, int64_t lda, Queue& q = blas::default_queue()) {
    blas::check(rlhip_laset_f32(q.ctx(), (char)t, m, n, offd, diag, A, lda), "laset");
}
inline void add_diag(int64_t n, double alpha, double* A, int64_t lda, Queue& q = blas::default_queue()) {
    blas::check(rlhip_add_diag_f64(q.ctx(), n, alpha, A, lda), "add_diag");
}
inline void add_diag(int64_t n, float alpha, float* A, int64_t lda, Queue& q = blas::default_queue()) {
    blas::check(rlhip_add_diag_f32(q.ctx(), n, alpha, A, lda), "add_diag");
}
// column-pivoted QR of a device matrix; jpvt (device int64, 1-based on exit), tau (device)
inline int64_t geqp3(int64_t m, int64_t n, double* A, int64_t lda, int64_t* jpvt, double* tau, Queue& q = blas::default_queue()) {
    int rc = rlhip_geqp3_f64(q.ctx(), m, n, A, lda, jpvt, tau); blas::check(rc, "geqp3"); return rc;
}
inline int64_t geqp3(int64_t m, int64_t n, float* A, int64_t lda, int64_t* jpvt, float* tau, Queue& q = blas::default_queue()) {
    int rc = rlhip_geqp3_f32(q.ctx(), m, n, A, lda, jpvt, tau); blas::check(rc, "geqp3"); return rc;
}
inline void get_diag(int64_t n, double const* A, int64_t lda, double* diag_host, Queue& q = blas::default_queue()) {
    blas::check(rlhip_get_diag_f64(q.ctx(), n, A, lda, diag_host), "get_diag");
}
inline void get_diag(int64_t n, float const* A, int64_t lda, float* diag_host, Queue& q = blas::default_queue()) {
    blas::check(rlhip_get_diag_f32(q.ctx(), n, A, lda, diag_host), "get_diag");
}
inline bool any_abs_gt(int64_t n, double const* x, double thr, Queue& q = blas::default_queue()) { int a = 0; blas::check(rlhip_any_abs_gt_f64(q.ctx(), n, x, thr, &a), "any_abs_gt"); return a != 0; }
inline bool any_abs_gt(int64_t n, float const* x, float thr, Queue& q = blas::default_queue()) { int a = 0; blas::check(rlhip_any_abs_gt_f32(q.ctx(), n, x, thr, &a), "any_abs_gt"); return a != 0; }
inline void orhr_col(int64_t m, int64_t n, int64_t nb, double* A, int64_t lda, double* T, int64_t ldt, double* D, Queue& q = blas::default_queue()) { blas::check(rlhip_orhr_col_f64(q.ctx(), m, n, nb, A, lda, T, ldt, D), "orhr_col"); }
inline void orhr_col(int64_t m, int64_t n, int64_t nb, float* A, int64_t lda, float* T, int64_t ldt, float* D, Queue& q = blas::default_queue()) { blas::check(rlhip_orhr_col_f32(q.ctx(), m, n, nb, A, lda, T, ldt, D), "orhr_col"); }
inline void gemqrt(blas::Side s, blas::Op t, int64_t m, int64_t n, int64_t k, int64_t nb, double const* V, int64_t ldv, double const* T, int64_t ldt, double* C, int64_t ldc, Queue& q = blas::default_queue()) { blas::check(rlhip_gemqrt_f64(q.ctx(), (char)s, (char)t, m, n, k, nb, V, ldv, T, ldt, C, ldc), "gemqrt"); }
inline void gemqrt(blas::Side s, blas::Op t, int64_t m, int64_t n, int64_t k, int64_t nb, float const* V, int64_t ldv, float const* T, int64_t ldt, float* C, int64_t ldc, Queue& q = blas::default_queue()) { blas::check(rlhip_gemqrt_f32(q.ctx(), (char)s, (char)t, m, n, k, nb, V, ldv, T, ldt, C, ldc), "gemqrt"); }
// gemqrt(Left, Trans) of one compact-WY block in two calls (rlhip_gemqrt_head / _tail): after `head` the first k rows of C are final
inline void gemqrt_head(int64_t m, int64_t n, int64_t k, double const* V, int64_t ldv, double const* T, int64_t ldt, double* C, int64_t ldc, double* W2, Queue& q = blas::default_queue()) { blas::check(rlhip_gemqrt_head_f64(q.ctx(), m, n, k, V, ldv, T, ldt, C, ldc, W2), "gemqrt_head"); }
inline void gemqrt_head(int64_t m, int64_t n, int64_t k, float const* V, int64_t ldv, float const* T, int64_t ldt, float* C, int64_t ldc, float* W2, Queue& q = blas::default_queue()) { blas::check(rlhip_gemqrt_head_f32(q.ctx(), m, n, k, V, ldv, T, ldt, C, ldc, W2), "gemqrt_head"); }
inline void gemqrt_tail(int64_t m, int64_t n, int64_t k, double const* V, int64_t ldv, double const* W2, double* C, int64_t ldc, Queue& q = blas::default_queue()) { blas::check(rlhip_gemqrt_tail_f64(q.ctx(), m, n, k, V, ldv, W2, C, ldc), "gemqrt_tail"); }
inline void gemqrt_tail(int64_t m, int64_t n, int64_t k, float const* V, int64_t ldv, float const* W2, float* C, int64_t ldc, Queue& q = blas::default_queue()) { blas::check(rlhip_gemqrt_tail_f32(q.ctx(), m, n, k, V, ldv, W2, C, ldc), "gemqrt_tail"); }
inline void larft(int64_t m, int64_t k, double const* V, int64_t ldv, double const* tau, double* T, int64_t ldt, Queue& q = blas::default_queue()) { blas::check(rlhip_larft_f64(q.ctx(), m, k, V, ldv, tau, T, ldt), "larft"); }
inline void larft(int64_t m, int64_t k, float const* V, int64_t ldv, float const* tau, float* T, int64_t ldt, Queue& q = blas::default_queue()) { blas::check(rlhip_larft_f32(q.ctx(), m, k, V, ldv, tau, T, ldt), "larft"); }
inline void row_sign(int64_t n, double* R, int64_t ldr, double const* D, Queue& q = blas::default_queue()) { blas::check(rlhip_row_sign_f64(q.ctx(), n, R, ldr, D), "row_sign"); }
inline void row_sign(int64_t n, float* R, int64_t ldr, float const* D, Queue& q = blas::default_queue()) { blas::check(rlhip_row_sign_f32(q.ctx(), n, R, ldr, D), "row_sign"); }
inline void tau_from_t(int64_t k, int64_t nb, double const* T, int64_t ldt, double* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_tau_from_t_f64(q.ctx(), k, nb, T, ldt, tau), "tau_from_t"); }
inline void tau_from_t(int64_t k, int64_t nb, float const* T, int64_t ldt, float* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_tau_from_t_f32(q.ctx(), k, nb, T, ldt, tau), "tau_from_t"); }
// Householder QR / LU building blocks (device tau / ipiv)
inline int64_t geqrf(int64_t m, int64_t n, double* A, int64_t lda, double* tau, Queue& q = blas::default_queue()) { int rc = rlhip_geqrf_f64(q.ctx(), m, n, A, lda, tau); blas::check(rc, "geqrf"); return rc; }
inline int64_t geqrf(int64_t m, int64_t n, float* A, int64_t lda, float* tau, Queue& q = blas::default_queue()) { int rc = rlhip_geqrf_f32(q.ctx(), m, n, A, lda, tau); blas::check(rc, "geqrf"); return rc; }
// geqrf + ungqr(m, n, n) in one call (rlhip_geqrf_q): true when done (A = Q, R = the n x n triangle, zero below), false when not taken -- the caller runs the two calls
inline bool geqrf_q(int64_t m, int64_t n, double* A, int64_t lda, double* R, int64_t ldr, Queue& q = blas::default_queue()) { int rc = rlhip_geqrf_q_f64(q.ctx(), m, n, A, lda, R, ldr); blas::check(rc, "geqrf_q"); return rc == 0; }
inline bool geqrf_q(int64_t m, int64_t n, float* A, int64_t lda, float* R, int64_t ldr, Queue& q = blas::default_queue()) { int rc = rlhip_geqrf_q_f32(q.ctx(), m, n, A, lda, R, ldr); blas::check(rc, "geqrf_q"); return rc == 0; }
inline void ungqr(int64_t m, int64_t n, int64_t k, double* A, int64_t lda, double const* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_ungqr_f64(q.ctx(), m, n, k, A, lda, tau), "ungqr"); }
inline void ungqr(int64_t m, int64_t n, int64_t k, float* A, int64_t lda, float const* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_ungqr_f32(q.ctx(), m, n, k, A, lda, tau), "ungqr"); }
// returns info (> 0: exactly singular U, factorization still complete)
inline int64_t getrf(int64_t m, int64_t n, double* A, int64_t lda, int64_t* ipiv, Queue& q = blas::default_queue()) { int rc = rlhip_getrf_f64(q.ctx(), m, n, A, lda, ipiv); blas::check(rc, "getrf"); return rc; }
inline int64_t getrf(int64_t m, int64_t n, float* A, int64_t lda, int64_t* ipiv, Queue& q = blas::default_queue()) { int rc = rlhip_getrf_f32(q.ctx(), m, n, A, lda, ipiv); blas::check(rc, "getrf"); return rc; }
/// getrf whose caller only reads ipiv (same pivots, factors left as scratch)
inline int64_t getrf_pivots(int64_t m, int64_t n, double* A, int64_t lda, int64_t* ipiv, Queue& q = blas::default_queue()) { int rc = rlhip_getrf_piv_f64(q.ctx(), m, n, A, lda, ipiv); blas::check(rc, "getrf_piv"); return rc; }
inline int64_t getrf_pivots(int64_t m, int64_t n, float* A, int64_t lda, int64_t* ipiv, Queue& q = blas::default_queue()) { int rc = rlhip_getrf_piv_f32(q.ctx(), m, n, A, lda, ipiv); blas::check(rc, "getrf_piv"); return rc; }
inline void laswp(int64_t n, double* A, int64_t lda, int64_t k1, int64_t k2, int64_t const* ipiv, int64_t incx, Queue& q = blas::default_queue()) {
    if (incx != 1) throw blas::Error("laswp: only incx = 1 is on the path");
    blas::check(rlhip_laswp_f64(q.ctx(), n, A, lda, k1, k2, ipiv), "laswp");
}
inline void laswp(int64_t n, float* A, int64_t lda, int64_t k1, int64_t k2, int64_t const* ipiv, int64_t incx, Queue& q = blas::default_queue()) {
    if (incx != 1) throw blas::Error("laswp: only incx = 1 is on the path");
    blas::check(rlhip_laswp_f32(q.ctx(), n, A, lda, k1, k2, ipiv), "laswp");
}
// lapack::geqrt(m, n, nb, A, lda, T, ldt): geqrf + one compact-WY T per nb-wide block (T is nb x n); tau is scratch (device, n)
template <typename T>
inline void geqrt(int64_t m, int64_t n, int64_t nb, T* A, int64_t lda, T* Tm, int64_t ldt, T* tau_scratch, Queue& q = blas::default_queue()) {
    geqrf(m, n, A, lda, tau_scratch, q);
    for (int64_t i = 0; i < n; i += nb) {
        const int64_t ib = std::min(nb, n - i);
        larft(m - i, ib, A + i + i * lda, lda, tau_scratch + i, Tm + i * ldt, ldt, q);
    }
}
// lapack::ormqr(Side::Left, Op::Trans, m, n, k, V, ldv, tau, C, ldc): one k x k compact-WY block (T_scratch: k*k device)
template <typename T>
inline void ormqr(blas::Side s, blas::Op t, int64_t m, int64_t n, int64_t k, T const* V, int64_t ldv, T const* tau, T* C, int64_t ldc, T* T_scratch, Queue& q = blas::default_queue()) {
    if (k == 0 || n == 0) return;
    larft(m, k, V, ldv, tau, T_scratch, k, q);
    gemqrt(s, t, m, n, k, k, V, ldv, T_scratch, k, C, ldc, q);
}
// lapack::lansy(Norm::Fro, uplo, n, A, lda) (test/comps/test_qb.cc:168): Frobenius norm of the symmetric matrix whose `uplo` triangle
// is stored: ||A||_F^2 = 2 ||triangle||_F^2 - ||diag||^2, the triangle's norm on the device
template <typename T>
inline T lansy(Norm nt, blas::Uplo u, int64_t n, T const* A, int64_t lda, Queue& q = blas::default_queue()) {
    if (nt != Norm::Fro) throw blas::Error("lansy: only Norm::Fro is on the path");
    if (n <= 0) return (T)0;
    blas::Scratch ws(q);
    T* W = ws.alloc<T>(n * n);
    laset(MatrixType::General, n, n, (T)0, (T)0, W, n, q);
    lacpy(u == blas::Uplo::Upper ? MatrixType::Upper : MatrixType::Lower, n, n, A, lda, W, n, q);
    const T tri = lange(Norm::Fro, n, n, W, n, q);
    std::vector<T> dg((size_t)n);
    get_diag(n, A, lda, dg.data(), q);
    double dd = 0;
    for (T v : dg) dd += (double)v * (double)v;
    const double s2 = 2.0 * (double)tri * (double)tri - dd;
    return (T)std::sqrt(s2 > 0 ? s2 : 0.0);
}
inline void qrp_partial(int64_t m, int64_t n, int64_t steps, double* A, int64_t lda, int64_t* jpvt, double* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_qrp_partial_f64(q.ctx(), m, n, steps, A, lda, jpvt, tau), "qrp_partial"); }
inline void qrp_partial(int64_t m, int64_t n, int64_t steps, float* A, int64_t lda, int64_t* jpvt, float* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_qrp_partial_f32(q.ctx(), m, n, steps, A, lda, jpvt, tau), "qrp_partial"); }
// the first `steps` steps of geqp3 (see rlhip_geqp3_steps_f64)
inline void geqp3_steps(int64_t m, int64_t n, int64_t steps, double* A, int64_t lda, int64_t* jpvt, double* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_geqp3_steps_f64(q.ctx(), m, n, steps, A, lda, jpvt, tau), "geqp3_steps"); }
inline void geqp3_steps(int64_t m, int64_t n, int64_t steps, float* A, int64_t lda, int64_t* jpvt, float* tau, Queue& q = blas::default_queue()) { blas::check(rlhip_geqp3_steps_f32(q.ctx(), m, n, steps, A, lda, jpvt, tau), "geqp3_steps"); }
inline void vrows_explicit(int64_t br, int64_t toff, int64_t tcnt, double const* Vtop, int64_t ldv, double* out, int64_t ldo, Queue& q = blas::default_queue()) { blas::check(rlhip_vrows_explicit_f64(q.ctx(), br, toff, tcnt, Vtop, ldv, out, ldo), "vrows_explicit"); }
inline void vrows_explicit(int64_t br, int64_t toff, int64_t tcnt, float const* Vtop, int64_t ldv, float* out, int64_t ldo, Queue& q = blas::default_queue()) { blas::check(rlhip_vrows_explicit_f32(q.ctx(), br, toff, tcnt, Vtop, ldv, out, ldo), "vrows_explicit"); }
inline void luqrcp_piv(int64_t sd, int64_t cols, int64_t const* ipiv, int64_t* J, Queue& q = blas::default_queue()) { blas::check(rlhip_luqrcp_piv(q.ctx(), sd, cols, ipiv, J), "luqrcp_piv"); }
// Job::SomeVec, tall (m >= n).  Returns info (>0: Jacobi did not converge).
inline int64_t gesdd(Job job, int64_t m, int64_t n, double* A, int64_t lda, double* S, double* U, int64_t ldu,
                     double* VT, int64_t ldvt, Queue& q = blas::default_queue()) {
    if (job != Job::SomeVec) throw blas::Error("gesdd: only Job::SomeVec is on the path");
    int rc = rlhip_gesdd_f64(q.ctx(), m, n, A, lda, S, U, ldu, VT, ldvt, nullptr); blas::check(rc, "gesdd"); return rc;
}
inline int64_t gesdd(Job job, int64_t m, int64_t n, float* A, int64_t lda, float* S, float* U, int64_t ldu,
                     float* VT, int64_t ldvt, Queue& q = blas::default_queue()) {
    if (job != Job::SomeVec) throw blas::Error("gesdd: only Job::SomeVec is on the path");
    int rc = rlhip_gesdd_f32(q.ctx(), m, n, A, lda, S, U, ldu, VT, ldvt, nullptr); blas::check(rc, "gesdd"); return rc;
}
// singular values only of a (small) matrix copy -- used by util::cond_num_check
inline int64_t gesvdj(int64_t m, int64_t n, double* A, int64_t lda, double* S, double* VT, int64_t ldvt, Queue& q = blas::default_queue()) {
    int rc = rlhip_gesvdj_f64(q.ctx(), m, n, A, lda, S, VT, ldvt, nullptr); blas::check(rc, "gesvdj"); return rc;
}
inline int64_t gesvdj(int64_t m, int64_t n, float* A, int64_t lda, float* S, float* VT, int64_t ldvt, Queue& q = blas::default_queue()) {
    int rc = rlhip_gesvdj_f32(q.ctx(), m, n, A, lda, S, VT, ldvt, nullptr); blas::check(rc, "gesvdj"); return rc;
}

}  // namespace lapack

namespace RandLAPACK {
using lapack::Job;
using lapack::MatrixType;
using lapack::Norm;
}  // namespace RandLAPACK
