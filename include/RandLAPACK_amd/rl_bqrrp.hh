// BQRRPalg / BQRRP (reference: RandLAPACK/drivers/rl_bqrrp.hh:19-665; device twin rl_bqrrp_gpu.hh): blocked QR with
// randomized pivoting, GEQP3-compatible output (R above the diagonal, Householder vectors below, tau, 1-based J).
//
// Device status of the sub-routine options (BQRRPSubroutines):
//   qrcp_wide     geqp3  -> persistent device geqp3 (qrcp.hip)              | luqr  -> device getrf (lu.hip) + wide geqrf
//   qr_tall       cholqr -> trsm/syrk/potrf/trsm + orhr_col (house.hip)     | geqrf, geqrt -> step-synchronous Householder kernel
//                                                                              (correct, BLAS-2 bound: cholqr is the fast one)
//   apply_trans_q gemqrt -> compact-WY block apply on the MFMA GEMMs        | ormqr -> the same apply (tau is the
//                                                                             diagonal of T, :490-491, so both name one operator)
// Every option of the reference is available.  The object DEFAULTS to {luqr, cholqr, gemqrt}: luqr is the reference's own
// default, cholqr/gemqrt are the BLAS-3 choices (the reference's CPU defaults geqrf/ormqr are supported, just slower here).
#pragma once
#include <chrono>
#include <cmath>
#include <limits>
#include <vector>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"
#include "rl_util.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class BQRRPalg {
public:
    virtual ~BQRRPalg() {}
    virtual int call(int64_t m, int64_t n, T* A, int64_t lda, T d_factor, T* tau, int64_t* J,
                     RandBLAS::RNGState<RNG>& state) = 0;
};

struct BQRRPSubroutines {
    enum QRCPWide { luqr, geqp3 };
    enum QRTall { geqrt, cholqr, geqrf };
    enum ApplyTransQ { ormqr, gemqrt };
};

template <typename T, typename RNG>
class BQRRP : public BQRRPalg<T, RNG> {
public:
    using Subroutines = BQRRPSubroutines;

    BQRRP(blas::Queue& queue, bool time_subroutines, int64_t b_sz) : q(queue) {
        randlapack_require(b_sz > 0) << "BQRRP block size b_sz=" << b_sz << " must be > 0";
        timing = time_subroutines;
        tol = std::numeric_limits<T>::epsilon();
        block_size = b_sz;
        internal_nb = b_sz;
        qrcp_wide = Subroutines::QRCPWide::luqr;       // the reference's default (rl_bqrrp.hh:74) and the faster one here
        qr_tall = Subroutines::QRTall::cholqr;         // reference CPU default: geqrf (:75); cholqr is the BLAS-3 panel on the device
        apply_trans_q = Subroutines::ApplyTransQ::gemqrt;
        rank = 0;
    }

    /// A (m x n, lda), tau (min(m,n)), J (n, int64): DEVICE buffers.  Returns 0.  `rank` as in the reference (an upper
    /// bound on the numerical rank).  SURVEY.md A.8 is the behavioural spec; line numbers refer to rl_bqrrp.hh.
    int call(int64_t m, int64_t n, T* A, int64_t lda, T d_factor, T* tau, int64_t* J, RandBLAS::RNGState<RNG>& state) override {
        randlapack_require(m >= 0 && n >= 0 && lda >= m) << "bad dimensions";
        const int64_t mn = std::min(m, n);
        if (mn == 0) { rank = 0; return 0; }
        using clk = std::chrono::steady_clock;
        auto stamp = [&]() { if (timing) q.sync(); return clk::now(); };
        auto us = [](clk::time_point a, clk::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        long t_skop = 0, t_qrcp = 0, t_pre = 0, t_tall = 0, t_rec = 0, t_apply = 0, t_upd = 0;
        auto t_begin = stamp();

        int64_t rows = m, cols = n, curr_sz = 0, b_sz = block_size;
        const int64_t maxiter = (int64_t)std::ceil(mn / (T)b_sz);                                         // :220
        const int64_t b_sz_const = b_sz;
        const int64_t d = (int64_t)(d_factor * b_sz);                                                       // :224
        int64_t sampling_dimension = d, block_rank = b_sz, inb = internal_nb;
        T* A_work = A;

        blas::Scratch ws(q);
        int64_t* J_buffer = ws.alloc<int64_t>(n);
        T* A_sk_base = ws.alloc<T>(d * n);
        T* R_tall_qr = ws.alloc<T>(b_sz_const * b_sz_const);
        T* T_dat = ws.alloc<T>(b_sz_const * b_sz_const);
        T* Work2 = ws.alloc<T>(n);
        const bool lu = (qrcp_wide == Subroutines::QRCPWide::luqr);
        T* T_ormqr = ws.alloc<T>(b_sz_const * b_sz_const);
        T* A_sk_trans = lu ? ws.alloc<T>(n * d) : nullptr;                                                   // :262-266
        int64_t* J_buffer_lu = lu ? ws.alloc<int64_t>(std::min(d, n)) : nullptr;
        T* A_sk = A_sk_base;

        auto t0 = stamp();
        if (sketch_override) {
            lapack::lacpy(MatrixType::General, d, n, sketch_override, d, A_sk, d, q);
        } else {                                                                                            // :309-313
            T* S = blas::device_malloc<T>(d * m, q);
            RandBLAS::DenseDist D(d, m);
            state = RandBLAS::fill_dense(D, S, state, q);
            blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, d, n, m, (T)1.0, S, d, A, lda, (T)0.0, A_sk, d, q);   // lda, not m (SURVEY B)
            blas::device_free(S, q);
        }
        if (sketch_export) lapack::lacpy(MatrixType::General, d, n, A_sk, d, sketch_export, d, q);
        t_skop = us(t0, stamp());
        std::vector<T> diag(b_sz_const);

        for (int64_t iter = 0; iter < maxiter; ++iter) {
            b_sz = std::min(b_sz, mn - curr_sz);                                                            // :322-324
            inb = std::min(inb, b_sz);
            block_rank = b_sz;
            auto ta = stamp();
            if (!lu) {
                lapack::geqp3(sampling_dimension, cols, A_sk, d, J_buffer, Work2, q);                       // :336
            } else {                                                                                        // :337-357
                blas::check(transpose_call(sampling_dimension, cols, A_sk, d, A_sk_trans, n), "transposition");
                lapack::getrf(cols, sampling_dimension, A_sk_trans, n, J_buffer_lu, q);
                lapack::luqrcp_piv(sampling_dimension, cols, J_buffer_lu, J_buffer, q);
                util::col_swap(sampling_dimension, cols, cols, A_sk, d, J_buffer, q);
                lapack::geqrf(sampling_dimension, cols, A_sk, d, Work2, q);
            }
            t_qrcp += us(ta, stamp());
            ta = stamp();
            util::col_swap(m, cols, cols, &A[lda * curr_sz], lda, J_buffer, q);                             // :369
            bool block_zero = !lapack::any_abs_gt(rows, A_work, std::numeric_limits<T>::epsilon(), q);     // :373-379
            if (iter == 0) blas::device_copy_vector(cols, J_buffer, J, q);                                  // :383-387 / :402-406
            else util::col_swap(cols, cols, &J[curr_sz], J_buffer, q);
            if (block_zero) { rank = curr_sz; finish(t_begin, t_skop, t_qrcp, t_pre, t_tall, t_rec, t_apply, t_upd); return 0; }   // :380-399
            T* Work1 = &A_work[lda * b_sz];
            T* R_sk = A_sk;
            lapack::get_diag(b_sz, R_sk, d, diag.data(), q);
            for (int64_t i = 0; i < b_sz; ++i) {                                                            // :421-427
                if (std::abs(diag[i]) / std::abs(diag[0]) < tol) { block_rank = i; inb = std::min(inb, block_rank); break; }
            }
            T* tau_sub = &tau[curr_sz];
            T* R11 = A_work;
            bool have_T = true;
            if (qr_tall == Subroutines::QRTall::cholqr) {                                                   // :454-505
                blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, rows, block_rank, (T)1.0, R_sk, d, A_work, lda, q);
                t_pre += us(ta, stamp());
                ta = stamp();
                lapack::laset(MatrixType::General, b_sz_const, b_sz_const, (T)0, (T)0, R_tall_qr, b_sz_const, q);
                blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, block_rank, rows, (T)1.0, A_work, lda, (T)0.0, R_tall_qr, b_sz_const, q);
                lapack::potrf(Uplo::Upper, block_rank, R_tall_qr, b_sz_const, q);   // failure handled "gracefully" as in the reference (:461)
                blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, rows, block_rank, (T)1.0, R_tall_qr, b_sz_const, A_work, lda, q);
                t_tall += us(ta, stamp());
                ta = stamp();
                lapack::orhr_col(rows, block_rank, inb, A_work, lda, T_dat, b_sz_const, Work2, q);           // :480
                lapack::row_sign(block_rank, R_tall_qr, b_sz_const, Work2, q);                              // :485-487
                lapack::tau_from_t(block_rank, inb, T_dat, b_sz_const, tau_sub, q);                         // :490-491
                blas::trmm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, block_rank, b_sz, (T)1.0, R_sk, d, R_tall_qr, b_sz_const, q);   // :497
                lapack::lacpy(MatrixType::Upper, block_rank, b_sz, R_tall_qr, b_sz_const, A_work, lda, q); // :504
                t_rec += us(ta, stamp());
            } else if (qr_tall == Subroutines::QRTall::geqrt) {                                             // :438-453
                t_pre += us(ta, stamp());
                ta = stamp();
                lapack::geqrt(rows, b_sz, inb, A_work, lda, T_dat, b_sz_const, Work2, q);
                lapack::tau_from_t(block_rank, inb, T_dat, b_sz_const, tau_sub, q);
                t_tall += us(ta, stamp());
            } else {                                                                                        // geqrf :506-523
                t_pre += us(ta, stamp());
                ta = stamp();
                lapack::geqrf(rows, b_sz, A_work, lda, tau_sub, q);
                have_T = false;
                t_tall += us(ta, stamp());
            }
            ta = stamp();
            // ---- apply Q^T to the trailing columns (:535-547)
            const int64_t q_rows = (block_rank != b_sz_const) ? block_rank : rows;
            if (cols - b_sz > 0 && block_rank > 0) {
                if (apply_trans_q == Subroutines::ApplyTransQ::gemqrt && have_T)                            // :535-547
                    lapack::gemqrt(Side::Left, Op::Trans, q_rows, cols - b_sz, block_rank, inb, A_work, lda, T_dat, b_sz_const, Work1, lda, q);
                else   // ormqr: the same reflectors applied from (V, tau); T_dat is free to hold the k x k block when T is not needed again
                    lapack::ormqr(Side::Left, Op::Trans, q_rows, cols - b_sz, block_rank, A_work, lda, tau_sub, Work1, lda, T_ormqr, q);
            }
            t_apply += us(ta, stamp());
            T* R12 = &R11[lda * b_sz];
            curr_sz += b_sz;
            if (curr_sz >= mn || block_rank != b_sz_const) {                                                 // :576-618
                rank = curr_sz;
                finish(t_begin, t_skop, t_qrcp, t_pre, t_tall, t_rec, t_apply, t_upd);
                return 0;
            }
            ta = stamp();
            A_work = &Work1[b_sz];                                                                           // :624
            // sketch down-date (:633-651)
            if (b_sz > 1) lapack::laset(MatrixType::Lower, b_sz - 1, b_sz, (T)0, (T)0, R_sk + 1, d, q);     // get_U(b, b, R_sk, d)
            blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, b_sz, b_sz, (T)1.0, R11, lda, R_sk, d, q);
            blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, b_sz, cols - b_sz, b_sz, (T)-1.0, R_sk, d, R12, lda, (T)1.0, &R_sk[d * b_sz], d, q);
            sampling_dimension = std::min(sampling_dimension, cols);
            if (sampling_dimension - b_sz > 1)
                lapack::laset(MatrixType::Lower, sampling_dimension - b_sz - 1, sampling_dimension - b_sz, (T)0, (T)0,
                              &R_sk[(d + 1) * b_sz] + 1, d, q);
            A_sk = &A_sk[d * b_sz];
            rows -= b_sz;
            cols -= b_sz;
            t_upd += us(ta, stamp());
        }
        return 0;
    }

    int transpose_call(int64_t mm, int64_t nn, const T* X, int64_t ldx, T* XT, int64_t ldxt) {
        if constexpr (std::is_same<T, double>::value) return rlhip_transpose_f64(q.ctx(), mm, nn, X, ldx, XT, ldxt, 0);
        else return rlhip_transpose_f32(q.ctx(), mm, nn, X, ldx, XT, ldxt, 0);
    }

    blas::Queue& q;
    bool timing;
    RandBLAS::RNGState<RNG> state;
    int64_t rank;
    int64_t block_size;
    int64_t internal_nb;
    T tol;
    std::vector<long> times;   // {skop, qrcp_wide, panel_preprocessing, qr_tall, q_reconstruction, apply_transq, sample_update, other, total} us
    Subroutines::QRCPWide qrcp_wide;
    Subroutines::QRTall qr_tall;
    Subroutines::ApplyTransQ apply_trans_q;
    // testing hooks (not in the reference): the d x n sketch to use instead of S*A, and a buffer receiving the sketch
    const T* sketch_override = nullptr;
    T* sketch_export = nullptr;

private:
    void finish(std::chrono::steady_clock::time_point t_begin, long a, long b, long c, long d_, long e, long f, long g) {
        if (!timing) return;
        q.sync();
        long total = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count();
        times = {a, b, c, d_, e, f, g, total - (a + b + c + d_ + e + f + g), total};
    }
};

/// BQRRP_GPU (reference: drivers/rl_bqrrp_gpu.hh:27-149): the reference's device driver takes the SKETCH as an input
/// (A_sk_dev, d x n, ld d) instead of generating it; same thing here on top of BQRRP (qr_tall in {cholqr, geqrf} as there).
template <typename T, typename RNG = RandBLAS::DefaultRNG>
class BQRRP_GPU {
public:
    BQRRP_GPU(blas::Queue& queue, bool time_subroutines, int64_t b_sz) : impl(queue, time_subroutines, b_sz), rank(0), block_size(b_sz) {
        impl.qrcp_wide = BQRRPSubroutines::QRCPWide::luqr;                                        // LU-QR only (:354-399)
        impl.qr_tall = BQRRPSubroutines::QRTall::cholqr;
        impl.apply_trans_q = BQRRPSubroutines::ApplyTransQ::gemqrt;
    }
    int call(int64_t m, int64_t n, T* A, int64_t lda, T* A_sk, int64_t d, T* tau, int64_t* J) {
        randlapack_require(block_size > 0 && d % block_size == 0) << "BQRRP_GPU: d=" << d << " must be a multiple of the block size";
        impl.sketch_override = A_sk;
        RandBLAS::RNGState<RNG> unused;
        const int rc = impl.call(m, n, A, lda, (T)d / (T)block_size, tau, J, unused);
        rank = impl.rank;
        times = impl.times;
        return rc;
    }
    BQRRP<T, RNG> impl;
    int64_t rank;
    int64_t block_size;
    std::vector<long> times;
};

}  // namespace RandLAPACK
