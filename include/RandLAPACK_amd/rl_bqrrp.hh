// BQRRPalg / BQRRP (reference: RandLAPACK/drivers/rl_bqrrp.hh:19-665; device twin rl_bqrrp_gpu.hh): blocked QR with
// randomized pivoting, GEQP3-compatible output (R above the diagonal, Householder vectors below, tau, 1-based J).
//
// Device status of the sub-routine options (BQRRPSubroutines):
//   qrcp_wide     geqp3  -> persistent device geqp3 (qrcp.hip)              | luqr  -> device getrf (lu.hip) + wide geqrf
//   qr_tall       cholqr -> trsm/syrk/potrf/trsm + orhr_col (house.hip)     | geqrf, geqrt -> step-synchronous Householder kernel
//                                                                              (correct, BLAS-2 bound: cholqr is the fast one)
//   apply_trans_q gemqrt -> compact-WY block apply on the MFMA GEMMs        | ormqr -> the same apply (tau is the
//                                                                             diagonal of T, :490-491, so both name one operator)
// Every option of the reference is available.  The object DEFAULTS to the reference's {luqr, geqrf, ormqr} (rl_bqrrp.hh:74-76);
// `use_fast_subroutines()` selects {luqr, cholqr, gemqrt}, the BLAS-3 triple.  The blocked loop itself lives in
// detail::bqrrp_factor and is shared with BQRRP_GPU (rl_bqrrp_gpu.hh), which differs only in taking the sketch from its caller.
#pragma once
#include <cstdlib>
#include <memory>
#include <chrono>
#include <cmath>
#include <limits>
#include <type_traits>
#include <vector>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"
#include "rl_util.hh"
#include "rl_sharded_panel.hh"

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

namespace detail {

/// Per-stage wall time of one factorization in microseconds (armed by BqrrpOpts::timing; every lap drains the stream first).
/// BQRRP::times folds them into the 9 entries of rl_bqrrp.hh:581-590, BQRRP_GPU::times into the 15 of rl_bqrrp_gpu.hh:829-834.
struct BqrrpLaps {
    long qrcp_main = 0;   // qrcp_wide without the sketch permutation: transpose + getrf + pivot conversion + wide geqrf, or geqp3
    long qrcp_piv = 0;    // col_swap of the sketch
    long piv_A = 0;       // col_swap of the trailing columns of A, zero test, rank estimate of the block
    long upd_J = 0;       // J <- J[J_buffer]
    long precond = 0;     // cholqr panels: A_pre = A_panel * inv(R_sk)
    long qr_tall = 0;
    long recon = 0;       // cholqr panels: Householder reconstruction, signs, R11 = R_chol * R_sk
    long apply = 0;       // Q^T applied to the trailing columns
    long upd_sk = 0;      // sketch down-date
    long lookaheads = 0;  // (not a time) block iterations whose down-date + next QRCP ran on the side queue
};

template <typename T>
struct BqrrpOpts {
    int64_t block_size;
    int64_t internal_nb;
    T tol;
    BQRRPSubroutines::QRCPWide qrcp_wide;
    BQRRPSubroutines::QRTall qr_tall;
    BQRRPSubroutines::ApplyTransQ apply_trans_q;
    bool cholqr_fallback;
    T cholqr_cond_limit_inv;
    bool timing;
    bool lookahead = true;     // run the sketch down-date and the next QRCP of the sketch BESIDE the trailing update (see bqrrp_factor)
    double lookahead_min_elems = 2.5e8;   // ... from m * n elements on (a side queue costs a scratch arena; 8192^2 measured +1 %, 16384^2 -1.4 %, 65536^2 -2.5 %)
    int64_t lookahead_min_block = 256;    // ... and from this block size on
};

/// The blocked loop of BQRRP (rl_bqrrp.hh:318-661) and of BQRRP_GPU (rl_bqrrp_gpu.hh:336-934) on one device: both classes run
/// THIS function; they differ in where the d x n sketch A_sk (ld d) comes from -- BQRRP forms S*A (:309-313), BQRRP_GPU receives it
/// from the caller (rl_bqrrp_gpu.hh:120-129) -- and in how they report the laps.  A_sk is overwritten (the reference's device
/// class says the same, rl_bqrrp_gpu.hh:354-355).  All pointers are DEVICE pointers.  Sets rank; returns 0.
template <typename T>
int bqrrp_factor(blas::Queue& q, const BqrrpOpts<T>& P, int64_t m, int64_t n, T* A, int64_t lda, T* A_sk, int64_t d, T* tau, int64_t* J,
                 int64_t& rank, int64_t& cholqr_fallbacks, BqrrpLaps& L) {
    using Sub = BQRRPSubroutines;
    using clk = std::chrono::steady_clock;
    auto stamp = [&]() { if (P.timing) q.sync(); return clk::now(); };
    auto lap = [&](long& acc, clk::time_point& t0) {
        if (!P.timing) return;
        auto t1 = stamp();
        acc += (long)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
        t0 = t1;
    };
    const int64_t mn = std::min(m, n);
    int64_t rows = m, cols = n, curr_sz = 0, b_sz = P.block_size;
    cholqr_fallbacks = 0;
    rank = 0;
    const int64_t maxiter = (int64_t)std::ceil(mn / (T)b_sz);                                               // :220
    const int64_t b_sz_const = b_sz;
    int64_t sampling_dimension = d, block_rank = b_sz, inb = P.internal_nb;
    T* A_work = A;

    blas::Scratch ws(q);
    int64_t* J_buffer = ws.alloc<int64_t>(n);
    T* R_tall_qr = ws.alloc<T>(b_sz_const * b_sz_const);
    T* T_dat = ws.alloc<T>(b_sz_const * b_sz_const);
    T* Work2 = ws.alloc<T>(n);
    const bool lu = (P.qrcp_wide == Sub::QRCPWide::luqr);
    T* T_ormqr = ws.alloc<T>(b_sz_const * b_sz_const);
    T* A_sk_trans = lu ? ws.alloc<T>(n * d) : nullptr;                                                       // :262-266
    int64_t* J_buffer_lu = lu ? ws.alloc<int64_t>(std::min(d, n)) : nullptr;
    std::vector<T> diag(b_sz_const);
    auto transpose_call = [&](blas::Queue& qq, int64_t mm, int64_t nn, const T* X, int64_t ldx, T* XT, int64_t ldxt) {
        if constexpr (std::is_same<T, double>::value) return rlhip_transpose_f64(qq.ctx(), mm, nn, X, ldx, XT, ldxt, 0);
        else return rlhip_transpose_f32(qq.ctx(), mm, nn, X, ldx, XT, ldxt, 0);
    };
    // ---- look-ahead.  The block row R12 of the trailing matrix is final after the HEAD of the compact-WY apply (W2 = T^T V^T C, C1 -= V1 W2);
    // the TAIL (C2 -= V2 W2, half of the apply's flops) is read by nothing on the way to the next panel's pivots.  So the sketch down-date and
    // the next QRCP of the sketch -- latency-bound kernels that leave most CUs idle (a quarter busy in the LU panels of C4) -- are enqueued
    // on a SIDE queue (own high-priority stream) right behind the head and run beside the tail; the main stream joins it before it touches
    // A again.  The same operations on the same data in the same order (the tail goes through the tiled GEMM kernel instead of the persistent
    // one, whose workgroups would hold every CU: different rounding of that one product, nothing else).  Off when the subroutines are timed
    // (every lap drains the streams) and for small problems (a side queue costs a scratch arena): BQRRP::lookahead, ::lookahead_min_elems and
    // ::lookahead_min_block are members, so a test forces the side-queue path at any size and compares it with the serial loop.
    // What it buys is limited by how badly the latency-bound panel kernels run BESIDE a GEMM that saturates the memory system: at C4 the LU of
    // the sketch takes 55 ms beside the tail against 15 ms alone, and the cooperative sketch QR waits for the tail's last workgroups
    // (rocprofv3 trace): 4.39 -> 4.27 s.
    const bool la_ok = P.lookahead && !P.timing && (double)m * (double)n >= P.lookahead_min_elems && b_sz_const >= P.lookahead_min_block;
    std::unique_ptr<blas::Queue> side;
    T* W2_la = la_ok ? ws.try_alloc<T>(b_sz_const * n) : nullptr;
    bool pre_on_side = false;          // this iteration's sketch (down-dated) lives on the side stream: its QRCP goes there too

    for (int64_t iter = 0; iter < maxiter; ++iter) {
        blas::Range ph_iter("Iteration");        // phase names: the reference's NVTX ranges (rl_bqrrp_gpu.hh:335-403)
        blas::Phases ph;
        b_sz = std::min(b_sz, mn - curr_sz);                                                                // :322-324
        inb = std::min(inb, b_sz);
        block_rank = b_sz;
        auto ta = stamp();
        ph("qrcp_wide");
        {
            blas::Queue& qq = pre_on_side ? *side : q;
            if (!lu) {
                lapack::geqp3(sampling_dimension, cols, A_sk, d, J_buffer, Work2, qq);                      // :336
                lap(L.qrcp_main, ta);
            } else {                                                                                        // :337-357
                blas::check(transpose_call(qq, sampling_dimension, cols, A_sk, d, A_sk_trans, n), "transposition");
                lapack::getrf_pivots(cols, sampling_dimension, A_sk_trans, n, J_buffer_lu, qq);   // only J_buffer_lu is read below
                lapack::luqrcp_piv(sampling_dimension, cols, J_buffer_lu, J_buffer, qq);
                lap(L.qrcp_main, ta);
                util::col_swap(sampling_dimension, cols, cols, A_sk, d, J_buffer, qq);
                lap(L.qrcp_piv, ta);
                lapack::geqrf(sampling_dimension, cols, A_sk, d, Work2, qq);
                lap(L.qrcp_main, ta);
            }
            if (pre_on_side) { q.wait_for(*side); pre_on_side = false; }     // the main stream continues behind the pivots and behind its own tail
        }
        ph("piv_A");
        util::col_swap(m, cols, cols, &A[lda * curr_sz], lda, J_buffer, q);                                 // :369
        bool block_zero = !lapack::any_abs_gt(rows, A_work, std::numeric_limits<T>::epsilon(), q);         // :373-379
        lap(L.piv_A, ta);
        ph("update_J");
        if (iter == 0) blas::device_copy_vector(cols, J_buffer, J, q);                                      // :383-387 / :402-406
        else util::col_swap(cols, cols, &J[curr_sz], J_buffer, q);
        lap(L.upd_J, ta);
        if (block_zero) { rank = curr_sz; return 0; }                                                       // :380-399
        T* Work1 = &A_work[lda * b_sz];
        T* R_sk = A_sk;
        ph("qr_tall");
        lapack::get_diag(b_sz, R_sk, d, diag.data(), q);
        for (int64_t i = 0; i < b_sz; ++i) {                                                                // :421-427
            if (std::abs(diag[i]) / std::abs(diag[0]) < P.tol) { block_rank = i; inb = std::min(inb, block_rank); break; }
        }
        lap(L.piv_A, ta);
        T* tau_sub = &tau[curr_sz];
        T* R11 = A_work;
        bool have_T = true;
        if (P.qr_tall == Sub::QRTall::cholqr) {                                                             // :454-505
            blas::Scratch ws_panel(q);
            T* panel_copy = nullptr;                    // the unpreconditioned panel, kept for the Householder fallback below
            if (P.cholqr_fallback) {
                panel_copy = ws_panel.alloc<T>(rows * block_rank);
                lapack::lacpy(MatrixType::General, rows, block_rank, A_work, lda, panel_copy, rows, q);
            }
            blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, rows, block_rank, (T)1.0, R_sk, d, A_work, lda, q);
            lap(L.precond, ta);
            lapack::laset(MatrixType::General, b_sz_const, b_sz_const, (T)0, (T)0, R_tall_qr, b_sz_const, q);
            blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, block_rank, rows, (T)1.0, A_work, lda, (T)0.0, R_tall_qr, b_sz_const, q);
            const int64_t chol_info = lapack::potrf(Uplo::Upper, block_rank, R_tall_qr, b_sz_const, q);
            bool chol_bad = chol_info != 0;
            if (P.cholqr_fallback && !chol_bad && block_rank > 0) {
                // Cholesky QR is only as orthogonal as eps * cond(A_pre)^2.  The preconditioner should leave cond(A_pre) = O(1); a
                // graded diagonal of R_chol (a lower bound on cond(A_pre)) says it did not -- the panel sits on the noise floor of
                // a numerically rank-deficient matrix -- and the panel goes to Householder as well.
                std::vector<T> dg((size_t)block_rank);
                lapack::get_diag(block_rank, R_tall_qr, b_sz_const, dg.data(), q);
                T dmin = std::abs(dg[0]), dmax = std::abs(dg[0]);
                for (int64_t i = 1; i < block_rank; ++i) { dmin = std::min(dmin, std::abs(dg[(size_t)i])); dmax = std::max(dmax, std::abs(dg[(size_t)i])); }
                chol_bad = !(dmin > P.cholqr_cond_limit_inv * dmax);          // also catches NaN
            }
            if (chol_bad && P.cholqr_fallback) {
                // On a Cholesky breakdown the reference carries on with the partially factored Gram matrix (:461 "handles potrf failure gracefully");
                // what the panel then holds depends on where the host potrf happened to stop, and on the device it can blow
                // up (Kahan matrix, 512 x 512: tau up to 115, ||Q'Q - I|| ~ 1e46).  A Cholesky breakdown means the
                // preconditioned panel is numerically rank deficient: restore the panel and factor THIS panel with
                // Householder reflectors instead (the qr_tall = geqrf branch), which needs no positive definiteness.
                ++cholqr_fallbacks;
                lapack::lacpy(MatrixType::General, rows, block_rank, panel_copy, rows, A_work, lda, q);   // (multiplying R_sk back would lose cond(R_sk) * eps)
                lapack::geqrf(rows, b_sz, A_work, lda, tau_sub, q);
                have_T = false;
                lap(L.qr_tall, ta);
            } else {
                blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, rows, block_rank, (T)1.0, R_tall_qr, b_sz_const, A_work, lda, q);
                lap(L.qr_tall, ta);
                ph("orhr_col");
                lapack::orhr_col(rows, block_rank, inb, A_work, lda, T_dat, b_sz_const, Work2, q);           // :480
                lapack::row_sign(block_rank, R_tall_qr, b_sz_const, Work2, q);                              // :485-487
                lapack::tau_from_t(block_rank, inb, T_dat, b_sz_const, tau_sub, q);                         // :490-491
                blas::trmm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, block_rank, b_sz, (T)1.0, R_sk, d, R_tall_qr, b_sz_const, q);   // :497
                lapack::lacpy(MatrixType::Upper, block_rank, b_sz, R_tall_qr, b_sz_const, A_work, lda, q); // :504
                lap(L.recon, ta);
            }
        } else if (P.qr_tall == Sub::QRTall::geqrt) {                                                       // :438-453
            lapack::geqrt(rows, b_sz, inb, A_work, lda, T_dat, b_sz_const, Work2, q);
            lapack::tau_from_t(block_rank, inb, T_dat, b_sz_const, tau_sub, q);
            lap(L.qr_tall, ta);
        } else {                                                                                            // geqrf :506-523
            lapack::geqrf(rows, b_sz, A_work, lda, tau_sub, q);
            have_T = false;
            lap(L.qr_tall, ta);
        }
        // ---- apply Q^T to the trailing columns (:535-547)
        const int64_t q_rows = (block_rank != b_sz_const) ? block_rank : rows;
        const bool more = (curr_sz + b_sz < mn) && (block_rank == b_sz_const);           // another panel follows (:576-618)
        const bool use_t = (P.apply_trans_q == Sub::ApplyTransQ::gemqrt && have_T);
        bool la = false;
        ph("update_A");
        if (cols - b_sz > 0 && block_rank > 0) {
            // one compact-WY block and a next panel to prepare: head on the main stream, the side queue starts behind it, tail on the main stream
            la = la_ok && W2_la && more && (!use_t || inb >= block_rank) && q_rows > block_rank;
            if (la) {
                ++L.lookaheads;
                rlhip_path_note(q.ctx(), 12, 1);
                // (the context's CACHED side queue: a side context per call costs a stream, two mailboxes and a 64 MiB arena to create and to
                //  free -- and device memory that is unmapped and remapped between launches is what this library never does in steady state,
                //  see DESIGN 4.12 "a mapping that outlived its memory")
                if (!side) side = std::make_unique<blas::Queue>(q, typename blas::Queue::CachedSide{});
                const T* Tptr = T_dat;
                int64_t ldt = b_sz_const;
                if (!use_t) { lapack::larft(q_rows, block_rank, A_work, lda, tau_sub, T_ormqr, block_rank, q); Tptr = T_ormqr; ldt = block_rank; }
                lapack::gemqrt_head(q_rows, cols - b_sz, block_rank, A_work, lda, Tptr, ldt, Work1, lda, W2_la, q);
                side->wait_for(q);                                                        // R11, R12, R_sk are final from here on
                lapack::gemqrt_tail(q_rows, cols - b_sz, block_rank, A_work, lda, W2_la, Work1, lda, q);
            } else if (use_t)                                                                                // :535-547
                lapack::gemqrt(Side::Left, Op::Trans, q_rows, cols - b_sz, block_rank, inb, A_work, lda, T_dat, b_sz_const, Work1, lda, q);
            else   // ormqr: the same reflectors applied from (V, tau); T_dat is free to hold the k x k block when T is not needed again
                lapack::ormqr(Side::Left, Op::Trans, q_rows, cols - b_sz, block_rank, A_work, lda, tau_sub, Work1, lda, T_ormqr, q);
        }
        lap(L.apply, ta);
        T* R12 = &R11[lda * b_sz];
        curr_sz += b_sz;
        if (curr_sz >= mn || block_rank != b_sz_const) { rank = curr_sz; return 0; }                        // :576-618
        A_work = &Work1[b_sz];                                                                               // :624
        // sketch down-date (:633-651) -- beside the tail of the apply when the look-ahead is on
        blas::Queue& qd = la ? *side : q;
        ph("update_Sk");
        if (b_sz > 1) lapack::laset(MatrixType::Lower, b_sz - 1, b_sz, (T)0, (T)0, R_sk + 1, d, qd);        // get_U(b, b, R_sk, d)
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, b_sz, b_sz, (T)1.0, R11, lda, R_sk, d, qd);
        blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, b_sz, cols - b_sz, b_sz, (T)-1.0, R_sk, d, R12, lda, (T)1.0, &R_sk[d * b_sz], d, qd);
        sampling_dimension = std::min(sampling_dimension, cols);
        if (sampling_dimension - b_sz > 1)
            lapack::laset(MatrixType::Lower, sampling_dimension - b_sz - 1, sampling_dimension - b_sz, (T)0, (T)0,
                          &R_sk[(d + 1) * b_sz] + 1, d, qd);
        pre_on_side = la;
        A_sk = &A_sk[d * b_sz];
        rows -= b_sz;
        cols -= b_sz;
        lap(L.upd_sk, ta);
    }
    return 0;
}

}  // namespace detail

template <typename T, typename RNG>
class BQRRP : public BQRRPalg<T, RNG> {
public:
    using Subroutines = BQRRPSubroutines;

    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    BQRRP(bool time_subroutines, int64_t b_sz) : BQRRP(blas::default_queue(), time_subroutines, b_sz) {}                                 // rl_bqrrp.hh:66-68
    BQRRP(blas::Queue& queue, bool time_subroutines, int64_t b_sz) : q(queue) {
        randlapack_require(b_sz > 0) << "BQRRP block size b_sz=" << b_sz << " must be > 0";
        timing = time_subroutines;
        tol = std::numeric_limits<T>::epsilon();
        block_size = b_sz;
        internal_nb = b_sz;
        qrcp_wide = Subroutines::QRCPWide::luqr;       // the reference's defaults (rl_bqrrp.hh:74-76): a default-constructed object
        qr_tall = Subroutines::QRTall::geqrf;          // takes the same numerical path as the reference's
        apply_trans_q = Subroutines::ApplyTransQ::ormqr;
        rank = 0;
    }
    /// Named preset: the fastest subroutine triple on the device -- LU-QR pivoting, Cholesky-QR panels (BLAS-3: syrk / potrf / trsm +
    /// Householder reconstruction) and the compact-WY apply that reuses the panel's T factor.  The reference's BQRRP_GPU runs the
    /// same Cholesky-QR panels (rl_bqrrp_gpu.hh:615-667); benchmarks and the row-sharded call use this preset.
    void use_fast_subroutines() {
        qrcp_wide = Subroutines::QRCPWide::luqr;
        qr_tall = Subroutines::QRTall::cholqr;
        apply_trans_q = Subroutines::ApplyTransQ::gemqrt;
    }

    /// A (m x n, lda), tau (min(m,n)), J (n, int64): DEVICE buffers.  Returns 0.  `rank` as in the reference (an upper
    /// bound on the numerical rank).  SURVEY.md A.8 is the behavioural spec; line numbers refer to rl_bqrrp.hh.
    int call(int64_t m, int64_t n, T* A, int64_t lda, T d_factor, T* tau, int64_t* J, RandBLAS::RNGState<RNG>& state) override {
        randlapack_require(m >= 0 && n >= 0 && lda >= m) << "bad dimensions";
        if (q.world() > 1) return call_sharded(m, n, A, lda, d_factor, tau, J, state);
        const int64_t mn = std::min(m, n);
        if (mn == 0) { rank = 0; return 0; }
        using clk = std::chrono::steady_clock;
        auto stamp = [&]() { if (timing) q.sync(); return clk::now(); };
        auto us = [](clk::time_point a, clk::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        auto t_begin = stamp();
        const int64_t d = (int64_t)(d_factor * block_size);                                                  // :224
        blas::Scratch ws(q);
        T* A_sk = ws.alloc<T>(d * n);
        if (sketch_override) {
            lapack::lacpy(MatrixType::General, d, n, sketch_override, d, A_sk, d, q);
        } else {                                                                                            // :309-313
            blas::Range ph("skop");
            T* S = blas::device_malloc<T>(d * m, q);
            RandBLAS::DenseDist D(d, m);
            state = RandBLAS::fill_dense(D, S, state, q);
            blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, d, n, m, (T)1.0, S, d, A, lda, (T)0.0, A_sk, d, q);   // lda, not m (SURVEY B)
            blas::device_free(S, q);
        }
        if (sketch_export) lapack::lacpy(MatrixType::General, d, n, A_sk, d, sketch_export, d, q);
        const long t_skop = us(t_begin, stamp());
        detail::BqrrpOpts<T> P{block_size, internal_nb, tol, qrcp_wide, qr_tall, apply_trans_q, cholqr_fallback, cholqr_cond_limit_inv, timing,
                               lookahead, lookahead_min_elems, lookahead_min_block};
        detail::BqrrpLaps L;
        detail::bqrrp_factor(q, P, m, n, A, lda, A_sk, d, tau, J, rank, cholqr_fallbacks, L);
        lookaheads = L.lookaheads;
        if (timing) {                // the reference's 9 entries (:581-590); the laps of the shared loop are finer (BQRRP_GPU reports all of them)
            q.sync();
            const long total = us(t_begin, clk::now());
            const long qrcp = L.qrcp_main + L.qrcp_piv, pre = L.piv_A + L.upd_J + L.precond;
            times = {t_skop, qrcp, pre, L.qr_tall, L.recon, L.apply, L.upd_sk,
                     total - (t_skop + qrcp + pre + L.qr_tall + L.recon + L.apply + L.upd_sk), total};
        }
        return 0;
    }


    /// Row-block sharded BQRRP (BASELINE config 4): this rank holds rows [row0, row0 + m) of the global matrix.
    /// Replicated on every rank: the sketch A_sk and everything derived from it (pivots J, rank estimates, R_sk), the b x b factors
    /// (Gram matrix, R11, T, D), tau.  Sharded by rows: A itself, i.e. the reflectors V and the trailing matrix.  Exchanges per
    /// block iteration (all-reduces over RCCL): the b x b Gram matrix of the panel ("R-factor all-reduce"), the b x b top block of
    /// the orthonormal panel (so that every rank can run the Householder reconstruction on [top block; its own rows]), W = V^T C
    /// (b x (cols - b)) for the compact-WY apply, and the b x (cols - b) block row R12 for the sketch down-date; once at the start the
    /// d x n sketch.  Options: qrcp_wide free (it acts on the replicated sketch); qr_tall = cholqr (Gram all-reduce) or geqrf / geqrt (TSQR:
    /// the b x b triangles stacked by one all-reduce) -- the reference's default triple {luqr, geqrf, ormqr} runs sharded as it stands;
    /// apply_trans_q: gemqrt and ormqr name the same operator (one b x b T block per panel, internal_nb = b) on full-rank blocks; on a
    /// rank-deficient block (block_rank < b_sz, the last one) they differ as in the single-device loop -- gemqrt applies the T factor of the whole
    /// panel, ormqr (and every geqrf panel) the T factor of the reflectors CUT to their first block_rank rows (:535-547 with q_rows = block_rank).
    ///
    /// Look-ahead (as in detail::bqrrp_factor, same members): the replicated chain -- sketch down-date + the next QRCP of the sketch, the same
    /// work on every rank whatever the world size, so the part that Amdahl's law leaves standing -- goes to the side queue as soon as the
    /// block row R12 is final and exchanged: W = V^T C and its all-reduce, W2 = T^T W, then the HEAD (my rows of the top block: C1 -= V1 W2,
    /// which finishes R12), the R12 all-reduce, and from there the side queue runs beside the TAIL (my rows below the block: C2 -= V2 W2, half
    /// of this rank's flops of the apply).  Collectives stay on the main stream in the same order on every rank; the decision is a function of
    /// rank-uniform quantities only.
    int call_sharded(int64_t m, int64_t n, T* A, int64_t lda, T d_factor, T* tau, int64_t* J, RandBLAS::RNGState<RNG>& state) {
        // qr_tall: cholqr -> Cholesky-QR of the preconditioned panel (one b x b Gram all-reduce); geqrf / geqrt -> TSQR of the panel itself
        // (local Householder QR, the ranks' b x b triangles stacked by ONE all-reduce, the stack factored on every rank).  Either way the
        // orthonormal panel is then turned into the reflectors of the WHOLE panel by Householder reconstruction, which is LAPACK's geqrf
        // representation (unique up to rounding): the sharded factorization equals the single-device one for every option.
        const bool tsqr_panels = (qr_tall != Subroutines::QRTall::cholqr);
        int64_t m_glob = m, row0 = 0;
        q.shard_extent(m, m_glob, row0);
        // Row layout of this rank: segments (first global row, count) in increasing global order, stacked in A.
        //   contiguous (default): one segment [row0, row0 + m).
        //   rows_block_cyclic: global row blocks of block_size rows dealt round-robin (block g lives on rank g % P), the layout
        //   SURVEY.md 8e asks for -- the active rows shrink from the top by one block per iteration, so contiguous row blocks idle
        //   the low ranks early while cyclic blocks keep every rank within one block of the same load.
        std::vector<int64_t> seg_g0, seg_cnt, seg_l0;
        if (rows_block_cyclic) {
            const int64_t P = q.world(), me = q.rank(), nblk = (m_glob + block_size - 1) / block_size;
            int64_t l0 = 0;
            for (int64_t g = me; g < nblk; g += P) {
                const int64_t cnt = std::min(block_size, m_glob - g * block_size);
                seg_g0.push_back(g * block_size); seg_cnt.push_back(cnt); seg_l0.push_back(l0);
                l0 += cnt;
            }
            randlapack_require(l0 == m) << "block-cyclic layout: this rank should hold " << l0 << " rows of the " << m_glob << ", got m=" << m;
        } else if (m > 0) {
            seg_g0.push_back(row0); seg_cnt.push_back(m); seg_l0.push_back(0);
        }
        // local index of my first row with global index >= g (m when there is none)
        auto local_from = [&](int64_t g) {
            for (size_t i = 0; i < seg_g0.size(); ++i)
                if (seg_g0[i] + seg_cnt[i] > g) return seg_l0[i] + std::max<int64_t>(0, g - seg_g0[i]);
            return m;
        };
        // global index of local row l
        auto global_of = [&](int64_t l) {
            for (size_t i = 0; i < seg_g0.size(); ++i)
                if (l < seg_l0[i] + seg_cnt[i]) return seg_g0[i] + (l - seg_l0[i]);
            return m_glob;
        };
        const int64_t mn = std::min(m_glob, n);
        lookaheads = 0;
        if (mn == 0) { rank = 0; return 0; }
        int64_t cols = n, curr_sz = 0, b_sz = block_size;
        const int64_t maxiter = (int64_t)std::ceil(mn / (T)b_sz);
        const int64_t b_sz_const = b_sz;
        const int64_t d = (int64_t)(d_factor * b_sz);
        int64_t sampling_dimension = d, block_rank = b_sz;
        blas::Scratch ws(q);
        int64_t* J_buffer = ws.alloc<int64_t>(n);
        T* A_sk_base = ws.alloc<T>(d * n);
        T* R_tall_qr = ws.alloc<T>(b_sz_const * b_sz_const);
        T* T_dat = ws.alloc<T>(b_sz_const * b_sz_const);
        T* Q1buf = ws.alloc<T>(b_sz_const * b_sz_const);
        T* Work2 = ws.alloc<T>(n);
        const bool lu = (qrcp_wide == Subroutines::QRCPWide::luqr);
        T* A_sk_trans = lu ? ws.alloc<T>(n * d) : nullptr;
        int64_t* J_buffer_lu = lu ? ws.alloc<int64_t>(std::min(d, n)) : nullptr;
        T* A_sk = A_sk_base;
        const bool la_ok = lookahead && (double)m_glob * (double)n >= lookahead_min_elems && b_sz_const >= lookahead_min_block;
        T* R12_la = la_ok ? ws.try_alloc<T>(b_sz_const * n) : nullptr;         // read by the side queue: lives as long as the call
        std::unique_ptr<blas::Queue> side;
        bool pre_on_side = false;
        // ---- sketch: S (d x m_glob) is ONE global Gaussian operator; this rank generates its column block S[:, row0 : row0 + m]
        //      (stream positions d*row0 ... of the same fill) and contributes S_g A_g; the state advances as for the full fill
        if (sketch_override) {
            lapack::lacpy(MatrixType::General, d, n, sketch_override, d, A_sk, d, q);
        } else {
            blas::Range ph("skop");
            blas::Scratch w2(q);
            T* S = w2.alloc<T>(std::max<int64_t>(d * m, 1));
            RandBLAS::DenseDist Dall(d * m_glob, 1);
            auto st_in = state;
            state = RandBLAS::fill_dense_rows(Dall, 0, 0, S, st_in, q);                       // the state after the full fill
            for (size_t i = 0; i < seg_g0.size(); ++i)                                         // S[:, my rows], segment by segment
                RandBLAS::fill_dense_rows(Dall, d * seg_g0[i], d * seg_cnt[i], S + d * seg_l0[i], st_in, q);
            if (m > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, d, n, m, (T)1.0, S, d, A, lda, (T)0.0, A_sk, d, q);
            else lapack::laset(MatrixType::General, d, n, (T)0, (T)0, A_sk, d, q);
            q.allreduce_sum(A_sk, d * n);
        }
        if (sketch_export) lapack::lacpy(MatrixType::General, d, n, A_sk, d, sketch_export, d, q);
        std::vector<T> diag(b_sz_const);

        for (int64_t iter = 0; iter < maxiter; ++iter) {
            blas::Range ph_iter("Iteration");
            b_sz = std::min(b_sz, mn - curr_sz);
            block_rank = b_sz;
            {
                blas::Range ph("qrcp_wide");
                blas::Queue& qq = pre_on_side ? *side : q;                                     // replicated work: every rank, the same bits
                if (!lu) {
                    lapack::geqp3(sampling_dimension, cols, A_sk, d, J_buffer, Work2, qq);
                } else {
                    blas::check(transpose_call(qq, sampling_dimension, cols, A_sk, d, A_sk_trans, n), "transposition");
                    lapack::getrf_pivots(cols, sampling_dimension, A_sk_trans, n, J_buffer_lu, qq);   // only J_buffer_lu is read below
                    lapack::luqrcp_piv(sampling_dimension, cols, J_buffer_lu, J_buffer, qq);
                    util::col_swap(sampling_dimension, cols, cols, A_sk, d, J_buffer, qq);
                    lapack::geqrf(sampling_dimension, cols, A_sk, d, Work2, qq);
                }
                if (pre_on_side) { q.wait_for(*side); pre_on_side = false; }
            }
            // local row bookkeeping for this iteration (global rows [curr_sz, m_glob) are active)
            const int64_t act_loc = local_from(curr_sz);                                      // my active rows are the local suffix [act_loc, m)
            const int64_t loc_rows = m - act_loc;
            T* A_work = (loc_rows > 0) ? &A[act_loc + lda * curr_sz] : nullptr;               // my active rows of the panel / trailing matrix
            bool block_zero;
            {
                blas::Range ph("piv_A");
                if (m > 0) util::col_swap(m, cols, cols, &A[lda * curr_sz], lda, J_buffer, q);
                double nz = 0;                                                                  // zero test on the panel's first column
                if (loc_rows > 0) nz = lapack::any_abs_gt(loc_rows, A_work, std::numeric_limits<T>::epsilon(), q) ? 1.0 : 0.0;
                q.allreduce_sum_host(&nz, 1);
                block_zero = (nz == 0.0);
            }
            {
                blas::Range ph("update_J");
                if (iter == 0) blas::device_copy_vector(cols, J_buffer, J, q);
                else util::col_swap(cols, cols, &J[curr_sz], J_buffer, q);
            }
            if (block_zero) { rank = curr_sz; return 0; }
            T* R_sk = A_sk;
            lapack::get_diag(b_sz, R_sk, d, diag.data(), q);
            for (int64_t i = 0; i < b_sz; ++i)
                if (std::abs(diag[i]) / std::abs(diag[0]) < tol) { block_rank = i; break; }
            const int64_t br = block_rank;
            // pc = the panel columns that get reflectors: a Cholesky-QR panel factors its block_rank leading columns (:454-505: the others stay as
            // they are, R11's columns to their right are R_chol R_sk), a Householder panel ALL b_sz columns whatever the rank estimate says
            // (geqrf(rows, b_sz, ...), :506-523) -- so R11 is a full b_sz x b_sz triangle and tau has b_sz entries there
            const int64_t pc = tsqr_panels ? b_sz : br;
            T* tau_sub = &tau[curr_sz];
            if (!tsqr_panels) {
                // ---- CholQR of the sharded panel: the Gram matrix is summed over the ranks
                blas::Range ph("qr_tall");
                if (loc_rows > 0) blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, loc_rows, br, (T)1.0, R_sk, d, A_work, lda, q);
                lapack::laset(MatrixType::General, b_sz_const, b_sz_const, (T)0, (T)0, R_tall_qr, b_sz_const, q);
                if (loc_rows > 0) blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, br, loc_rows, (T)1.0, A_work, lda, (T)0.0, R_tall_qr, b_sz_const, q);
                q.allreduce_sum(R_tall_qr, b_sz_const * b_sz_const);
                lapack::potrf(Uplo::Upper, br, R_tall_qr, b_sz_const, q);
                if (loc_rows > 0) blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, loc_rows, br, (T)1.0, R_tall_qr, b_sz_const, A_work, lda, q);
            } else {
                // ---- TSQR of the sharded panel (the reference's default qr_tall = geqrf, rl_bqrrp.hh:506-523, on row-sharded data): my active rows
                //      A_g = Q_g R_g, the ranks' triangles stacked by ONE all-reduce (disjoint slots: an all-gather, bit for bit), the stack = Qt R
                //      on every rank (same bits in, same kernels: same bits out), panel <- Q_g Qt_g, R_tall_qr <- R (rl_sharded_panel.hh)
                blas::Range ph("qr_tall");
                T* Rp = R_tall_qr;
                blas::Scratch wr(q);
                if (pc != b_sz_const) Rp = wr.alloc<T>(pc * pc);                              // (a last, narrower block)
                detail::tsqr(q, loc_rows, pc, A_work, lda, Rp, pc);
                if (Rp != R_tall_qr) {
                    lapack::laset(MatrixType::General, b_sz_const, b_sz_const, (T)0, (T)0, R_tall_qr, b_sz_const, q);
                    lapack::lacpy(MatrixType::Upper, pc, pc, Rp, pc, R_tall_qr, b_sz_const, q);
                }
            }
            // ---- Householder reconstruction on [top block (gathered); my rows below it]
            const int64_t top_hi = curr_sz + pc;                                             // global rows [curr_sz, top_hi) form the top block
            // my rows of the top block are the first tcnt rows of my active suffix (both layouts keep local rows in global order,
            // and a cyclic block never straddles a panel); toff = where they sit inside the block
            const int64_t tcnt = local_from(top_hi) - act_loc;
            const int64_t toff = (tcnt > 0) ? global_of(act_loc) - curr_sz : 0;
            const int64_t t_loc = act_loc, b_loc = act_loc + tcnt;                            // local starts: my top rows / my rows below the block
            const int64_t below = m - b_loc;
            // the rows the apply acts on: all my active rows, or -- on a deficient block, as the reference (:535-547 with q_rows = block_rank) --
            // only those among the first block_rank rows of the block: a prefix of my top rows
            const int64_t acnt = (br != b_sz_const) ? std::max<int64_t>(0, std::min(tcnt, local_from(curr_sz + br) - act_loc)) : tcnt;
            const bool more = (curr_sz + b_sz < mn) && (br == b_sz_const);                    // another panel follows
            bool la = false;
            if (pc > 0) {
                blas::Scratch w3(q);
                const int64_t ldp = pc + below;
                T* Pst = w3.alloc<T>(ldp * pc);
                T* Dv = w3.alloc<T>(pc);
                {
                    blas::Range ph("orhr_col");
                    lapack::laset(MatrixType::General, pc, pc, (T)0, (T)0, Q1buf, pc, q);
                    if (tcnt > 0) lapack::lacpy(MatrixType::General, tcnt, pc, &A[t_loc + lda * curr_sz], lda, Q1buf + toff, pc, q);
                    q.allreduce_sum(Q1buf, pc * pc);
                    lapack::lacpy(MatrixType::General, pc, pc, Q1buf, pc, Pst, ldp, q);
                    if (below > 0) lapack::lacpy(MatrixType::General, below, pc, &A[b_loc + lda * curr_sz], lda, Pst + pc, ldp, q);
                    lapack::orhr_col(ldp, pc, pc, Pst, ldp, T_dat, b_sz_const, Dv, q);            // one pc x pc T block (internal_nb = b)
                    lapack::row_sign(pc, R_tall_qr, b_sz_const, Dv, q);
                    lapack::tau_from_t(pc, pc, T_dat, b_sz_const, tau_sub, q);
                    if (!tsqr_panels)
                        blas::trmm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, br, b_sz, (T)1.0, R_sk, d, R_tall_qr, b_sz_const, q);   // R11 = R_chol R_sk (replicated)
                    // my rows of V back into A (strictly lower part of the top block is V1, the rest of my rows V2) ...
                    if (tcnt > 0) lapack::lacpy(MatrixType::General, tcnt, pc, Pst + toff, ldp, &A[t_loc + lda * curr_sz], lda, q);
                    if (below > 0) lapack::lacpy(MatrixType::General, below, pc, Pst + pc, ldp, &A[b_loc + lda * curr_sz], lda, q);
                    // ... and my rows of R11 on and above the diagonal of the top block
                    if (tcnt > 0) lapack::lacpy(MatrixType::Upper, tcnt, b_sz - toff, R_tall_qr + toff + toff * b_sz_const, b_sz_const,
                                                &A[t_loc + lda * (curr_sz + toff)], lda, q);
                }
                // ---- compact-WY apply to the trailing columns: W = sum over ranks of V_g^T C_g, C_g -= V_g (T^T W)
                const int64_t rest = cols - b_sz;
                const int64_t vrows = (br != b_sz_const) ? acnt : (tcnt + below);
                if (rest > 0 && br > 0) {
                    blas::Range ph("update_A");
                    T* Vexp = w3.alloc<T>(std::max<int64_t>(vrows, 1) * br);
                    T* W = w3.alloc<T>(br * rest);
                    T* W2 = w3.alloc<T>(br * rest);
                    const int64_t ldvx = std::max<int64_t>(vrows, 1);
                    if (acnt > 0) lapack::vrows_explicit(br, toff, acnt, Pst, ldp, Vexp, ldvx, q);
                    if (vrows > acnt) lapack::lacpy(MatrixType::General, below, br, Pst + pc, ldp, Vexp + acnt, ldvx, q);
                    // T of the br reflectors: the leading block of the panel's T factor, or -- deficient block under ormqr semantics / geqrf panels --
                    // the T factor of the reflectors cut to their first br rows (every rank holds the whole top block: replicated, no exchange)
                    const T* Tap = T_dat;
                    int64_t ldtap = b_sz_const;
                    const bool full_t = (br == b_sz_const) || (apply_trans_q == Subroutines::ApplyTransQ::gemqrt && qr_tall != Subroutines::QRTall::geqrf);
                    if (!full_t) {
                        T* Tcut = w3.alloc<T>(br * br);
                        lapack::larft(br, br, Pst, ldp, tau_sub, Tcut, br, q);
                        Tap = Tcut; ldtap = br;
                    }
                    T* Cg = (vrows > 0) ? &A[act_loc + lda * (curr_sz + b_sz)] : nullptr;
                    if (vrows > 0) blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, br, rest, vrows, (T)1.0, Vexp, ldvx, Cg, lda, (T)0.0, W, br, q);
                    else lapack::laset(MatrixType::General, br, rest, (T)0, (T)0, W, br, q);
                    q.allreduce_sum(W, br * rest);
                    blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, br, rest, br, (T)1.0, Tap, ldtap, W, br, (T)0.0, W2, br, q);
                    la = la_ok && R12_la && more;
                    if (!la) {
                        if (vrows > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, vrows, rest, br, (T)-1.0, Vexp, ldvx, W2, br, (T)1.0, Cg, lda, q);
                    } else {
                        ++lookaheads;
                        rlhip_path_note(q.ctx(), 12, 1);
                        if (!side) side = std::make_unique<blas::Queue>(q, typename blas::Queue::CachedSide{});
                        // head: my rows of the top block -> R12 is final; exchanged; the side queue may start
                        if (tcnt > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, tcnt, rest, br, (T)-1.0, Vexp, ldvx, W2, br, (T)1.0, Cg, lda, q);
                        lapack::laset(MatrixType::General, b_sz, rest, (T)0, (T)0, R12_la, b_sz, q);
                        if (tcnt > 0) lapack::lacpy(MatrixType::General, tcnt, rest, Cg, lda, R12_la + toff, b_sz, q);
                        q.allreduce_sum(R12_la, b_sz * rest);
                        side->wait_for(q);
                        // tail: my rows below the block, on kernels whose workgroups retire continuously (the side queue's kernels find CUs)
                        if (below > 0) {
                            blas::GiveWay gw(q);
                            blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, below, rest, br, (T)-1.0, Vexp + tcnt, ldvx, W2, br, (T)1.0, Cg + tcnt, lda, q);
                        }
                    }
                }
            }
            curr_sz += b_sz;
            if (curr_sz >= mn || block_rank != b_sz_const) { rank = curr_sz; return 0; }
            // ---- sketch down-date: R12 = the b_sz rows just finished of the updated trailing matrix, gathered from their owners
            {
                blas::Range ph("update_Sk");
                blas::Queue& qd = la ? *side : q;
                blas::Scratch w4(q);
                const int64_t rest = cols - b_sz;
                T* R12 = R12_la;
                if (!la) {
                    R12 = w4.alloc<T>(b_sz * rest);
                    lapack::laset(MatrixType::General, b_sz, rest, (T)0, (T)0, R12, b_sz, q);
                    if (tcnt > 0) lapack::lacpy(MatrixType::General, tcnt, rest, &A[t_loc + lda * curr_sz], lda, R12 + toff, b_sz, q);
                    q.allreduce_sum(R12, b_sz * rest);
                }
                if (b_sz > 1) lapack::laset(MatrixType::Lower, b_sz - 1, b_sz, (T)0, (T)0, R_sk + 1, d, qd);
                blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, b_sz, b_sz, (T)1.0, R_tall_qr, b_sz_const, R_sk, d, qd);
                blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, b_sz, rest, b_sz, (T)-1.0, R_sk, d, R12, b_sz, (T)1.0, &R_sk[d * b_sz], d, qd);
                sampling_dimension = std::min(sampling_dimension, cols);
                if (sampling_dimension - b_sz > 1)
                    lapack::laset(MatrixType::Lower, sampling_dimension - b_sz - 1, sampling_dimension - b_sz, (T)0, (T)0, &R_sk[(d + 1) * b_sz] + 1, d, qd);
                pre_on_side = la;
            }
            A_sk = &A_sk[d * b_sz];
            cols -= b_sz;
        }
        return 0;
    }

    int transpose_call(blas::Queue& qq, int64_t mm, int64_t nn, const T* X, int64_t ldx, T* XT, int64_t ldxt) {
        if constexpr (std::is_same<T, double>::value) return rlhip_transpose_f64(qq.ctx(), mm, nn, X, ldx, XT, ldxt, 0);
        else return rlhip_transpose_f32(qq.ctx(), mm, nn, X, ldx, XT, ldxt, 0);
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
    bool cholqr_fallback = true;      // qr_tall = cholqr: a panel whose Cholesky factorization breaks down is factored by geqrf instead
    int64_t cholqr_fallbacks = 0;     // number of panels of the last call that took that route
    T cholqr_cond_limit_inv = std::pow(std::numeric_limits<T>::epsilon(), (T)0.25);   // min/max of diag(R_chol) below this (1.2e-4 in double) = ill-conditioned panel
    // (not in the reference) look-ahead: the sketch down-date and the next QRCP of the sketch run on a side queue beside the tail of the
    // compact-WY apply (detail::bqrrp_factor).  Same operations on the same data in the same order; engaged for untimed calls from
    // lookahead_min_elems = m * n elements and lookahead_min_block columns per block on.  lookaheads = iterations of the last call that took it.
    bool lookahead = true;
    double lookahead_min_elems = 2.5e8;
    int64_t lookahead_min_block = 256;
    int64_t lookaheads = 0;
    bool rows_block_cyclic = false;   // sharded queue only: rows are dealt to the ranks in blocks of block_size (see call_sharded)
    // testing hooks (not in the reference): the d x n sketch to use instead of S*A, and a buffer receiving the sketch
    const T* sketch_override = nullptr;
    T* sketch_export = nullptr;
};

}  // namespace RandLAPACK
