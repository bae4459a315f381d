// Stabilization objects (reference: RandLAPACK/comps/rl_orth.hh).  Same class names, constructor arguments and
// call() contract; the extra leading constructor argument is the device queue.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include "rl_blaspp.hh"
#include "rl_exceptions.hh"
#include "rl_lapackpp.hh"
#include "rl_util.hh"

namespace RandLAPACK {

template <typename T>
class Stabilization {                                             // rl_orth.hh:13-23
public:
    virtual ~Stabilization() {}
    virtual int call(int64_t m, int64_t k, T* A) = 0;
};

namespace detail {

/// Q factor of the Householder QR of a ROW-SHARDED tall matrix (rank g holds m_g >= n rows), in place -- TSQR with one exchange:
///   local  A_g = Q_g R_g (geqrf + ungqr),  the N upper triangles stacked into an (N n) x n matrix by ONE all-reduce of a buffer in
///   which every rank fills its own slot,  stack = Qt R on every rank (same bits in, same kernels: same bits out),  A_g <- Q_g Qt_g.
/// The product diag(Q_g) Qt is the Q factor of the whole matrix up to the signs of its columns (Householder fixes them by the data, a
/// tree of Householder factorizations by its own intermediate data): an orthonormal basis of the same column space, which is all a
/// stabiliser or an orthonormal completion needs.
template <typename T>
int tsqr_q(blas::Queue& q, int64_t m, int64_t n, T* A, int64_t lda) {
    const int64_t N = q.world(), g = q.rank();
    // every rank must hold at least n rows (a shorter block has no n x n triangle to contribute); agreed on by all ranks
    double short_blocks = (m < n) ? 1.0 : 0.0;
    q.allreduce_sum_host(&short_blocks, 1);
    randlapack_require(short_blocks == 0.0) << "row-sharded Householder QR: every rank needs at least n = " << n << " rows";
    blas::Scratch ws(q);
    T* tau = ws.alloc<T>(n);
    T* stack = ws.alloc<T>(N * n * n);
    T* Qg = ws.alloc<T>(m * n);
    if (lapack::geqrf(m, n, A, lda, tau, q)) return 1;
    lapack::laset(MatrixType::General, N * n, n, T(0), T(0), stack, N * n, q);
    lapack::lacpy(MatrixType::Upper, n, n, A, lda, stack + g * n, N * n, q);
    lapack::ungqr(m, n, n, A, lda, tau, q);
    lapack::lacpy(MatrixType::General, m, n, A, lda, Qg, m, q);
    q.allreduce_sum(stack, N * n * n);                                   // disjoint slots, zeros elsewhere: an all-gather, bit for bit
    if (lapack::geqrf(N * n, n, stack, N * n, tau, q)) return 1;
    lapack::ungqr(N * n, n, n, stack, N * n, tau, q);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, T(1), Qg, m, stack + g * n, N * n, T(0), A, lda, q);
    return 0;
}

/// Row-pivoted LU stabiliser of a ROW-SHARDED tall matrix: tournament pivoting (communication-avoiding LU) with one exchange.
///   every rank: LU with partial pivoting of its block picks n candidate rows;  the N x n candidates (original values) are stacked by ONE
///   all-reduce;  LU with partial pivoting of the stack -- replicated -- picks the n pivot rows and IS their factorization L11 U;
///   A_g <- A_g U^-1: every row of the whole matrix expressed in the basis of the pivot rows, i.e. the rows of the unit lower trapezoidal
///   factor of the tournament-pivoted LU in their ORIGINAL positions (the column space of A, entries bounded by the tournament's growth).
/// With one rank this is partial pivoting itself.  An exactly singular U (rank-deficient input) is reported as 1: there is no L to return.
template <typename T>
int tslu_l(blas::Queue& q, int64_t m, int64_t n, T* A, int64_t lda) {
    const int64_t N = q.world(), g = q.rank();
    blas::Scratch ws(q);
    const int64_t kc = std::min(m, n);                                    // candidates this rank can offer
    T* W = ws.alloc<T>(std::max<int64_t>(m, 1) * n);
    T* stack = ws.alloc<T>(N * n * n);
    int64_t* ipiv = ws.alloc<int64_t>(n);
    lapack::laset(MatrixType::General, N * n, n, T(0), T(0), stack, N * n, q);
    if (m > 0) {
        lapack::lacpy(MatrixType::General, m, n, A, lda, W, m, q);
        lapack::getrf_pivots(m, n, W, m, ipiv, q);                        // the block's own partial pivoting: which rows would it pick?
        lapack::lacpy(MatrixType::General, m, n, A, lda, W, m, q);
        lapack::laswp(n, W, m, 1, kc, ipiv, 1, q);                        // ... those rows, with their original values, to the top
        lapack::lacpy(MatrixType::General, kc, n, W, m, stack + g * n, N * n, q);
    }
    q.allreduce_sum(stack, N * n * n);
    const int64_t info = lapack::getrf(N * n, n, stack, N * n, ipiv, q);  // replicated: identical on every rank
    if (info > 0) return 1;
    if (m > 0) blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, n, T(1), stack, N * n, A, lda, q);
    return 0;
}

}  // namespace detail

/// Cholesky-QR, Q factor only, in place: G = A^T A (upper) -> G = R^T R -> A <- A R^{-1}.   (rl_orth.hh:26-98)
/// return 1 + chol_fail = true when the Cholesky breaks down; return 1 when cond_check is on and
/// cond(R) > 1/sqrt(eps).
template <typename T>
class CholQRQ : public Stabilization<T> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    CholQRQ(bool c_check, bool verb) : CholQRQ(blas::default_queue(), c_check, verb) {}                                                   // rl_orth.hh:35-38
    CholQRQ(blas::Queue& queue, bool c_check, bool verb) : q(queue) {
        cond_check = c_check;
        verbose = verb;
        chol_fail = false;
    }
    int call(int64_t m, int64_t k, T* A) override {
        blas::Scratch ws(q);
        T* gram = ws.alloc<T>(k * k);
        if (!cond_check) {
            // the three calls below as one stream of kernels with one host read (tall, 256-aligned k); 1 = not served, fall through
            int info = 0;
            const int frc = lapack::cholqrq(m, k, A, m, gram, q.reduce_over_rows(), info, q);
            if (frc == 0) {
                if (info) { chol_fail = true; return 1; }
                return 0;
            }
        }
        lapack::laset(MatrixType::General, k, k, T(0), T(0), gram, k, q);
        blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, k, m, T(1), A, m, T(0), gram, k, q);      // :78
        if (q.reduce_over_rows()) q.allreduce_sum(gram, k * k);   // row-sharded: Gram = sum of the ranks' Grams
        if (lapack::potrf(Uplo::Upper, k, gram, k, q)) {                                                 // :81
            chol_fail = true;
            return 1;
        }
        if (cond_check) {                                                                                // :88-93
            // the reference takes the SVD of the k x k buffer whose strictly lower part is zero
            if (k > 1) lapack::laset(MatrixType::Lower, k - 1, k, T(0), T(0), gram + 1, k, q);
            if (util::cond_num_check(k, k, gram, verbose, q) > (1 / std::sqrt(std::numeric_limits<T>::epsilon())))
                return 1;
        }
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, k, T(1), gram, k, A, m, q);  // :95
        return 0;
    }
    blas::Queue& q;
    bool chol_fail;
    bool cond_check;
    bool verbose;
};

/// Householder QR, Q factor only: geqrf -> ungqr.                                         (rl_orth.hh:100-164)
/// The stabiliser RS/RF fall back on when CholQR's Gram matrix is too ill-conditioned.
template <typename T>
class HQRQ : public Stabilization<T> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    HQRQ(bool c_check, bool verb) : HQRQ(blas::default_queue(), c_check, verb) {}                                                   // rl_orth.hh:110-113
    HQRQ(blas::Queue& queue, bool c_check, bool verb) : q(queue) {
        cond_check = c_check;
        verbose = verb;
    }
    int call(int64_t m, int64_t n, T* A) override {
        if (q.reduce_over_rows()) return detail::tsqr_q(q, m, n, A, m);   // row-sharded: TSQR, one n x n-sized exchange
        blas::Scratch ws(q);
        T* tau = ws.alloc<T>(n);
        // geqrf + ungqr as one pass over a tall panel when the device can (lapack::geqrf_q: same Q to rounding), else the two calls
        if (T* Rq = ws.try_alloc<T>(n * n); Rq && lapack::geqrf_q(m, n, A, m, Rq, n, q)) return 0;
        if (lapack::geqrf(m, n, A, m, tau, q)) return 1;                                                 // :157
        lapack::ungqr(m, n, n, A, m, tau, q);                                                            // :162
        return 0;
    }
    blas::Queue& q;
    bool cond_check;
    bool verbose;
};

/// Row-pivoted LU stabiliser: A <- L[ipiv, :] (unit lower trapezoidal, rows interchanged).       (rl_orth.hh:166-230)
/// A singular U is not a failure (U is discarded); returns 0.
template <typename T>
class PLUL : public Stabilization<T> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    PLUL(bool c_check, bool verb) : PLUL(blas::default_queue(), c_check, verb) {}                                                   // rl_orth.hh:176-179
    PLUL(blas::Queue& queue, bool c_check, bool verb) : q(queue) {
        cond_check = c_check;
        verbose = verb;
    }
    int call(int64_t m, int64_t n, T* A) override {
        if (q.reduce_over_rows()) return detail::tslu_l(q, m, n, A, m);   // row-sharded: tournament pivoting, one n x n-sized exchange
        blas::Scratch ws(q);
        int64_t* ipiv = ws.alloc<int64_t>(n);
        lapack::getrf(m, n, A, m, ipiv, q);                                                              // :219
        lapack::laset(MatrixType::Upper, m, n, T(0), T(1), A, m, q);                                     // util::get_L(m, n, A, 1) :221
        lapack::laswp(n, A, m, 1, n, ipiv, 1, q);                                                        // :222
        return 0;
    }
    blas::Queue& q;
    bool cond_check;
    bool verbose;
};

}  // namespace RandLAPACK
