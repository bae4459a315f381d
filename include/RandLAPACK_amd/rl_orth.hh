// Stabilization objects (reference: RandLAPACK/comps/rl_orth.hh).  Same class names, constructor arguments and
// call() contract; the extra leading constructor argument is the device queue.
#pragma once
#include <cmath>
#include <limits>
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_util.hh"

namespace RandLAPACK {

template <typename T>
class Stabilization {                                             // rl_orth.hh:13-23
public:
    virtual ~Stabilization() {}
    virtual int call(int64_t m, int64_t k, T* A) = 0;
};

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
        randlapack_require(!q.reduce_over_rows()) << "HQRQ is not row-sharded (use CholQRQ across ranks)";
        blas::Scratch ws(q);
        T* tau = ws.alloc<T>(n);
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
        randlapack_require(!q.reduce_over_rows()) << "PLUL is not row-sharded";
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
