// CQRRTalg / CQRRT (reference: RandLAPACK/drivers/rl_cqrrt.hh:20-288): the unpivoted sibling of CQRRPT -- sketch (SASO),
// Householder QR of the sketch, A <- A R_sk^-1, Cholesky QR of the preconditioned matrix, R = R_chol R_sk.  ABRIK's
// qr_exp = cqrrt panel QR (rl_abrik.hh:281-285).  Row-block sharding: as CQRRPT (partial sketches and the Gram matrix are
// all-reduced) unless `rows_replicated` says the operand is a replicated n-long object.
#pragma once
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
class CQRRTalg {
public:
    virtual ~CQRRTalg() {}
    virtual int call(int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr, T d_factor, RandBLAS::RNGState<RNG>& state) = 0;
};

template <typename T, typename RNG = RandBLAS::DefaultRNG>
class CQRRT : public CQRRTalg<T, RNG> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    CQRRT(bool time_subroutines, T ep) : CQRRT(blas::default_queue(), time_subroutines, ep) {}
    CQRRT(blas::Queue& queue, bool time_subroutines, T ep) : q(queue) {                                          // :62-72
        timing = time_subroutines;
        eps = ep;
        orthogonalization = false;
        compute_Q = true;
        nnz = 2;
        rows_replicated = false;
    }

    /// A (m x n, lda, DEVICE) -> Q when compute_Q; R (n x n, ldr, DEVICE): upper triangle written.  Returns 0, or 1 when the
    /// sketch's R has a zero on the diagonal or the Cholesky factorization breaks down (:160-164,176-180).
    int call(int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr, T d_factor, RandBLAS::RNGState<RNG>& state) override {
        randlapack_require(m >= 0 && n >= 0) << "m=" << m << ", n=" << n << " must be >= 0";
        randlapack_require(lda >= m) << "lda=" << lda << " < m=" << m;
        randlapack_require(ldr >= n) << "ldr=" << ldr << " < n=" << n;
        randlapack_require(d_factor >= (T)1.0) << "d_factor=" << d_factor << " must be >= 1.0";                // :140
        randlapack_require(!(A == nullptr && m > 0 && n > 0)) << "A buffer is null but m=" << m << " and n=" << n << " imply a nonempty matrix";
        randlapack_require(!(R == nullptr && n > 0)) << "R buffer is null but n=" << n << " > 0";
        if (n == 0) return 0;
        const bool sharded = q.world() > 1 && !rows_replicated;
        const int64_t d = (int64_t)(d_factor * n);                                                              // :143
        randlapack_require(nnz >= 1 && nnz <= d) << "nnz=" << nnz << " nonzeros per column do not fit a sketch of d=" << d << " rows";
        blas::Scratch ws(q);
        T* A_hat = ws.alloc<T>(d * n);
        T* tau = ws.alloc<T>(n);
        {
            int64_t m_glob = m, row0 = 0;
            if (sharded) q.shard_extent(m, m_glob, row0);
            RandBLAS::SparseDist DS(d, m_glob, nnz);                                                            // :146-152
            RandBLAS::SparseSkOp<T, RNG> S(DS, state, q);
            state = S.next_state;
            if (sketch_override) lapack::lacpy(MatrixType::General, d, n, sketch_override, d, A_hat, d, q);
            else if (sharded) {
                RandBLAS::sketch_rows(S, n, (T)1.0, A, lda, row0, m, (T)0.0, A_hat, d, q);
                q.allreduce_sum(A_hat, d * n);
            } else
                RandBLAS::sketch_general(Layout::ColMajor, Op::NoTrans, Op::NoTrans, d, n, m, (T)1.0, S, 0, 0, A, lda, (T)0.0, A_hat, d, q);
            if (sketch_export) lapack::lacpy(MatrixType::General, d, n, A_hat, d, sketch_export, d, q);
        }
        lapack::geqrf(d, n, A_hat, d, tau, q);                                                                  // :155
        T* R_sk = R;
        lapack::lacpy(MatrixType::Upper, n, n, A_hat, d, R_sk, ldr, q);                                         // :157
        {
            std::vector<T> diag(n);
            lapack::get_diag(n, R_sk, ldr, diag.data(), q);
            for (int64_t i = 0; i < n; ++i)
                if (diag[i] == (T)0) return 1;                                                                  // :160-164
        }
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, n, (T)1.0, R_sk, ldr, A, lda, q);   // :165
        blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, n, m, (T)1.0, A, lda, (T)0.0, R_sk, ldr, q);       // :169
        if (sharded) {
            T* G = ws.alloc<T>(n * n);
            lapack::laset(MatrixType::General, n, n, (T)0, (T)0, G, n, q);
            lapack::lacpy(MatrixType::Upper, n, n, R_sk, ldr, G, n, q);
            q.allreduce_sum(G, n * n);
            lapack::lacpy(MatrixType::Upper, n, n, G, n, R_sk, ldr, q);
        }
        if (lapack::potrf(Uplo::Upper, n, R_sk, ldr, q)) return 1;                                              // :176-180
        if (compute_Q)
            blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, n, (T)1.0, R_sk, ldr, A, lda, q);   // :184
        if (!orthogonalization)
            blas::trmm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, n, n, (T)1.0, A_hat, d, R_sk, ldr, q);  // :190
        return 0;
    }

    blas::Queue& q;
    bool timing;
    T eps;
    bool orthogonalization;
    bool compute_Q;
    int64_t nnz;
    bool rows_replicated;            // sharded queue only: the operand is an n-long (replicated) object, do not reduce
    std::vector<long> times;
    const T* sketch_override = nullptr;   // test hooks, as in CQRRPT
    T* sketch_export = nullptr;
};

}  // namespace RandLAPACK
