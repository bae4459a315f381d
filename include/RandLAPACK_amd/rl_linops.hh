// linops::DenseLinOp (reference: RandLAPACK/linops/rl_dense_linop.hh:30-190, concept rl_concepts.hh): the dense
// operator ABRIK drives -- `A(Side::Left, layout, opA, opB, m, n, k, alpha, B, ldb, beta, C, ldc)` is a GEMM with the
// operator on the left, `fro_nrm()` its Frobenius norm.  Device flavour: A_buff, B and C are DEVICE pointers and the
// operator carries the queue.  Only ColMajor / Side::Left is on the path (that is all ABRIK uses, rl_abrik.hh:311,364,494).
#pragma once
#include <cmath>
#include <cstdint>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"

namespace RandLAPACK::linops {

template <typename T>
struct DenseLinOp {
    using scalar_t = T;
    const int64_t n_rows;
    const int64_t n_cols;
    const T* A_buff;
    const int64_t lda;
    const Layout buff_layout;
    blas::Queue& q;

    DenseLinOp(int64_t rows, int64_t cols, const T* A, int64_t ld, Layout layout, blas::Queue& queue)
        : n_rows(rows), n_cols(cols), A_buff(A), lda(ld), buff_layout(layout), q(queue) {
        randlapack_require(layout == Layout::ColMajor) << "DenseLinOp on the device: ColMajor storage only";
        randlapack_require(lda >= n_rows) << "lda=" << lda << " < n_rows=" << n_rows << " (lda must be >= n_rows under ColMajor)";   // :59
    }

    /// `row_sharded`: this rank holds a row block of the operator (one process per GPU); A^T X is then summed over the ranks and
    /// the norm is the global one.  n_rows is the LOCAL row count.
    bool row_sharded = false;

    T fro_nrm() {                                                                                                // :67-70
        const T loc = lapack::lange(Norm::Fro, n_rows, n_cols, A_buff, lda, q);
        if (!(row_sharded && q.world() > 1)) return loc;
        double ss = (double)loc * (double)loc;
        q.allreduce_sum_host(&ss, 1);
        return (T)std::sqrt(ss);
    }

    /// C := alpha * op(A) * op(B) + beta * C                                                                     (:94-147)
    void operator()(Side side, Layout layout, Op trans_A, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb,
                    T beta, T* C, int64_t ldc) {
        randlapack_require(side == Side::Left && layout == Layout::ColMajor) << "DenseLinOp on the device: Side::Left, ColMajor only";
        const int64_t rows_A = (trans_A == Op::NoTrans) ? m : k, cols_A = (trans_A == Op::NoTrans) ? k : m;
        randlapack_require(rows_A == n_rows) << "op(A) row dim inferred from (m, k, trans_A) is " << rows_A << " but operator n_rows=" << n_rows;
        randlapack_require(cols_A == n_cols) << "op(A) col dim inferred from (m, k, trans_A) is " << cols_A << " but operator n_cols=" << n_cols;
        blas::gemm(layout, trans_A, trans_B, m, n, k, alpha, A_buff, lda, B, ldb, beta, C, ldc, q);
        if (row_sharded && q.world() > 1 && trans_A != Op::NoTrans) {     // A^T X sums over the row blocks
            randlapack_require(beta == (T)0 && ldc == m) << "sharded A^T X needs beta = 0 and a contiguous result";
            q.allreduce_sum(C, m * n);
        }
    }
    void operator()(Layout layout, Op trans_A, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb, T beta, T* C,
                    int64_t ldc) {
        (*this)(Side::Left, layout, trans_A, trans_B, m, n, k, alpha, B, ldb, beta, C, ldc);
    }
};

}  // namespace RandLAPACK::linops
