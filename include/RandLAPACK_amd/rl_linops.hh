// Linear operators, device flavour (reference: RandLAPACK/linops/rl_concepts.hh:31-39 for the concept,
// rl_dense_linop.hh:30-330, rl_sparse_linop.hh:42-330, rl_composite_linop.hh:43-530 for the concrete types).
//
// A linear operator is anything with `n_rows`, `n_cols` and a GEMM-like call
//     A(side, layout, trans_A, trans_B, m, n, k, alpha, B, ldb, beta, C, ldc)
// (Side::Left: C = alpha op(A) op(B) + beta C;  Side::Right: C = alpha op(B) op(A) + beta C), plus the overload taking a RandBLAS
// sketching operator in place of B (Side::Right: C = alpha S op(A) + beta C -- how CQRRT_linops sketches, rl_cqrrt_linops.hh:200,207).
// ABRIK (rl_abrik.hh:311,364,494) and the linop QR drivers (rl_qr_linops.hh) are templated over it.  All B / C are DEVICE pointers
// and every operator carries the queue it runs on (member `q`).
//
//   DenseLinOp        column-major device matrix; MFMA GEMM.  `row_sharded`: one row block per rank, A^T X and S A all-reduced.
//   SparseLinOp       device CSR (+ the CSR of the transpose, built once at construction); gather SpMM, csrc/sparse.hip.
//   CompositeOperator implicit product left_op * right_op through a scratch buffer.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"

namespace RandLAPACK::linops {

namespace detail {
inline void csr_spmm(char layout, int64_t m, int64_t n, int64_t k, double alpha, const int64_t* rp, const int64_t* ci, const double* v,
                     const double* B, int64_t ldb, double beta, double* C, int64_t ldc, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_csr_spmm_f64(q.ctx(), layout, m, n, k, alpha, rp, ci, v, B, ldb, beta, C, ldc), "csr_spmm");
}
inline void csr_spmm(char layout, int64_t m, int64_t n, int64_t k, float alpha, const int64_t* rp, const int64_t* ci, const float* v,
                     const float* B, int64_t ldb, float beta, float* C, int64_t ldc, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_csr_spmm_f32(q.ctx(), layout, m, n, k, alpha, rp, ci, v, B, ldb, beta, C, ldc), "csr_spmm");
}
inline void csr_transpose(int64_t m, int64_t k, const int64_t* rp, const int64_t* ci, const double* v, int64_t* rpt, int64_t* cit, double* vt,
                          blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_csr_transpose_f64(q.ctx(), m, k, rp, ci, v, rpt, cit, vt), "csr_transpose");
}
inline void csr_transpose(int64_t m, int64_t k, const int64_t* rp, const int64_t* ci, const float* v, int64_t* rpt, int64_t* cit, float* vt,
                          blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_csr_transpose_f32(q.ctx(), m, k, rp, ci, v, rpt, cit, vt), "csr_transpose");
}
inline void csr_densify_cols(int64_t m, const int64_t* rpt, const int64_t* cit, const double* vt, int64_t c0, int64_t b, double* out, int64_t ldo,
                             blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_csr_densify_cols_f64(q.ctx(), m, rpt, cit, vt, c0, b, out, ldo), "csr_densify_cols");
}
inline void csr_densify_cols(int64_t m, const int64_t* rpt, const int64_t* cit, const float* vt, int64_t c0, int64_t b, float* out, int64_t ldo,
                             blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_csr_densify_cols_f32(q.ctx(), m, rpt, cit, vt, c0, b, out, ldo), "csr_densify_cols");
}
inline void saso_apply_csr(rlhip_saso* S, int64_t n, double alpha, const int64_t* rpt, const int64_t* cit, const double* vt, double beta, double* C,
                           int64_t ldc, int64_t row0, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_saso_apply_csr_f64(q.ctx(), S, n, alpha, rpt, cit, vt, beta, C, ldc, row0), "saso_apply_csr");
}
inline void saso_apply_csr(rlhip_saso* S, int64_t n, float alpha, const int64_t* rpt, const int64_t* cit, const float* vt, float beta, float* C,
                           int64_t ldc, int64_t row0, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_saso_apply_csr_f32(q.ctx(), S, n, alpha, rpt, cit, vt, beta, C, ldc, row0), "saso_apply_csr");
}
}  // namespace detail

// ------------------------------------------------------------------------------------------------ dense
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
        randlapack_require(lda >= n_rows) << "lda=" << lda << " < n_rows=" << n_rows << " (lda must be >= n_rows under ColMajor)";   // rl_dense_linop.hh:59
    }

    /// `row_sharded`: this rank holds a row block of the operator (one process per GPU); A^T X and S A are then summed over the
    /// ranks and the norm is the global one.  n_rows is the LOCAL row count.
    bool row_sharded = false;

    T fro_nrm() {                                                                                                // :67-70
        const T loc = lapack::lange(Norm::Fro, n_rows, n_cols, A_buff, lda, q);
        if (!(row_sharded && q.world() > 1)) return loc;
        double ss = (double)loc * (double)loc;
        q.allreduce_sum_host(&ss, 1);
        return (T)std::sqrt(ss);
    }

    /// dense operand                                                                                             (:94-147)
    void operator()(Side side, Layout layout, Op trans_A, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb,
                    T beta, T* C, int64_t ldc) {
        randlapack_require(layout == Layout::ColMajor) << "DenseLinOp on the device: ColMajor operands only";
        if (side == Side::Left) {
            const int64_t rows_A = (trans_A == Op::NoTrans) ? m : k, cols_A = (trans_A == Op::NoTrans) ? k : m;
            randlapack_require(rows_A == n_rows) << "op(A) row dim inferred from (m, k, trans_A) is " << rows_A << " but operator n_rows=" << n_rows;
            randlapack_require(cols_A == n_cols) << "op(A) col dim inferred from (m, k, trans_A) is " << cols_A << " but operator n_cols=" << n_cols;
            blas::gemm(layout, trans_A, trans_B, m, n, k, alpha, A_buff, lda, B, ldb, beta, C, ldc, q);
            if (row_sharded && q.world() > 1 && trans_A != Op::NoTrans) {     // A^T X sums over the row blocks
                randlapack_require(beta == (T)0 && ldc == m) << "sharded A^T X needs beta = 0 and a contiguous result";
                q.allreduce_sum(C, m * n);
            }
        } else {
            const int64_t rows_A = (trans_A == Op::NoTrans) ? k : n, cols_A = (trans_A == Op::NoTrans) ? n : k;
            randlapack_require(rows_A == n_rows) << "op(A) row dim inferred from (k, n, trans_A) is " << rows_A << " but operator n_rows=" << n_rows;
            randlapack_require(cols_A == n_cols) << "op(A) col dim inferred from (k, n, trans_A) is " << cols_A << " but operator n_cols=" << n_cols;
            randlapack_require(!(row_sharded && q.world() > 1)) << "Side::Right with a dense operand is not defined for a row-sharded operator";
            blas::gemm(layout, trans_B, trans_A, m, n, k, alpha, B, ldb, A_buff, lda, beta, C, ldc, q);
        }
    }
    void operator()(Layout layout, Op trans_A, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb, T beta, T* C,
                    int64_t ldc) {
        (*this)(Side::Left, layout, trans_A, trans_B, m, n, k, alpha, B, ldb, beta, C, ldc);
    }

    /// sketching operand, Side::Right: C (d x n) = alpha * S (d x m) * A + beta * C                              (:218-330)
    template <typename RNG>
    void operator()(Side side, Layout layout, Op trans_A, Op trans_S, int64_t d, int64_t n, int64_t m, T alpha, RandBLAS::SparseSkOp<T, RNG>& S,
                    T beta, T* C, int64_t ldc) {
        check_sketch_call(side, layout, trans_A, trans_S, d, n, m, S.dist.n_rows, S.dist.n_cols, ldc);
        if (row_sharded && q.world() > 1) {
            randlapack_require(beta == (T)0 && ldc == d) << "sharded S A needs beta = 0 and a contiguous result";
            int64_t m_glob, row0;
            q.shard_extent(n_rows, m_glob, row0);
            randlapack_require(S.dist.n_cols == m_glob) << "sketching operator has " << S.dist.n_cols << " columns, the sharded operator " << m_glob << " rows";
            RandBLAS::sketch_rows(S, n, alpha, A_buff, lda, row0, n_rows, (T)0, C, ldc, q);
            q.allreduce_sum(C, d * n);
        } else
            RandBLAS::sketch_general(layout, trans_S, trans_A, d, n, m, alpha, S, 0, 0, A_buff, lda, beta, C, ldc, q);
    }
    template <typename RNG>
    void operator()(Side side, Layout layout, Op trans_A, Op trans_S, int64_t d, int64_t n, int64_t m, T alpha, RandBLAS::DenseSkOp<T, RNG>& S,
                    T beta, T* C, int64_t ldc) {
        check_sketch_call(side, layout, trans_A, trans_S, d, n, m, S.dist.n_rows, S.dist.n_cols, ldc);
        randlapack_require(!(row_sharded && q.world() > 1)) << "dense sketching operators are not sharded: use the sparse one";
        RandBLAS::fill_dense(S);
        blas::gemm(layout, Op::NoTrans, Op::NoTrans, d, n, m, alpha, S.buff, d, A_buff, lda, beta, C, ldc, q);
    }

private:
    void check_sketch_call(Side side, Layout layout, Op trans_A, Op trans_S, int64_t d, int64_t n, int64_t m, int64_t s_rows, int64_t s_cols,
                           int64_t ldc) const {
        randlapack_require(side == Side::Right && layout == Layout::ColMajor && trans_A == Op::NoTrans && trans_S == Op::NoTrans)
            << "sketching operand: only the plain left sketch C = S * A (Side::Right, ColMajor, NoTrans, NoTrans) is on the path";
        randlapack_require(m == n_rows || row_sharded) << "inner dimension " << m << " != operator n_rows=" << n_rows;
        randlapack_require(n == n_cols) << "result has " << n << " columns but operator n_cols=" << n_cols;
        randlapack_require(s_rows == d && (s_cols == m || row_sharded)) << "sketching operator is " << s_rows << " x " << s_cols << ", call asks for " << d << " x " << m;
        randlapack_require(ldc >= d) << "ldc=" << ldc << " < d=" << d;
    }
};

// ------------------------------------------------------------------------------------------------ sparse
/// Device CSR operator.  rowptr (n_rows + 1), colidx (nnz), vals (nnz) are DEVICE arrays with int64 indices, borrowed for the
/// lifetime of the operator; entries of a row need not be sorted; duplicates are summed.  The CSR of the transpose is built on
/// first use (stable counting sort on the device), so that op(A) X is always a gather.
/// `row_sharded`: this rank holds a block of ROWS of the operator (n_rows = local rows, indices local); A^T X, S A and the norm
/// are then summed over the ranks, exactly as for DenseLinOp.
template <typename T>
struct SparseLinOp {
    using scalar_t = T;
    static constexpr bool prefers_row_major = true;     // the SpMM kernels are row-major inside (rl_qr_linops.hh exploits it)
    const int64_t n_rows;
    const int64_t n_cols;
    const int64_t nnz;
    const int64_t* rowptr;
    const int64_t* colidx;
    const T* vals;
    blas::Queue& q;
    int64_t* rowptr_t = nullptr;
    int64_t* colidx_t = nullptr;
    T* vals_t = nullptr;

    SparseLinOp(int64_t rows, int64_t cols, int64_t nnz_, const int64_t* rp, const int64_t* ci, const T* v, blas::Queue& queue)
        : n_rows(rows), n_cols(cols), nnz(nnz_), rowptr(rp), colidx(ci), vals(v), q(queue) {
        randlapack_require(rows >= 0 && cols >= 0 && nnz_ >= 0) << "negative dimension";
    }
    /// the CSR of the transpose, built on FIRST use (stable counting sort on the device, ~1 ms for 2e6 nonzeros): an operator that is
    /// only ever applied as A * X (or lives for a single product, as behind rlhip_linop_apply) never pays for it
    void ensure_transpose() {
        if (rowptr_t) return;
        rowptr_t = blas::device_malloc<int64_t>(n_cols + 1, q);
        colidx_t = blas::device_malloc<int64_t>(nnz, q);
        vals_t = blas::device_malloc<T>(nnz, q);
        detail::csr_transpose(n_rows, n_cols, rowptr, colidx, vals, rowptr_t, colidx_t, vals_t, q);
    }
    SparseLinOp(SparseLinOp const&) = delete;
    SparseLinOp& operator=(SparseLinOp const&) = delete;
    ~SparseLinOp() {
        if (rowptr_t) blas::device_free(rowptr_t, q);
        if (colidx_t) blas::device_free(colidx_t, q);
        if (vals_t) blas::device_free(vals_t, q);
    }

    bool row_sharded = false;

    T fro_nrm() {
        const T loc = nnz > 0 ? lapack::lange(Norm::Fro, nnz, 1, vals, nnz, q) : (T)0;
        if (!(row_sharded && q.world() > 1)) return loc;
        double ss = (double)loc * (double)loc;
        q.allreduce_sum_host(&ss, 1);
        return (T)std::sqrt(ss);
    }

    /// dense operand (rl_sparse_linop.hh:125-197); op(B) = B only
    void operator()(Side side, Layout layout, Op trans_A, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb,
                    T beta, T* C, int64_t ldc) {
        randlapack_require(trans_B == Op::NoTrans) << "SparseLinOp on the device: op(B) = B only";
        const bool nt = (trans_A == Op::NoTrans);
        if (side == Side::Left) {
            const int64_t rows_A = nt ? m : k, cols_A = nt ? k : m;
            randlapack_require(rows_A == n_rows && cols_A == n_cols) << "op(A) inferred as " << rows_A << " x " << cols_A << " but the operator is " << n_rows << " x " << n_cols;
            if (nt) detail::csr_spmm((char)layout, m, n, k, alpha, rowptr, colidx, vals, B, ldb, beta, C, ldc, q);
            else {
                ensure_transpose();
                detail::csr_spmm((char)layout, m, n, k, alpha, rowptr_t, colidx_t, vals_t, B, ldb, beta, C, ldc, q);
                if (row_sharded && q.world() > 1) {                       // A^T X sums over the row blocks
                    randlapack_require(beta == (T)0 && ldc == ((layout == Layout::ColMajor) ? m : n)) << "sharded A^T X needs beta = 0 and a contiguous result";
                    q.allreduce_sum(C, m * n);
                }
            }
        } else {
            randlapack_require(!(row_sharded && q.world() > 1)) << "Side::Right with a dense operand is not defined for a row-sharded operator";
            // C (m x n) = B (m x k) * op(A) (k x n)  <=>  C^T = op(A)^T * B^T, and a column-major matrix IS its transpose in row-major
            const int64_t rows_A = nt ? k : n, cols_A = nt ? n : k;
            randlapack_require(rows_A == n_rows && cols_A == n_cols) << "op(A) inferred as " << rows_A << " x " << cols_A << " but the operator is " << n_rows << " x " << n_cols;
            const char flipped = (layout == Layout::ColMajor) ? 'R' : 'C';
            if (nt) { ensure_transpose(); detail::csr_spmm(flipped, n, m, k, alpha, rowptr_t, colidx_t, vals_t, B, ldb, beta, C, ldc, q); }
            else detail::csr_spmm(flipped, n, m, k, alpha, rowptr, colidx, vals, B, ldb, beta, C, ldc, q);
        }
    }
    void operator()(Layout layout, Op trans_A, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb, T beta, T* C,
                    int64_t ldc) {
        (*this)(Side::Left, layout, trans_A, trans_B, m, n, k, alpha, B, ldb, beta, C, ldc);
    }

    /// sketching operands, Side::Right: C (d x n) = alpha * S * A + beta * C                                     (:292-330)
    /// Sparse S: a scatter over the nonzeros of A with fixed-point integer LDS atomics (bitwise reproducible, csrc/sketch.hip) --
    /// nnz(S column) * nnz(A) updates instead of a pass over m x n.  Sketches too tall for LDS (d > 19200) fall back to expanding
    /// column blocks of A to dense and pushing them through the dense SASO kernel.
    template <typename RNG>
    void operator()(Side side, Layout layout, Op trans_A, Op trans_S, int64_t d, int64_t n, int64_t m, T alpha, RandBLAS::SparseSkOp<T, RNG>& S,
                    T beta, T* C, int64_t ldc) {
        const bool sharded = row_sharded && q.world() > 1;
        check_sketch_call(side, layout, trans_A, trans_S, d, n, m, S.dist.n_rows, sharded ? m : S.dist.n_cols, ldc);
        if (n == 0 || d == 0) return;
        if (sharded) {                                                    // S was built for the GLOBAL row count: my rows start at row0
            randlapack_require(beta == (T)0 && ldc == d && d <= 19200) << "sharded S A needs beta = 0, a contiguous result and d <= 19200";
            int64_t m_glob, row0;
            q.shard_extent(n_rows, m_glob, row0);
            randlapack_require(S.dist.n_cols == m_glob) << "sketching operator has " << S.dist.n_cols << " columns, the sharded operator " << m_glob << " rows";
            ensure_transpose();
            detail::saso_apply_csr(S.handle, n, alpha, rowptr_t, colidx_t, vals_t, (T)0, C, ldc, row0, q);
            q.allreduce_sum(C, d * n);
            return;
        }
        if (d <= 19200 && !force_densified_sketch) {
            ensure_transpose();
            detail::saso_apply_csr(S.handle, n, alpha, rowptr_t, colidx_t, vals_t, beta, C, ldc, 0, q);
            return;
        }
        const int64_t b = std::max<int64_t>(1, std::min<int64_t>(n, densify_budget / std::max<int64_t>(m, 1)));
        blas::Scratch ws(q);
        T* blk = ws.alloc<T>(m * b);
        for (int64_t j = 0; j < n; j += b) {
            const int64_t bj = std::min(b, n - j);
            ensure_transpose();
            detail::csr_densify_cols(m, rowptr_t, colidx_t, vals_t, j, bj, blk, m, q);
            RandBLAS::sketch_general(layout, Op::NoTrans, Op::NoTrans, d, bj, m, alpha, S, 0, 0, blk, m, beta, C + j * ldc, ldc, q);
        }
    }
    /// Dense S (d x m column-major = S^T row-major): C^T (n x d, row-major) = A^T * S^T is a single gather SpMM.
    template <typename RNG>
    void operator()(Side side, Layout layout, Op trans_A, Op trans_S, int64_t d, int64_t n, int64_t m, T alpha, RandBLAS::DenseSkOp<T, RNG>& S,
                    T beta, T* C, int64_t ldc) {
        randlapack_require(!(row_sharded && q.world() > 1)) << "dense sketching operators are not sharded: use the sparse one";
        check_sketch_call(side, layout, trans_A, trans_S, d, n, m, S.dist.n_rows, S.dist.n_cols, ldc);
        RandBLAS::fill_dense(S);
        ensure_transpose();
        detail::csr_spmm('R', n, d, m, alpha, rowptr_t, colidx_t, vals_t, S.buff, d, beta, C, ldc, q);
    }

    int64_t densify_budget = (int64_t)1 << 27;   // elements of dense scratch the densified sketch path may use (1 GiB in fp64)
    bool force_densified_sketch = false;          // tests: take the fallback path regardless of d

private:
    void check_sketch_call(Side side, Layout layout, Op trans_A, Op trans_S, int64_t d, int64_t n, int64_t m, int64_t s_rows, int64_t s_cols,
                           int64_t ldc) const {
        randlapack_require(side == Side::Right && layout == Layout::ColMajor && trans_A == Op::NoTrans && trans_S == Op::NoTrans)
            << "sketching operand: only the plain left sketch C = S * A (Side::Right, ColMajor, NoTrans, NoTrans) is on the path";
        randlapack_require(m == n_rows && n == n_cols) << "call asks for an operator of " << m << " x " << n << ", this one is " << n_rows << " x " << n_cols;
        randlapack_require(s_rows == d && s_cols == m) << "sketching operator is " << s_rows << " x " << s_cols << ", call asks for " << d << " x " << m;
        randlapack_require(ldc >= d) << "ldc=" << ldc << " < d=" << d;
    }
};

// ------------------------------------------------------------------------------------------------ composite
/// left_op * right_op, never formed (rl_composite_linop.hh:43-127: both operands borrowed).  The intermediate lives in the queue's
/// scratch arena for the duration of one call.
template <typename LinOp1, typename LinOp2>
struct CompositeOperator {
    using T = typename LinOp1::scalar_t;
    using scalar_t = T;
    const int64_t n_rows;
    const int64_t n_cols;
    LinOp1& left_op;
    LinOp2& right_op;
    blas::Queue& q;

    CompositeOperator(int64_t rows, int64_t cols, LinOp1& left, LinOp2& right) : n_rows(rows), n_cols(cols), left_op(left), right_op(right), q(left.q) {
        randlapack_require(left_op.n_rows == n_rows) << "left_op.n_rows=" << left_op.n_rows << " must match composite operator n_rows=" << n_rows;
        randlapack_require(left_op.n_cols == right_op.n_rows) << "left_op.n_cols=" << left_op.n_cols << " must match right_op.n_rows=" << right_op.n_rows << " for composite operator";   // :125
        randlapack_require(right_op.n_cols == n_cols) << "right_op.n_cols=" << right_op.n_cols << " must match composite operator n_cols=" << n_cols;                               // :126
    }

    /// dense operand (:168-282).  Side::Left: NoTrans = left (right B), Trans = right^T (left^T B).
    /// Side::Right: NoTrans = (B left) right, Trans = (B right^T) left^T.
    void operator()(Side side, Layout layout, Op trans_comp, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb,
                    T beta, T* C, int64_t ldc) {
        randlapack_require(layout == Layout::ColMajor) << "CompositeOperator on the device: ColMajor operands only";
        const int64_t inner = left_op.n_cols;
        blas::Scratch ws(q);
        if (side == Side::Left) {
            T* tmp = ws.alloc<T>(inner * n);
            if (trans_comp == Op::NoTrans) {
                right_op(Side::Left, layout, Op::NoTrans, trans_B, inner, n, k, (T)1, B, ldb, (T)0, tmp, inner);
                left_op(Side::Left, layout, Op::NoTrans, Op::NoTrans, m, n, inner, alpha, tmp, inner, beta, C, ldc);
            } else {
                left_op(Side::Left, layout, Op::Trans, trans_B, inner, n, k, (T)1, B, ldb, (T)0, tmp, inner);
                right_op(Side::Left, layout, Op::Trans, Op::NoTrans, m, n, inner, alpha, tmp, inner, beta, C, ldc);
            }
        } else {
            T* tmp = ws.alloc<T>(m * inner);
            if (trans_comp == Op::NoTrans) {
                left_op(Side::Right, layout, Op::NoTrans, trans_B, m, inner, k, (T)1, B, ldb, (T)0, tmp, m);
                right_op(Side::Right, layout, Op::NoTrans, Op::NoTrans, m, n, inner, alpha, tmp, m, beta, C, ldc);
            } else {
                right_op(Side::Right, layout, Op::Trans, trans_B, m, inner, k, (T)1, B, ldb, (T)0, tmp, m);
                left_op(Side::Right, layout, Op::Trans, Op::NoTrans, m, n, inner, alpha, tmp, m, beta, C, ldc);
            }
        }
    }
    void operator()(Layout layout, Op trans_comp, Op trans_B, int64_t m, int64_t n, int64_t k, T alpha, const T* B, int64_t ldb, T beta, T* C,
                    int64_t ldc) {
        (*this)(Side::Left, layout, trans_comp, trans_B, m, n, k, alpha, B, ldb, beta, C, ldc);
    }

    /// sketching operand, Side::Right, NoTrans: C = alpha * (S left) right + beta * C                             (:399-482)
    template <typename SkOp>
    void operator()(Side side, Layout layout, Op trans_comp, Op trans_S, int64_t d, int64_t n, int64_t m, T alpha, SkOp& S, T beta, T* C,
                    int64_t ldc) {
        randlapack_require(side == Side::Right && layout == Layout::ColMajor && trans_comp == Op::NoTrans && trans_S == Op::NoTrans)
            << "sketching operand: only the plain left sketch C = S * A (Side::Right, ColMajor, NoTrans, NoTrans) is on the path";
        const int64_t inner = left_op.n_cols;
        blas::Scratch ws(q);
        T* tmp = ws.alloc<T>(d * inner);
        left_op(Side::Right, layout, Op::NoTrans, trans_S, d, inner, m, (T)1, S, (T)0, tmp, d);
        right_op(Side::Right, layout, Op::NoTrans, Op::NoTrans, d, n, inner, alpha, tmp, d, beta, C, ldc);
    }
};

}  // namespace RandLAPACK::linops
