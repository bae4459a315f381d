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
//   SparseLinOp       device CSR (+ the CSR of the transpose, built on first use); gather SpMM, csrc/sparse.hip.  Also built from CSC
//                     (= the CSR of the transpose, SparseLinOp::from_csc) and COO (SparseLinOp::from_coo: two stable device sorts).
//   CompositeOperator implicit product left_op * right_op through a scratch buffer.
// Block views (rl_dense_linop.hh:295-330, rl_sparse_linop.hh:393-465 over rl_sparse_views.hh:41-216, rl_composite_linop.hh:505-530):
// row_block / col_block / submatrix on all three.  Dense views are pointer offsets.  Sparse views in the storage direction (rows of the
// CSR, columns of the transpose) borrow the parent's index / value arrays behind a rebased pointer array; the cross direction is built by
// one stable device transposition.  The parent must outlive its views (as in the reference); submatrix keeps its intermediate alive.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>
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

    // ---- block views (rl_dense_linop.hh:295-330): non-owning, same storage, same queue
    DenseLinOp<T> row_block(int64_t row_start, int64_t row_count) const {
        randlapack_require(row_start >= 0) << "row_start=" << row_start << " must be >= 0";
        randlapack_require(row_count > 0) << "row_count=" << row_count << " must be > 0";
        randlapack_require(row_start + row_count <= n_rows) << "row_start=" << row_start << " + row_count=" << row_count << " exceeds n_rows=" << n_rows;
        return DenseLinOp<T>(row_count, n_cols, A_buff + row_start, lda, buff_layout, q);
    }
    DenseLinOp<T> col_block(int64_t col_start, int64_t col_count) const {
        randlapack_require(col_start >= 0) << "col_start=" << col_start << " must be >= 0";
        randlapack_require(col_count > 0) << "col_count=" << col_count << " must be > 0";
        randlapack_require(col_start + col_count <= n_cols) << "col_start=" << col_start << " + col_count=" << col_count << " exceeds n_cols=" << n_cols;
        return DenseLinOp<T>(n_rows, col_count, A_buff + col_start * lda, lda, buff_layout, q);
    }
    DenseLinOp<T> submatrix(int64_t row_start, int64_t col_start, int64_t row_count, int64_t col_count) const {
        randlapack_require(row_start >= 0 && col_start >= 0) << "row_start=" << row_start << ", col_start=" << col_start << " must be >= 0";
        randlapack_require(row_count > 0 && col_count > 0) << "row_count=" << row_count << ", col_count=" << col_count << " must be > 0";
        randlapack_require(row_start + row_count <= n_rows) << "row_start=" << row_start << " + row_count=" << row_count << " exceeds n_rows=" << n_rows;
        randlapack_require(col_start + col_count <= n_cols) << "col_start=" << col_start << " + col_count=" << col_count << " exceeds n_cols=" << n_cols;
        return DenseLinOp<T>(row_count, col_count, A_buff + row_start + col_start * lda, lda, buff_layout, q);
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
    // CSR of A (null until built when the operator was given by its columns) and CSR of A^T = CSC of A (null until first needed)
    const int64_t* rowptr;
    const int64_t* colidx;
    const T* vals;
    blas::Queue& q;
    const int64_t* rowptr_t = nullptr;
    const int64_t* colidx_t = nullptr;
    const T* vals_t = nullptr;

    /// CSR input: rowptr (rows + 1), colidx (nnz), vals (nnz)
    SparseLinOp(int64_t rows, int64_t cols, int64_t nnz_, const int64_t* rp, const int64_t* ci, const T* v, blas::Queue& queue)
        : n_rows(rows), n_cols(cols), nnz(nnz_), rowptr(rp), colidx(ci), vals(v), q(queue) {
        randlapack_require(rows >= 0 && cols >= 0 && nnz_ >= 0) << "negative dimension";
    }
    /// CSC input (RandBLAS::sparse_data::CSCMatrix: colptr (cols + 1), rowidx (nnz), vals (nnz)): the CSR of the transpose, borrowed as it is;
    /// the CSR of A itself is built by one device transposition the first time a product needs it
    static SparseLinOp from_csc(int64_t rows, int64_t cols, int64_t nnz_, const int64_t* colptr, const int64_t* rowidx, const T* v, blas::Queue& queue) {
        SparseLinOp op(rows, cols, nnz_, nullptr, nullptr, nullptr, queue);
        op.rowptr_t = colptr; op.colidx_t = rowidx; op.vals_t = v;
        return op;
    }
    /// COO input (RandBLAS::sparse_data::COOMatrix: rows[nnz], cols[nnz], vals[nnz], any order): duplicate (row, col) entries are KEPT as separate
    /// entries -- every product sums them, but fro_nrm() (the norm of the value array, as in the reference) then is not ||A||_F; sorted
    /// into CSR by the stable device counting sort of csr_transpose -- once for the values, once for the column indices (the sort is
    /// deterministic and stable, so both come out in the same order)
    static SparseLinOp from_coo(int64_t rows, int64_t cols, int64_t nnz_, const int64_t* rowidx, const int64_t* colidx_in, const T* v, blas::Queue& queue) {
        randlapack_require(rows >= 0 && cols >= 0 && nnz_ >= 0) << "negative dimension";
        SparseLinOp op(rows, cols, nnz_, nullptr, nullptr, nullptr, queue);
        int64_t* rp = blas::device_malloc<int64_t>(rows + 1, queue);
        int64_t* ci = blas::device_malloc<int64_t>(std::max<int64_t>(nnz_, 1), queue);
        T* vv = blas::device_malloc<T>(std::max<int64_t>(nnz_, 1), queue);
        op.own_.assign({rp, ci, vv});
        blas::Scratch ws(queue);
        int64_t* one_row = ws.alloc<int64_t>(2);                       // the COO list as a 1 x rows "CSR" whose column indices are the ROW indices
        int64_t* junk = ws.alloc<int64_t>(std::max<int64_t>(nnz_, 1));
        double* ci_sorted = ws.alloc<double>(std::max<int64_t>(nnz_, 1));
        const int64_t ends[2] = {0, nnz_};
        blas::check(rlhip_memcpy_h2d(queue.ctx(), one_row, ends, sizeof(ends)), "h2d");
        detail::csr_transpose(1, rows, one_row, rowidx, v, rp, junk, vv, queue);
        // (the column indices ride through the same sort as 8-byte payloads: the sort only moves them)
        detail::csr_transpose(1, rows, one_row, rowidx, reinterpret_cast<const double*>(colidx_in), rp, junk, ci_sorted, queue);
        blas::device_copy_vector(nnz_, reinterpret_cast<const int64_t*>(ci_sorted), ci, queue);
        queue.sync();                                                   // (the scratch goes back when this returns)
        op.rowptr = rp; op.colidx = ci; op.vals = vv;
        return op;
    }
    /// the CSR of the transpose, built on FIRST use (stable counting sort on the device, ~1 ms for 2e6 nonzeros): an operator that is
    /// only ever applied as A * X (or lives for a single product, as behind rlhip_linop_apply) never pays for it
    void ensure_transpose() {
        if (rowptr_t) return;
        int64_t* rp = blas::device_malloc<int64_t>(n_cols + 1, q);
        int64_t* ci = blas::device_malloc<int64_t>(std::max<int64_t>(nnz, 1), q);
        T* vv = blas::device_malloc<T>(std::max<int64_t>(nnz, 1), q);
        own_.insert(own_.end(), {rp, ci, vv});
        detail::csr_transpose(n_rows, n_cols, rowptr, colidx, vals, rp, ci, vv, q);
        rowptr_t = rp; colidx_t = ci; vals_t = vv;
    }
    /// the CSR of A itself when the operator was given by its columns (from_csc, col_block views)
    void ensure_forward() {
        if (rowptr) return;
        int64_t* rp = blas::device_malloc<int64_t>(n_rows + 1, q);
        int64_t* ci = blas::device_malloc<int64_t>(std::max<int64_t>(nnz, 1), q);
        T* vv = blas::device_malloc<T>(std::max<int64_t>(nnz, 1), q);
        own_.insert(own_.end(), {rp, ci, vv});
        detail::csr_transpose(n_cols, n_rows, rowptr_t, colidx_t, vals_t, rp, ci, vv, q);
        rowptr = rp; colidx = ci; vals = vv;
    }
    SparseLinOp(SparseLinOp const&) = delete;
    SparseLinOp& operator=(SparseLinOp const&) = delete;
    SparseLinOp(SparseLinOp&& o) noexcept
        : n_rows(o.n_rows), n_cols(o.n_cols), nnz(o.nnz), rowptr(o.rowptr), colidx(o.colidx), vals(o.vals), q(o.q), rowptr_t(o.rowptr_t),
          colidx_t(o.colidx_t), vals_t(o.vals_t), row_sharded(o.row_sharded), densify_budget(o.densify_budget),
          force_densified_sketch(o.force_densified_sketch), own_(std::move(o.own_)), parent_(std::move(o.parent_)) {
        o.own_.clear();
    }
    ~SparseLinOp() {
        for (void* p : own_) blas::device_free(p, q);
    }

    // ---- block views (rl_sparse_linop.hh:393-465).  A view in the storage direction borrows the parent's index / value arrays and owns
    //      only its rebased pointer array; the parent must outlive it.
    SparseLinOp row_block(int64_t row_start, int64_t row_count) {
        randlapack_require(row_start >= 0 && row_count > 0) << "row_start=" << row_start << " must be >= 0 and row_count=" << row_count << " must be > 0";
        randlapack_require(row_start + row_count <= n_rows) << "row_start=" << row_start << " + row_count=" << row_count << " exceeds n_rows=" << n_rows;
        ensure_forward();
        int64_t base = 0;
        int64_t* rp = rebased(rowptr, row_start, row_count, base);
        int64_t end = 0;
        blas::check(rlhip_memcpy_d2h(q.ctx(), &end, rp + row_count, sizeof(int64_t)), "d2h");
        SparseLinOp v(row_count, n_cols, end, rp, colidx + base, vals + base, q);
        v.own_.push_back(rp);
        v.row_sharded = row_sharded; v.densify_budget = densify_budget; v.force_densified_sketch = force_densified_sketch;
        return v;
    }
    SparseLinOp col_block(int64_t col_start, int64_t col_count) {
        randlapack_require(col_start >= 0 && col_count > 0 && col_start + col_count <= n_cols)
            << "column range must satisfy col_start=" << col_start << " >= 0, col_count=" << col_count << " > 0, col_start+col_count <= n_cols=" << n_cols;
        ensure_transpose();
        int64_t base = 0;
        int64_t* cp = rebased(rowptr_t, col_start, col_count, base);
        int64_t end = 0;
        blas::check(rlhip_memcpy_d2h(q.ctx(), &end, cp + col_count, sizeof(int64_t)), "d2h");
        SparseLinOp v(n_rows, col_count, end, nullptr, nullptr, nullptr, q);     // given by its columns: the rows of the parent's transpose
        v.rowptr_t = cp; v.colidx_t = colidx_t + base; v.vals_t = vals_t + base;
        v.own_.push_back(cp);
        v.row_sharded = row_sharded; v.densify_budget = densify_budget; v.force_densified_sketch = force_densified_sketch;
        return v;
    }
    SparseLinOp submatrix(int64_t row_start, int64_t col_start, int64_t row_count, int64_t col_count) {
        auto rows_view = std::make_shared<SparseLinOp>(row_block(row_start, row_count));
        SparseLinOp v = rows_view->col_block(col_start, col_count);
        v.parent_ = rows_view;                                                   // (its arrays are slices of the row view's transpose)
        return v;
    }

    bool row_sharded = false;

    T fro_nrm() {
        const T* any_vals = vals ? vals : vals_t;                         // (the same entries in either storage order)
        const T loc = nnz > 0 ? lapack::lange(Norm::Fro, nnz, 1, any_vals, nnz, q) : (T)0;
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
            if (nt) { ensure_forward(); detail::csr_spmm((char)layout, m, n, k, alpha, rowptr, colidx, vals, B, ldb, beta, C, ldc, q); }
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
            else { ensure_forward(); detail::csr_spmm(flipped, n, m, k, alpha, rowptr, colidx, vals, B, ldb, beta, C, ldc, q); }
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
    std::vector<void*> own_;                      // device arrays this operator allocated (transposes, rebased pointer arrays, COO sorts)
    std::shared_ptr<SparseLinOp> parent_;         // submatrix: the row view whose arrays this view slices
    /// ptr[start .. start + count] - ptr[start] as a new device array (host round trip of count + 1 words, as the reference's
    /// csr_row_block builds its rebased vector); `base` = ptr[start]
    int64_t* rebased(const int64_t* ptr, int64_t start, int64_t count, int64_t& base) {
        std::vector<int64_t> h((size_t)count + 1);
        blas::check(rlhip_memcpy_d2h(q.ctx(), h.data(), ptr + start, h.size() * sizeof(int64_t)), "d2h");
        base = h[0];
        for (auto& x : h) x -= base;
        int64_t* out = blas::device_malloc<int64_t>(count + 1, q);
        blas::check(rlhip_memcpy_h2d(q.ctx(), out, h.data(), h.size() * sizeof(int64_t)), "h2d");
        return out;
    }

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

    // block views own the operand(s) they cut (rl_composite_linop.hh:84-104: shared ownership of the blocked side, the other side borrowed)
    CompositeOperator(int64_t rows, int64_t cols, std::shared_ptr<LinOp1> left, LinOp2& right)
        : n_rows(rows), n_cols(cols), left_op(*left), right_op(right), q(left->q), own_left_(std::move(left)) { check_dims(); }
    CompositeOperator(int64_t rows, int64_t cols, LinOp1& left, std::shared_ptr<LinOp2> right)
        : n_rows(rows), n_cols(cols), left_op(left), right_op(*right), q(left.q), own_right_(std::move(right)) { check_dims(); }
    CompositeOperator(int64_t rows, int64_t cols, std::shared_ptr<LinOp1> left, std::shared_ptr<LinOp2> right)
        : n_rows(rows), n_cols(cols), left_op(*left), right_op(*right), q(left->q), own_left_(std::move(left)), own_right_(std::move(right)) { check_dims(); }
    /// rows [row_start, row_start + row_count) of left * right = (those rows of left) * right; columns likewise from the right operand
    CompositeOperator row_block(int64_t row_start, int64_t row_count) {                                                      // :505-510
        return CompositeOperator(row_count, n_cols, std::make_shared<LinOp1>(left_op.row_block(row_start, row_count)), right_op);
    }
    CompositeOperator col_block(int64_t col_start, int64_t col_count) {                                                      // :513-518
        return CompositeOperator(n_rows, col_count, left_op, std::make_shared<LinOp2>(right_op.col_block(col_start, col_count)));
    }
    CompositeOperator submatrix(int64_t row_start, int64_t col_start, int64_t row_count, int64_t col_count) {                // :521-529
        return CompositeOperator(row_count, col_count, std::make_shared<LinOp1>(left_op.row_block(row_start, row_count)),
                                 std::make_shared<LinOp2>(right_op.col_block(col_start, col_count)));
    }

    CompositeOperator(int64_t rows, int64_t cols, LinOp1& left, LinOp2& right) : n_rows(rows), n_cols(cols), left_op(left), right_op(right), q(left.q) { check_dims(); }

private:
    std::shared_ptr<LinOp1> own_left_;
    std::shared_ptr<LinOp2> own_right_;
    void check_dims() const {
        randlapack_require(left_op.n_rows == n_rows) << "left_op.n_rows=" << left_op.n_rows << " must match composite operator n_rows=" << n_rows;
        randlapack_require(left_op.n_cols == right_op.n_rows) << "left_op.n_cols=" << left_op.n_cols << " must match right_op.n_rows=" << right_op.n_rows << " for composite operator";   // :125
        randlapack_require(right_op.n_cols == n_cols) << "right_op.n_cols=" << right_op.n_cols << " must match composite operator n_cols=" << n_cols;                               // :126
    }

public:
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
