// Cholesky-QR drivers for abstract linear operators, device flavour.  One header for the four reference classes:
//   CholQR_linops          RandLAPACK/drivers/rl_cholqr_linops.hh:30-322     R = chol(A^T A)
//   sCholQR3_linops        RandLAPACK/drivers/rl_scholqr3_linops.hh:60-520   shifted CholQR + two CholQR corrections, Q never formed
//   sCholQR3_linops_basic  RandLAPACK/drivers/rl_scholqr3_linops.hh:540-786  the same with Q materialised and updated in place
//   CQRRT_linops           RandLAPACK/drivers/rl_cqrrt_linops.hh:34-449      sketch-preconditioned CholQR
// All of them see the operator only through A(Side::Left, ..., NoTrans/Trans, ...) (and the Side::Right sketch overload for CQRRT),
// so they run unchanged on linops::DenseLinOp, SparseLinOp and CompositeOperator (rl_linops.hh).
//
// Device design: every n x n object (the running inverse factor M, the Gram matrices, R) stays in HBM; the only host traffic per
// call is the potrf info word and the n diagonal entries the shift / zero-diagonal tests read.  The column-block loop
// "buf = A * M[:, blk];  G[:, blk] = A^T * buf" that all four drivers share is `detail::gram_through_operator` below; with
// block_size = 0 it is one forward and one adjoint product.
//
// R (n x n, ldr) and Q are DEVICE buffers; Q (test_mode only) is owned by the object and freed with it, as in the reference.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <type_traits>
#include <vector>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"
#include "rl_util.hh"
#include "rl_linops.hh"

namespace RandLAPACK {

namespace detail {

/// operators whose kernels want the tall intermediate in row-major order advertise it (SparseLinOp: the SpMM gathers whole rows)
template <typename GLO, typename = void>
struct prefers_row_major : std::false_type {};
template <typename GLO>
struct prefers_row_major<GLO, std::void_t<decltype(GLO::prefers_row_major)>> : std::bool_constant<GLO::prefers_row_major> {};

inline void transpose(int64_t m, int64_t n, const double* A, int64_t lda, double* AT, int64_t ldat, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_transpose_f64(q.ctx(), m, n, A, lda, AT, ldat, 0), "transpose");
}
inline void transpose(int64_t m, int64_t n, const float* A, int64_t lda, float* AT, int64_t ldat, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_transpose_f32(q.ctx(), m, n, A, lda, AT, ldat, 0), "transpose");
}

/// G (n x n, ldg) = A^T * (A * M) in column blocks of width b_eff; `buf` holds m x b_eff.  With `post` != nullptr every block is
/// first formed in Z (n x b_eff) and G[:, blk] = post^T * Z (the M^T (A^T A M) step of rl_scholqr3_linops.hh:300-312).
/// `row_major_ok`: the caller does not look at `buf` afterwards, so an operator that prefers it may be driven with
/// Layout::RowMajor operands -- buf and Z are then row-major and only n x n objects are ever transposed.
template <typename T, typename GLO>
void gram_through_operator(GLO& A, int64_t m, int64_t n, int64_t b_eff, const T* M, T* buf, const T* post, T* Z, T* G, int64_t ldg, blas::Queue& q,
                           bool row_major_ok = false) {
    if (row_major_ok && prefers_row_major<GLO>::value) {
        blas::Scratch ws(q);
        T* Mt = ws.alloc<T>(n * n);                 // M^T column-major == M row-major
        T* Zr = ws.alloc<T>(n * b_eff);             // n x bj row-major, ld bj
        transpose(n, n, M, n, Mt, n, q);
        for (int64_t j = 0; j < n; j += b_eff) {
            const int64_t bj = std::min(b_eff, n - j);
            A(Side::Left, Layout::RowMajor, Op::NoTrans, Op::NoTrans, m, bj, n, (T)1, Mt + j, n, (T)0, buf, bj);
            A(Side::Left, Layout::RowMajor, Op::Trans, Op::NoTrans, n, bj, m, (T)1, buf, bj, (T)0, Zr, bj);
            // Zr read as column-major is Z^T (bj x n, ld bj)
            if (post) blas::gemm(Layout::ColMajor, Op::Trans, Op::Trans, n, bj, n, (T)1, post, n, Zr, bj, (T)0, G + j * ldg, ldg, q);
            else transpose(bj, n, Zr, bj, G + j * ldg, ldg, q);
        }
        return;
    }
    for (int64_t j = 0; j < n; j += b_eff) {
        const int64_t bj = std::min(b_eff, n - j);
        A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, bj, n, (T)1, M + j * n, n, (T)0, buf, m);
        if (post) {
            A(Side::Left, Layout::ColMajor, Op::Trans, Op::NoTrans, n, bj, m, (T)1, buf, m, (T)0, Z, n);
            blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, n, bj, n, (T)1, post, n, Z, n, (T)0, G + j * ldg, ldg, q);
        } else
            A(Side::Left, Layout::ColMajor, Op::Trans, Op::NoTrans, n, bj, m, (T)1, buf, m, (T)0, G + j * ldg, ldg);
    }
}

/// zero the strictly lower triangle, then potrf(Upper); returns the LAPACK info
template <typename T>
int64_t chol_upper_clean(int64_t n, T* G, int64_t ldg, blas::Queue& q = blas::default_queue()) {
    if (n > 1) lapack::laset(MatrixType::Lower, n - 1, n - 1, (T)0, (T)0, G + 1, ldg, q);
    return lapack::potrf(Uplo::Upper, n, G, ldg, q);
}

template <typename T>
struct QHolder {          // the Q / Q_rows / Q_cols triple of the reference classes, device-owned
    T* Q = nullptr;
    int64_t Q_rows = 0, Q_cols = 0;
    blas::Queue* owner = nullptr;
    void reset(blas::Queue& q, int64_t rows, int64_t cols) {
        release();
        Q = blas::device_malloc<T>(rows * cols, q);
        Q_rows = rows; Q_cols = cols; owner = &q;
    }
    void release() {
        if (Q && owner) blas::device_free(Q, *owner);
        Q = nullptr; Q_rows = Q_cols = 0;
    }
};

}  // namespace detail

// ------------------------------------------------------------------------------------------------ CholQR
template <typename T>
class CholQR_linops {
public:
    bool timing;
    bool test_mode;
    T eps;
    T* Q = nullptr;
    int64_t Q_rows = 0, Q_cols = 0;
    std::vector<long> times;
    int64_t block_size;

    CholQR_linops(bool time_subroutines, T ep, bool enable_test_mode = false)
        : timing(time_subroutines), test_mode(enable_test_mode), eps(ep), block_size(0) {}
    ~CholQR_linops() { qh.release(); }

    /// R (n x n, ldr, DEVICE) = upper Cholesky factor of A^T A.  Returns 1 when the Cholesky factorization breaks down.
    template <typename GLO>
    int call(GLO& A, T* R, int64_t ldr) {
        blas::Queue& q = A.q;
        const int64_t m = A.n_rows, n = A.n_cols;
        randlapack_require(ldr >= n) << "ldr=" << ldr << " < n=" << n;
        if (n == 0) return 0;
        const int64_t b_eff = (block_size > 0 && block_size < n) ? block_size : n;
        blas::Scratch ws(q);
        T* I_mat = ws.alloc<T>(n * n);
        util::eye(n, n, I_mat, q);
        T* buf = nullptr;
        if (test_mode) { qh.reset(q, m, n); buf = qh.Q; }       // the full-width buffer doubles as Q (rl_cholqr_linops.hh:236)
        else buf = ws.alloc<T>(m * b_eff);
        detail::gram_through_operator<T>(A, m, n, b_eff, I_mat, buf, (const T*)nullptr, (T*)nullptr, R, ldr, q, !test_mode);
        if (detail::chol_upper_clean(n, R, ldr, q)) { publish_q(false); return 1; }
        if (test_mode) {
            if (b_eff < n) A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, (T)1, I_mat, n, (T)0, buf, m);
            blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, n, (T)1, R, ldr, buf, m, q);
        }
        publish_q(test_mode);
        return 0;
    }

private:
    detail::QHolder<T> qh;
    void publish_q(bool ok) {
        if (!ok) qh.release();
        Q = qh.Q; Q_rows = qh.Q_rows; Q_cols = qh.Q_cols;
    }
};

// ------------------------------------------------------------------------------------------------ shifted CholQR3
/// Common state of the two sCholQR3 flavours.  G{1,2,3}_factor are the three Cholesky factors (host copies, n x n column-major,
/// upper triangle), kept for inspection exactly as the reference keeps them.
template <typename T>
class sCholQR3_state {
public:
    bool timing;
    bool test_mode;
    T eps;
    T* Q = nullptr;
    int64_t Q_rows = 0, Q_cols = 0;
    std::vector<T> G1_factor, G2_factor, G3_factor;
    std::vector<long> times;

protected:
    sCholQR3_state(bool time_subroutines, T ep, bool enable_test_mode) : timing(time_subroutines), test_mode(enable_test_mode), eps(ep) {}
    ~sCholQR3_state() { qh.release(); }
    detail::QHolder<T> qh;
    void publish_q(bool ok) {
        if (!ok) qh.release();
        Q = qh.Q; Q_rows = qh.Q_rows; Q_cols = qh.Q_cols;
    }
    void keep_factor(std::vector<T>& dst, int64_t n, const T* G_dev, blas::Queue& q = blas::default_queue()) {
        blas::Scratch ws(q);
        T* U = ws.alloc<T>(n * n);
        lapack::laset(MatrixType::General, n, n, (T)0, (T)0, U, n, q);
        lapack::lacpy(MatrixType::Upper, n, n, G_dev, n, U, n, q);
        dst.assign((size_t)(n * n), (T)0);
        blas::copy_to_host(n * n, U, dst.data(), q);
    }
    /// G += 11 eps n trace(G) I  -- the shift of rl_scholqr3_linops.hh:243-251 (||A||_F^2 read off the Gram diagonal)
    void shift_gram(int64_t n, T* G, blas::Queue& q = blas::default_queue()) {
        std::vector<T> dg((size_t)n);
        lapack::get_diag(n, G, n, dg.data(), q);
        T norm_A_sq = 0;
        for (int64_t i = 0; i < n; ++i) norm_A_sq += dg[(size_t)i];
        lapack::add_diag(n, (T)11 * std::numeric_limits<T>::epsilon() * (T)n * norm_A_sq, G, n, q);
    }
    /// R <- G * R on the upper triangles (rl_scholqr3_linops.hh:349-354)
    void accumulate_R(int64_t n, const T* G, T* R, int64_t ldr, T* R_temp, blas::Queue& q = blas::default_queue()) {
        lapack::lacpy(MatrixType::Upper, n, n, R, ldr, R_temp, n, q);
        blas::trmm(Layout::ColMajor, Side::Left, Uplo::Upper, Op::NoTrans, Diag::NonUnit, n, n, (T)1, G, n, R_temp, n, q);
        lapack::lacpy(MatrixType::Upper, n, n, R_temp, n, R, ldr, q);
    }
};

/// Q-less variant: M = (R1 R2 R3)^-1 is carried as an explicit n x n matrix and every Gram matrix is M^T (A^T (A M)).
/// Return value: 0, or the index (1, 2, 3) of the Cholesky factorization that broke down.
template <typename T>
class sCholQR3_linops : public sCholQR3_state<T> {
public:
    int64_t block_size;
    sCholQR3_linops(bool time_subroutines, T ep, bool enable_test_mode = false)
        : sCholQR3_state<T>(time_subroutines, ep, enable_test_mode), block_size(0) {}

    template <typename GLO>
    int call(GLO& A, T* R, int64_t ldr) {
        blas::Queue& q = A.q;
        const int64_t m = A.n_rows, n = A.n_cols;
        randlapack_require(ldr >= n) << "ldr=" << ldr << " < n=" << n;
        if (n == 0) return 0;
        const int64_t b_eff = (block_size > 0 && block_size < n) ? block_size : n;
        blas::Scratch ws(q);
        T* G = ws.alloc<T>(n * n);
        T* R_temp = ws.alloc<T>(n * n);
        T* M = ws.alloc<T>(n * n);
        T* A_temp = ws.alloc<T>(m * b_eff);
        T* Z_buf = ws.alloc<T>(n * b_eff);
        util::eye(n, n, M, q);
        lapack::laset(MatrixType::General, n, n, (T)0, (T)0, R_temp, n, q);
        this->publish_q(false);

        // iteration 1: shifted Gram matrix of A itself (M = I)                                        (:224-270)
        detail::gram_through_operator<T>(A, m, n, b_eff, M, A_temp, (const T*)nullptr, (T*)nullptr, G, n, q, true);
        this->shift_gram(n, G, q);
        if (detail::chol_upper_clean(n, G, n, q)) return 1;
        this->keep_factor(this->G1_factor, n, G, q);
        lapack::lacpy(MatrixType::Upper, n, n, G, n, R, ldr, q);
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, n, n, (T)1, G, n, M, n, q);

        // iterations 2 and 3: G = M^T A^T A M, R <- chol(G) R, M <- M chol(G)^-1                       (:290-420)
        for (int it = 2; it <= 3; ++it) {
            detail::gram_through_operator<T>(A, m, n, b_eff, M, A_temp, M, Z_buf, G, n, q, true);
            if (detail::chol_upper_clean(n, G, n, q)) return it;
            this->keep_factor(it == 2 ? this->G2_factor : this->G3_factor, n, G, q);
            this->accumulate_R(n, G, R, ldr, R_temp, q);
            if (it == 2 || this->test_mode)
                blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, n, n, (T)1, G, n, M, n, q);
        }
        if (this->test_mode) {                                                                          // Q = A M  (:430-450)
            this->qh.reset(q, m, n);
            for (int64_t j = 0; j < n; j += b_eff) {
                const int64_t bj = std::min(b_eff, n - j);
                A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, bj, n, (T)1, M + j * n, n, (T)0, this->qh.Q + j * m, m);
            }
            this->publish_q(true);
        }
        return 0;
    }
};

/// Q-materialising variant: Q = A is formed once through the operator and then updated in place with syrk / trsm (:600-760).
template <typename T>
class sCholQR3_linops_basic : public sCholQR3_state<T> {
public:
    sCholQR3_linops_basic(bool time_subroutines, T ep, bool enable_test_mode = false) : sCholQR3_state<T>(time_subroutines, ep, enable_test_mode) {}

    template <typename GLO>
    int call(GLO& A, T* R, int64_t ldr) {
        blas::Queue& q = A.q;
        const int64_t m = A.n_rows, n = A.n_cols;
        randlapack_require(ldr >= n) << "ldr=" << ldr << " < n=" << n;
        if (n == 0) return 0;
        blas::Scratch ws(q);
        T* G = ws.alloc<T>(n * n);
        T* R_temp = ws.alloc<T>(n * n);
        T* M = ws.alloc<T>(n * n);
        util::eye(n, n, M, q);
        lapack::laset(MatrixType::General, n, n, (T)0, (T)0, R_temp, n, q);
        lapack::laset(MatrixType::General, n, n, (T)0, (T)0, G, n, q);
        this->publish_q(false);
        this->qh.reset(q, m, n);
        T* Q_buf = this->qh.Q;
        struct Guard {                      // an early return (Cholesky breakdown) must not leave a half-built Q behind
            sCholQR3_linops_basic* self; bool keep = false;
            ~Guard() { self->publish_q(keep); }
        } guard{this};

        A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, (T)1, M, n, (T)0, Q_buf, m);           // Q = A I
        A(Side::Left, Layout::ColMajor, Op::Trans, Op::NoTrans, n, n, m, (T)1, Q_buf, m, (T)0, G, n);             // G = A^T Q
        this->shift_gram(n, G, q);
        if (detail::chol_upper_clean(n, G, n, q)) return 1;
        this->keep_factor(this->G1_factor, n, G, q);
        lapack::lacpy(MatrixType::Upper, n, n, G, n, R, ldr, q);
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, n, n, (T)1, G, n, M, n, q);
        A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, (T)1, M, n, (T)0, Q_buf, m);           // Q1 = A R1^-1

        for (int it = 2; it <= 3; ++it) {
            blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, n, m, (T)1, Q_buf, m, (T)0, G, n, q);
            if (detail::chol_upper_clean(n, G, n, q)) return it;
            this->keep_factor(it == 2 ? this->G2_factor : this->G3_factor, n, G, q);
            if (it == 2 || this->test_mode)
                blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, n, (T)1, G, n, Q_buf, m, q);
            this->accumulate_R(n, G, R, ldr, R_temp, q);
        }
        guard.keep = this->test_mode;
        return 0;
    }
};

// ------------------------------------------------------------------------------------------------ CQRRT
template <typename T, typename RNG = RandBLAS::DefaultRNG>
class CQRRT_linops {
public:
    bool timing;
    bool test_mode;
    T eps;
    T* Q = nullptr;
    int64_t Q_rows = 0, Q_cols = 0;
    std::vector<long> times;
    int64_t nnz;
    bool use_dense_sketch;
    int64_t block_size;

    CQRRT_linops(bool time_subroutines, T ep, bool enable_test_mode = false)
        : timing(time_subroutines), test_mode(enable_test_mode), eps(ep), nnz(2), use_dense_sketch(false), block_size(0) {}
    ~CQRRT_linops() { qh.release(); }

    /// R (n x n, ldr, DEVICE) of A = QR.  Returns 1 when the sketch's R factor has a zero on its diagonal or the Cholesky
    /// factorization breaks down (rl_cqrrt_linops.hh:231-236, 330-336).
    template <typename GLO>
    int call(GLO& A, T* R, int64_t ldr, T d_factor, RandBLAS::RNGState<RNG>& state) {
        blas::Queue& q = A.q;
        const int64_t m = A.n_rows, n = A.n_cols;
        randlapack_require(ldr >= n) << "ldr=" << ldr << " < n=" << n;
        if (n == 0) return 0;
        const int64_t d = (int64_t)(d_factor * n);                                                                 // :182
        randlapack_require(d >= n) << "d = d_factor * n = " << d << " must be >= n = " << n;
        blas::Scratch ws(q);
        T* A_hat = ws.alloc<T>(d * n);
        T* tau = ws.alloc<T>(n);
        publish_q(false);

        int64_t m_glob = m;                       // a row-sharded operator sketches with the operator of the GLOBAL row count
        if (q.world() > 1) { int64_t r0; q.shard_extent(m, m_glob, r0); }
        if (use_dense_sketch) {                                                                                    // :194-201
            RandBLAS::DenseDist DD(d, m_glob);
            RandBLAS::DenseSkOp<T, RNG> S(DD, state, q);
            state = S.next_state;
            if (!sketch_override) A(Side::Right, Layout::ColMajor, Op::NoTrans, Op::NoTrans, d, n, m, (T)1, S, (T)0, A_hat, d);
        } else {                                                                                                   // :202-209
            randlapack_require(nnz >= 1 && nnz <= d) << "nnz=" << nnz << " nonzeros per column do not fit a sketch of d=" << d << " rows";
            RandBLAS::SparseDist DS(d, m_glob, nnz);
            RandBLAS::SparseSkOp<T, RNG> S(DS, state, q);
            state = S.next_state;
            if (!sketch_override) A(Side::Right, Layout::ColMajor, Op::NoTrans, Op::NoTrans, d, n, m, (T)1, S, (T)0, A_hat, d);
        }
        if (sketch_override) lapack::lacpy(MatrixType::General, d, n, sketch_override, d, A_hat, d, q);
        if (sketch_export) lapack::lacpy(MatrixType::General, d, n, A_hat, d, sketch_export, d, q);

        lapack::geqrf(d, n, A_hat, d, tau, q);                                                                     // :217

        // R_sk^-1 as an explicit upper-triangular matrix: the operator cannot be overwritten by a trsm            (:226-243)
        T* R_sk_inv = ws.alloc<T>(n * n);
        util::eye(n, n, R_sk_inv, q);
        if (!util::diag_is_nonzero(n, A_hat, d, q)) return 1;
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, n, n, (T)1, A_hat, d, R_sk_inv, n, q);
        if (n > 1) lapack::laset(MatrixType::Lower, n - 1, n - 1, (T)0, (T)0, R_sk_inv + 1, n, q);

        const int64_t b_eff = (block_size > 0 && block_size < n) ? block_size : n;                                 // :254
        T* A_pre = nullptr;
        if (test_mode) { qh.reset(q, m, n); A_pre = qh.Q; }
        else A_pre = ws.alloc<T>(m * b_eff);
        detail::gram_through_operator<T>(A, m, n, b_eff, R_sk_inv, A_pre, (const T*)nullptr, (T*)nullptr, R, ldr, q, !test_mode);   // :264-318
        blas::trmm(Layout::ColMajor, Side::Left, Uplo::Upper, Op::Trans, Diag::NonUnit, n, n, (T)1, R_sk_inv, n, R, ldr, q);   // :322
        if (lapack::potrf(Uplo::Upper, n, R, ldr, q)) { publish_q(false); return 1; }                              // :330

        if (test_mode) {                                                                                           // :341-368
            if (b_eff < n) A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, (T)1, R_sk_inv, n, (T)0, A_pre, m);
            blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, n, (T)1, R, ldr, A_pre, m, q);
        }
        if (n > 1) lapack::laset(MatrixType::Lower, n - 1, n - 1, (T)0, (T)0, R + 1, ldr, q);                      // :376
        blas::trmm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, n, n, (T)1, A_hat, d, R, ldr, q);   // :386
        publish_q(test_mode);
        return 0;
    }

    const T* sketch_override = nullptr;   // test hooks, as in CQRRT / CQRRPT: d x n device buffers (ld = d)
    T* sketch_export = nullptr;

private:
    detail::QHolder<T> qh;
    void publish_q(bool ok) {
        if (!ok) qh.release();
        Q = qh.Q; Q_rows = qh.Q_rows; Q_cols = qh.Q_cols;
    }
};

}  // namespace RandLAPACK
