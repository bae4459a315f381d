// The symmetric (Nystrom) analogue of the sketch-and-factor path, device flavour: one header for
//   linops::ExplicitSymLinOp  RandLAPACK/linops/rl_sym_linops.hh:55-118   symmetric matrix given by one stored triangle
//   SYPS                      RandLAPACK/comps/rl_syps.hh:42-145           power sketch  S <- (A^p) Omega with QR stabilisation
//   SYRF                      RandLAPACK/comps/rl_syrf.hh:18-95            range finder  Q = orth(A S)
//   power_error_est, REVD2    RandLAPACK/drivers/rl_revd2.hh:34-246        A ~ V diag(eigvals) V^T, rank doubled until the
//                                                                          power-method error estimate meets the tolerance
//
// Device design.  The operator symmetrises its stored triangle ONCE into a full m x m HBM copy (the other triangle is never
// read) so that every product A * X is a plain MFMA GEMM; everything m x k lives in HBM, the k x k Nystrom core goes through
// the device potrf / trsm / gesdd, and only scalars (nu, the error estimate) and the k eigenvalues touch the host.
// std::vector cannot hold device memory: V and eigvals are DEVICE buffers the driver (re)allocates (blas::device_free them),
// exactly like RSVD's U, S, V; SYRF's Q is a caller-provided device buffer of m * k.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <vector>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"
#include "rl_util.hh"
#include "rl_orth.hh"

namespace RandLAPACK {

namespace linops {

template <typename T>
struct ExplicitSymLinOp {
    using scalar_t = T;
    const int64_t dim;
    const int64_t n_rows;
    const int64_t n_cols;
    const Uplo uplo;
    const T* A_buff;
    const int64_t lda;
    const Layout buff_layout;
    blas::Queue& q;
    T* full = nullptr;        // dim x dim symmetrised copy, built on first use

    ExplicitSymLinOp(int64_t m, Uplo ul, const T* A, int64_t ld, Layout layout, blas::Queue& queue)
        : dim(m), n_rows(m), n_cols(m), uplo(ul), A_buff(A), lda(ld), buff_layout(layout), q(queue) {
        randlapack_require(layout == Layout::ColMajor) << "ExplicitSymLinOp on the device: ColMajor storage only";
        randlapack_require(ul == Uplo::Upper || ul == Uplo::Lower) << "uplo must be Upper or Lower";
        randlapack_require(lda >= m) << "lda=" << lda << " < dim=" << m;
    }
    ExplicitSymLinOp(ExplicitSymLinOp const&) = delete;
    ExplicitSymLinOp& operator=(ExplicitSymLinOp const&) = delete;
    ~ExplicitSymLinOp() { if (full) blas::device_free(full, q); }

    /// C (dim x n) = alpha * A * B + beta * C                                                            (rl_sym_linops.hh:77-97)
    void operator()(Layout layout, int64_t n, T alpha, const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {
        randlapack_require(layout == Layout::ColMajor) << "ExplicitSymLinOp on the device: ColMajor operands only";
        randlapack_require(ldb >= dim) << "ldb=" << ldb << " < dim=" << dim;
        randlapack_require(ldc >= dim) << "ldc=" << ldc << " < dim=" << dim;
        materialise();
        blas::gemm(layout, Op::NoTrans, Op::NoTrans, dim, n, dim, alpha, full, dim, B, ldb, beta, C, ldc, q);
    }

private:
    void materialise() {
        if (full || dim == 0) return;
        full = blas::device_malloc<T>(dim * dim, q);
        if constexpr (sizeof(T) == 8) blas::check(rlhip_symmetrize_f64(q.ctx(), (char)uplo, dim, (const double*)A_buff, lda, (double*)full, dim), "symmetrize");
        else blas::check(rlhip_symmetrize_f32(q.ctx(), (char)uplo, dim, (const float*)A_buff, lda, (float*)full, dim), "symmetrize");
    }
};

/// A + mu_i I for one or several regularisation parameters (rl_sym_linops.hh:134-233): the stored UPPER triangle of a column-major A is
/// symmetrised once into HBM (every product is then an MFMA GEMM); with `set_eval_includes_reg(true)` column i of the result also receives
/// alpha * regs[min(i, num_ops - 1)] * B[:, i] (one device axpy per column, as the reference's loop).  regs is a HOST array (copied).
template <typename T>
struct RegExplicitSymLinOp {
    using scalar_t = T;
    const int64_t m;
    const int64_t dim;
    const int64_t n_rows;
    const int64_t n_cols;
    const T* A_buff;
    const int64_t lda;
    int64_t num_ops = 1;
    std::vector<T> regs;
    bool _eval_includes_reg = false;
    static constexpr Uplo uplo = Uplo::Upper;
    static constexpr Layout buff_layout = Layout::ColMajor;
    blas::Queue& q;

    RegExplicitSymLinOp(int64_t d, const T* A, int64_t ld, const T* arg_regs, int64_t arg_num_ops, blas::Queue& queue)
        : m(d), dim(d), n_rows(d), n_cols(d), A_buff(A), lda(ld), q(queue), sym_(d, Uplo::Upper, A, ld, Layout::ColMajor, queue) {
        randlapack_require(lda >= dim) << "lda=" << lda << " < dim=" << dim << " (lda must be >= operator dimension)";                 // :171
        num_ops = std::max<int64_t>(arg_num_ops, 1);                                                                                     // :173-174
        regs.assign((size_t)num_ops, T(0));
        for (int64_t i = 0; i < arg_num_ops; ++i) regs[(size_t)i] = arg_regs[i];
    }
    RegExplicitSymLinOp(int64_t d, const T* A, int64_t ld, std::vector<T>& arg_regs, blas::Queue& queue)
        : RegExplicitSymLinOp(d, A, ld, arg_regs.data(), (int64_t)arg_regs.size(), queue) {}
    void set_eval_includes_reg(bool eir) { _eval_includes_reg = eir; }

    /// C (dim x n) = alpha * (A [+ mu_i I]) * B + beta * C                                                                 (:200-217)
    void operator()(Layout layout, int64_t n, T alpha, const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {
        randlapack_require(layout == buff_layout) << "operation layout must match the operator storage layout (buff_layout)";
        randlapack_require(ldb >= dim) << "ldb=" << ldb << " < dim=" << dim << " (ldb must be >= operator dimension)";
        randlapack_require(ldc >= dim) << "ldc=" << ldc << " < dim=" << dim << " (ldc must be >= operator dimension)";
        sym_(layout, n, alpha, B, ldb, beta, C, ldc);
        if (_eval_includes_reg) {
            if (num_ops != 1) { randlapack_require(n == num_ops) << "with num_ops>1, n=" << n << " must equal num_ops=" << num_ops << " so each column gets its own regularization"; }
            for (int64_t i = 0; i < n; ++i) {
                const T coeff = alpha * regs[(size_t)std::min(i, num_ops - 1)];
                axpby_(dim, coeff, B + i * ldb, T(1), C + i * ldc);
            }
        }
    }

private:
    ExplicitSymLinOp<T> sym_;
    void axpby_(int64_t n, double a, const double* x, double b, double* y) { blas::check(rlhip_axpby_f64(q.ctx(), n, a, x, b, y), "axpby"); }
    void axpby_(int64_t n, float a, const float* x, float b, float* y) { blas::check(rlhip_axpby_f32(q.ctx(), n, a, x, b, y), "axpby"); }
};

}  // namespace linops

namespace detail {
inline void axpby(int64_t n, double a, const double* x, double b, double* y, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_axpby_f64(q.ctx(), n, a, x, b, y), "axpby"); }
inline void axpby(int64_t n, float a, const float* x, float b, float* y, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_axpby_f32(q.ctx(), n, a, x, b, y), "axpby"); }
inline void scal_cols_dev(int64_t m, int64_t n, double* A, int64_t lda, const double* s, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_scal_cols_f64(q.ctx(), m, n, A, lda, s), "scal_cols"); }
inline void scal_cols_dev(int64_t m, int64_t n, float* A, int64_t lda, const float* s, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_scal_cols_f32(q.ctx(), m, n, A, lda, s), "scal_cols"); }
}  // namespace detail

// ------------------------------------------------------------------------------------------------ SYPS
template <typename T, typename RNG>
class SYPS {
public:
    using scalar_t = T;
    using RNG_t = RNG;
    blas::Queue& q;
    int64_t passes_over_data;
    int64_t passes_per_stab;
    bool verbose;
    bool cond_check;
    std::vector<T> cond_nums;

    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    SYPS(int64_t p, int64_t q_, bool verb, bool cond) : SYPS(blas::default_queue(), p, q_, verb, cond) {}
    SYPS(blas::Queue& queue, int64_t p, int64_t q_, bool verb, bool cond) : q(queue), passes_over_data(p), passes_per_stab(q_), verbose(verb), cond_check(cond) {}

    /// skop_buff (m x k, DEVICE; allocated here when null, caller frees) <- the power sketch; work_buff: m x k DEVICE scratch or null.
    int call(Uplo uplo, int64_t m, const T* A, int64_t lda, int64_t k, RandBLAS::RNGState<RNG>& state, T*& skop_buff, T* work_buff) {   // :67-79
        linops::ExplicitSymLinOp<T> A_linop(m, uplo, A, lda, Layout::ColMajor, q);
        return call(A_linop, k, state, skop_buff, work_buff);
    }
    template <typename SLO>
    int call(SLO& A, int64_t k, RandBLAS::RNGState<RNG>& state, T*& skop_buff, T* work_buff) {                                           // :81-140
        randlapack_require(passes_per_stab >= 1) << "passes_per_stab=" << passes_per_stab << " must be >= 1";
        const int64_t m = A.dim, p = passes_over_data, qq = passes_per_stab;
        if (!skop_buff) skop_buff = blas::device_malloc<T>(m * k, q);
        RandBLAS::DenseDist D(m, k);
        state = RandBLAS::fill_dense(D, skop_buff, state, q);
        blas::Scratch ws(q);
        if (!work_buff) work_buff = ws.alloc<T>(m * k);
        T* tau = ws.alloc<T>(k);
        T* symm_out = work_buff;
        T* symm_in = skop_buff;
        for (int64_t p_done = 0; p_done < p;) {
            A(Layout::ColMajor, k, (T)1, symm_in, m, (T)0, symm_out, m);
            ++p_done;
            if (p_done % qq == 0) {
                if (lapack::geqrf(m, k, symm_out, m, tau, q)) throw std::runtime_error("GEQRF failed.");
                lapack::ungqr(m, k, k, symm_out, m, tau, q);
            }
            symm_out = (p_done % 2 == 1) ? skop_buff : work_buff;
            symm_in = (p_done % 2 == 1) ? work_buff : skop_buff;
        }
        if (p % 2 == 1) blas::device_copy_vector(m * k, work_buff, skop_buff, q);
        return 0;
    }
};

// ------------------------------------------------------------------------------------------------ SYRF
template <typename SYPS_t, typename Orth_t>
class SYRF {
public:
    using T = typename SYPS_t::scalar_t;
    using RNG = typename SYPS_t::RNG_t;
    SYPS_t& syps;
    Orth_t& orth;
    bool verbose;
    bool cond_check;
    std::vector<T> cond_nums;

    SYRF(SYPS_t& syps_obj, Orth_t& orth_obj, bool verb = false, bool cond = false) : syps(syps_obj), orth(orth_obj), verbose(verb), cond_check(cond) {}

    /// Q (m x k, DEVICE, caller-provided) <- orthonormal basis of A * S.  work_buff: m x k DEVICE scratch or null.        (rl_syrf.hh:44-92)
    int call(Uplo uplo, int64_t m, const T* A, int64_t k, T* Q, RandBLAS::RNGState<RNG>& state, T* work_buff) {
        linops::ExplicitSymLinOp<T> A_linop(m, uplo, A, m, Layout::ColMajor, syps.q);
        return call(A_linop, k, Q, state, work_buff);
    }
    template <typename SLO>
    int call(SLO& A, int64_t k, T* Q, RandBLAS::RNGState<RNG>& state, T* work_buff) {
        blas::Queue& q = syps.q;
        const int64_t m = A.dim;
        blas::Scratch ws(q);
        if (!work_buff) work_buff = ws.alloc<T>(m * k);
        T* sk = work_buff;
        syps.call(A, k, state, sk, Q);                                   // sketch lands in work_buff, Q is SYPS's scratch
        A(Layout::ColMajor, k, (T)1, work_buff, m, (T)0, Q, m);
        if (cond_check) cond_nums.push_back(util::cond_num_check(m, k, Q, verbose, q));
        if (orth.call(m, k, Q)) throw std::runtime_error("Orthogonalization failed.");
        return 0;
    }
};

// ------------------------------------------------------------------------------------------------ REVD2
/// p steps of the power method on (A - V diag(eigvals) V^T): returns the Rayleigh-quotient estimate of its dominant eigenvalue.
/// vector_buf: 4 m DEVICE entries, the first m holding the start vector; Mat_buf: m x k DEVICE scratch; eigvals: k DEVICE.   (:34-63)
template <typename T, typename SLO>
T power_error_est(SLO& A, int64_t k, int p, T* vector_buf, T* V, T* Mat_buf, const T* eigvals_dev, blas::Queue& q = blas::default_queue()) {
    const int64_t m = A.dim;
    T err = 0;
    blas::Scratch ws(q);
    T* dot_dev = ws.alloc<T>(1);
    for (int it = 0; it < p; ++it) {
        const T g_norm = lapack::lange(Norm::Fro, m, 1, vector_buf, m, q);
        detail::axpby(m, (T)0, vector_buf, (T)1 / g_norm, vector_buf, q);                                            // scal
        blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, k, 1, m, (T)1, V, m, vector_buf, m, (T)0, vector_buf + m, m, q);   // V^T g
        lapack::lacpy(MatrixType::General, m, k, V, m, Mat_buf, m, q);
        detail::scal_cols_dev(m, k, Mat_buf, m, eigvals_dev, q);                                                        // V diag(eigvals)
        blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, 1, k, (T)1, Mat_buf, m, vector_buf + m, m, (T)0, vector_buf + 2 * m, m, q);
        A(Layout::ColMajor, 1, (T)1, vector_buf, m, (T)0, vector_buf + 3 * m, m);
        detail::axpby(m, (T)-1, vector_buf + 2 * m, (T)1, vector_buf + 3 * m, q);                                      // (A - V E V^T) g
        blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, 1, 1, m, (T)1, vector_buf, m, vector_buf + 3 * m, m, (T)0, dot_dev, 1, q);
        blas::copy_to_host(1, dot_dev, &err, q);
        blas::device_copy_vector(m, vector_buf + 3 * m, vector_buf, q);
    }
    return err;
}

template <typename SYRF_t>
class REVD2 {
public:
    using T = typename SYRF_t::T;
    using RNG = typename SYRF_t::RNG;
    SYRF_t& syrf;
    int error_est_p;
    bool verbose;
    T last_err = 0;      // the error estimate the rank loop stopped on (diagnostic, not in the reference)

    REVD2(SYRF_t& syrf_obj, int error_est_power_iters, bool verb = false) : syrf(syrf_obj), error_est_p(error_est_power_iters), verbose(verb) {}

    /// A ~ V diag(eigvals) V^T.  k: in = starting rank, out = rank used.  V (m x k) and eigvals (k): DEVICE, (re)allocated here,
    /// freed by the caller with blas::device_free.                                                                      (:96-118)
    int call(Uplo uplo, int64_t m, const T* A, int64_t& k, T tol, T*& V, T*& eigvals, RandBLAS::RNGState<RNG>& state) {
        randlapack_require(m >= 0) << "m=" << m << " must be >= 0";
        randlapack_require(k > 0) << "target rank k=" << k << " must be > 0";
        randlapack_require(tol >= (T)0) << "tol=" << tol << " must be >= 0";
        randlapack_require(!(A == nullptr && m > 0)) << "A buffer is null but m=" << m << " > 0";
        linops::ExplicitSymLinOp<T> A_linop(m, uplo, A, m, Layout::ColMajor, syrf.syps.q);
        return call(A_linop, k, tol, V, eigvals, state);
    }
    template <typename SLO>
    int call(SLO& A, int64_t& k, T tol, T*& V, T*& eigvals, RandBLAS::RNGState<RNG>& state) {                              // :120-243
        randlapack_require(k > 0) << "target rank k=" << k << " must be > 0";
        randlapack_require(tol >= (T)0) << "tol=" << tol << " must be >= 0";
        blas::Queue& q = syrf.syps.q;
        const int64_t m = A.dim;
        randlapack_require(k <= m) << "target rank k=" << k << " exceeds the dimension m=" << m;
        RandBLAS::RNGState<RNG> error_est_state = state;
        if (++error_est_state.key[0] == 0) ++error_est_state.key[1];                                                       // key.incr(1), :135
        for (;;) {
            if (V) blas::device_free(V, q);
            if (eigvals) blas::device_free(eigvals, q);
            V = blas::device_malloc<T>(m * k, q);
            eigvals = blas::device_malloc<T>(k, q);
            blas::Scratch ws(q);
            T* Y = ws.alloc<T>(m * k);
            T* Omega = ws.alloc<T>(std::max<int64_t>(m * k, 4 * m));
            T* R = ws.alloc<T>(k * k);
            T* S = ws.alloc<T>(k);
            T* symrf_work = ws.alloc<T>(m * k);

            syrf.call(A, k, Omega, state, symrf_work);                                                                     // :147
            A(Layout::ColMajor, k, (T)1, Omega, m, (T)0, Y, m);                                                            // :150
            const T nu = std::numeric_limits<T>::epsilon() * lapack::lange(Norm::Fro, m, k, Y, m, q);                       // :153
            // R = chol(Omega^T Y + nu Omega^T Omega): the regularised core                                                  :159-170
            lapack::laset(MatrixType::General, k, k, (T)0, (T)0, R, k, q);
            blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, k, m, nu, Omega, m, (T)0, R, k, q);
            {
                T* Rt = ws.alloc<T>(k * k);                              // mirror the upper triangle: the gemm below adds to a full matrix
                util::transposition(k, k, R, k, Rt, k, 0, q);
                if (k > 1) lapack::lacpy(MatrixType::Lower, k - 1, k - 1, Rt + 1, k, R + 1, k, q);
            }
            blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, k, k, m, (T)1, Omega, m, Y, m, (T)1, R, k, q);
            if (lapack::potrf(Uplo::Upper, k, R, k, q)) throw std::runtime_error("Cholesky decomposition failed.");
            util::get_U(k, k, R, k, q);
            blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, k, (T)1, R, k, Y, m, q);   // B = Y R^-1
            lapack::gesdd(Job::SomeVec, m, k, Y, m, S, V, m, R, k, q);                                                       // :176

            std::vector<T> s_host((size_t)k), ev((size_t)k);
            blas::copy_to_host(k, S, s_host.data(), q);
            int64_t r = 0;
            for (int64_t i = 0; i < k; ++i) {                                                                               // :180-192
                ev[(size_t)i] = s_host[(size_t)i] * s_host[(size_t)i];
                if (ev[(size_t)i] > nu) ++r;
            }
            for (int64_t i = 0; i < r; ++i)
                if (!(ev[(size_t)i] - nu < 0)) ev[(size_t)i] -= nu;
            blas::copy_to_device(k, ev.data(), eigvals, q);
            if (r < k) lapack::laset(MatrixType::General, m, k - r, (T)0, (T)0, V + m * r, m, q);                            // :194

            RandBLAS::DenseDist g(m, 1);
            error_est_state = RandBLAS::fill_dense(g, Omega, error_est_state, q);                                           // :198-199
            const T err = power_error_est(A, k, error_est_p, Omega, V, Y, eigvals, q);
            last_err = err;
            if (err <= 5 * std::max(tol, nu) || k == m) break;                                                              // :203-209
            else if (2 * k > m) k = m;
            else k = 2 * k;
        }
        return 0;
    }
};

}  // namespace RandLAPACK
