// ABRIK (reference: RandLAPACK/drivers/rl_abrik.hh:30-768): randomized block Krylov iteration for a truncated SVD of any
// linear operator (BASELINE config 5).  Same object shape as the reference; buffers are DEVICE buffers and the
// operator is a device linop (rl_linops.hh).  The reference grows X_ev / Y_od / R / S with realloc after every block;
// here they live in `Grow` (amortised doubling, contents preserved, new columns zeroed) so that an unbounded
// max_krylov_iters (the default INT_MAX) does not force an n x n allocation up front.
//   qr_exp = geqrf_ungqr  -> device geqrf + ungqr;   qr_exp = cqrrt -> CQRRT panels (rl_cqrrt.hh).
// Row-sharded operator (one process per GPU, linop.row_sharded): X_ev / U are sharded by rows, everything n-long or k-sized is
// replicated; exchanges: A^T X (n x k, inside the linop), the two re-orthogonalisation inner products of the X side, and
// CQRRT's sketch + Gram all-reduces.  Both qr_exp run sharded: cqrrt as it is, geqrf_ungqr (the reference's default) as the sharded
// Cholesky-QR panel + the Householder sign vector of its gathered top block (sharded_householder_qr below).
#pragma once
#include <chrono>
#include <climits>
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
#include "rl_cqrrt.hh"

namespace RandLAPACK {

struct ABRIKSubroutines {
    enum QR_explicit { geqrf_ungqr, cqrrt };
};

template <typename T, typename RNG = RandBLAS::DefaultRNG>
class ABRIK {
    // column-growing device matrix with a fixed leading dimension (realloc semantics of the reference, :371-375,461-484)
    struct Grow {
        blas::Queue& q; int64_t ld; int64_t cap = 0; T* p = nullptr;
        Grow(blas::Queue& queue, int64_t ld_, int64_t cols) : q(queue), ld(ld_) { ensure(cols); }
        ~Grow() { if (p) blas::device_free(p, q); }
        void ensure(int64_t cols) {
            if (cols <= cap) return;
            const int64_t ncap = std::max<int64_t>(cols, 2 * cap);
            T* np_ = blas::device_malloc<T>(ld * ncap, q);
            if (cap > 0) blas::device_copy_vector(ld * cap, p, np_, q);
            blas::device_memset(np_ + ld * cap, 0, ld * (ncap - cap), q);
            if (p) { q.sync(); blas::device_free(p, q); }
            p = np_; cap = ncap;
        }
    };

public:
    using Subroutines = ABRIKSubroutines;

    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    ABRIK(bool verb, bool time_subroutines, T ep) : ABRIK(blas::default_queue(), verb, time_subroutines, ep) {}                       // rl_abrik.hh:64
    ABRIK(blas::Queue& queue, bool verb, bool time_subroutines, T ep) : q(queue) {                               // :64-77
        qr_exp = Subroutines::QR_explicit::geqrf_ungqr;
        verbose = verb;
        timing = time_subroutines;
        tol = ep;
        max_krylov_iters = INT_MAX;
        num_krylov_iters = 0;
        norm_R_end = 0;
        singular_triplets_found = 0;
    }

    /// ABRIK on a general dense matrix given by its pointer (rl_abrik.hh:122-143): A (m x n, lda) is a DEVICE buffer and is not modified.
    int call(int64_t m, int64_t n, T* A, int64_t lda, int64_t k, T*& U, T*& V, T*& Sigma, RandBLAS::RNGState<RNG>& state) {
        randlapack_require(m >= 0) << "m=" << m << " must be >= 0";
        randlapack_require(n >= 0) << "n=" << n << " must be >= 0";
        randlapack_require(lda >= m) << "lda=" << lda << " < m=" << m << " (lda must be >= m for ColMajor)";
        randlapack_require(k > 0) << "target rank k=" << k << " must be > 0";
        randlapack_require(!(A == nullptr && m > 0 && n > 0)) << "A buffer is null but m=" << m << " and n=" << n << " imply a nonempty matrix";
        linops::DenseLinOp<T> A_linop(m, n, A, lda, Layout::ColMajor, q);
        A_linop.row_sharded = q.world() > 1;          // on a sharded queue the pointer is this rank's row block
        return this->call(A_linop, k, U, V, Sigma, state);
    }

    /// ABRIK on a sparse matrix (rl_abrik.hh:146-162): SpMat is RandBLAS::sparse_data::CSRMatrix over device arrays (rl_randblas.hh).
    template <typename SpMat, typename = typename SpMat::index_t>
    int call(int64_t m, int64_t n, SpMat& A, int64_t k, T*& U, T*& V, T*& Sigma, RandBLAS::RNGState<RNG>& state) {
        randlapack_require(m >= 0) << "m=" << m << " must be >= 0";
        randlapack_require(n >= 0) << "n=" << n << " must be >= 0";
        randlapack_require(k > 0) << "target rank k=" << k << " must be > 0";
        randlapack_require(A.n_rows == m && A.n_cols == n) << "sparse matrix is " << A.n_rows << " x " << A.n_cols << ", call says " << m << " x " << n;
        linops::SparseLinOp<T> A_linop(m, n, A.nnz, A.rowptr, A.colidxs, A.vals, q);
        A_linop.row_sharded = q.world() > 1;
        return this->call(A_linop, k, U, V, Sigma, state);
    }

    /// A: linear operator (rl_linops.hh).  U (m x triplets), V (n x triplets), Sigma (triplets): allocated HERE on the
    /// device, owned by the caller afterwards (blas::device_free), like the reference's new[] (:678-680).  Returns 0.
    template <typename GLO>
    int call(GLO& A, int64_t k, T*& U, T*& V, T*& Sigma, RandBLAS::RNGState<RNG>& state) {
        randlapack_require(k > 0) << "target rank k=" << k << " must be > 0";                                   // :176
        const bool use_cqrrt = (qr_exp == Subroutines::QR_explicit::cqrrt);
        const bool sharded = q.world() > 1;       // the operator's rows (and every m-long object: X_ev, U) are sharded; n-long ones are replicated
        RandLAPACK::CQRRT<T, RNG> cqrrt(q, false, tol);                                                        // :281-285
        cqrrt.nnz = 2;
        const T d_factor = (T)1.25;
        T* R_11_trans = nullptr;
        // Row-sharded operator with the reference's DEFAULT panels (qr_exp = geqrf_ungqr, :333-342 / :552-570): the m-long panels X are
        // sharded by rows.  What geqrf + ungqr leave -- the orthonormal factor in LAPACK's Householder sign convention and its triangle -- is,
        // by the reconstruction contract (lapack::orhr_col), any orthonormal factor times diag(D), D = the sign vector of the sign-modified LU
        // of its TOP k x k block.  So: the sharded sketch-preconditioned Cholesky-QR panel (CQRRT, which row-shards: one sketch and one Gram
        // exchange) on a PRIVATE random state (the reference's geqrf path draws nothing), the top block gathered from its owners by one k x k
        // exchange, D on every rank, Q <- Q D, R <- D R.  Same Q and R as the single-device call to rounding; the n-long panels Y are
        // replicated and take the ordinary single-rank route.
        RandBLAS::RNGState<RNG> panel_state(0x5eedu);
        int64_t m_glob_x = 0, row0_x = 0;
        if (sharded && !use_cqrrt) q.shard_extent(A.n_rows, m_glob_x, row0_x);
        auto sharded_householder_qr = [&](int64_t rows, T* P, T* Rout, int64_t ldr) {
            cqrrt.rows_replicated = false;
            lapack::laset(MatrixType::General, k, k, (T)0, (T)0, Rout, ldr, q);      // CQRRT fills the triangle and multiplies the full k x k array (:190: the cqrrt route zeroes its buffer too)
            const int rc = cqrrt.call(rows, k, P, rows, Rout, ldr, d_factor, panel_state);
            randlapack_require(rc == 0) << "ABRIK: sharded panel factorization failed (code " << rc << ")";
            blas::Scratch w5(q);
            T* Qt = w5.alloc<T>(k * k);
            T* Tm = w5.alloc<T>(k * k);
            T* Dv = w5.alloc<T>(k);
            lapack::laset(MatrixType::General, k, k, (T)0, (T)0, Qt, k, q);
            const int64_t cnt = (row0_x < k) ? std::min<int64_t>(rows, k - row0_x) : 0;      // my rows among the global rows [0, k)
            if (cnt > 0) lapack::lacpy(MatrixType::General, cnt, k, P, rows, Qt + row0_x, k, q);
            q.allreduce_sum(Qt, k * k);
            {
                blas::LocalOnly one_rank(q);
                lapack::orhr_col(k, k, k, Qt, k, Tm, k, Dv, q);
            }
            lapack::row_sign(k, Rout, ldr, Dv, q);
            if (rows > 0) {
                if constexpr (std::is_same<T, double>::value) blas::check(rlhip_scal_cols_f64(q.ctx(), rows, k, P, rows, Dv), "scal_cols");
                else blas::check(rlhip_scal_cols_f32(q.ctx(), rows, k, P, rows, Dv), "scal_cols");
            }
        };
        // explicit QR of a panel P (rows x k, ld rows): Q in place, R (k x k upper) to Rout (ld ldr)
        auto panel_qr = [&](int64_t rows, T* P, T* Rout, int64_t ldr, bool m_long, RandBLAS::RNGState<RNG>& st) {
            if (use_cqrrt) {
                cqrrt.rows_replicated = !m_long;
                const int rc = cqrrt.call(rows, k, P, rows, Rout, ldr, d_factor, st);
                randlapack_require(rc == 0) << "ABRIK: CQRRT panel factorization failed (code " << rc << ")";
            }
        };
        auto reduce_m = [&](T* buf, int64_t rows_, int64_t cols_, int64_t ld_) {      // sum an inner product over the row shards
            if (!sharded) return;
            if (ld_ == rows_) { q.allreduce_sum(buf, rows_ * cols_); return; }
            blas::Scratch w3(q);
            T* tmp = w3.alloc<T>(rows_ * cols_);
            lapack::lacpy(MatrixType::General, rows_, cols_, buf, ld_, tmp, rows_, q);
            q.allreduce_sum(tmp, rows_ * cols_);
            lapack::lacpy(MatrixType::General, rows_, cols_, tmp, rows_, buf, ld_, q);
        };
        // subroutine timers of the reference (:176-212): every lap drains the stream first, so they are only armed when `timing` is set
        using clk = std::chrono::steady_clock;
        long allocation_t = 0, get_factors_t = 0, ungqr_t = 0, reorth_t = 0, qr_t = 0, gemm_A_t = 0, main_loop_t = 0, sketching_t = 0, r_cpy_t = 0,
             s_cpy_t = 0, norm_t = 0;
        clk::time_point lap_t0, total_t0, loop_t0;
        auto tic = [&]() { if (timing) { q.sync(); lap_t0 = clk::now(); } };
        auto toc = [&](long& acc) { if (timing) { q.sync(); acc += (long)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - lap_t0).count(); } };
        if (timing) { q.sync(); total_t0 = clk::now(); }
        tic();
        const int64_t m = A.n_rows, n = A.n_cols;
        int64_t iter = 0, iter_od = 0, iter_ev = 0, end_rows = 0, end_cols = 0;
        T norm_R = 0;
        const int max_iters = max_krylov_iters;
        Grow Y_od(q, n, k), X_ev(q, m, k), R(q, n, k), S(q, n + k, k);
        blas::Scratch ws(q);
        T* Y_orth_buf = nullptr;          // k x (iter_ev k), ld k      -- sized on demand (reference: k x n up front, :247)
        T* X_orth_buf = nullptr;          // (iter_od k) x k, ld n + k  -- idem (:248)
        T* tau = ws.alloc<T>(k);
        T* R_qr = ws.alloc<T>(k * k);            // the triangle of a fused geqrf + ungqr (lapack::geqrf_q)
        if (use_cqrrt) R_11_trans = ws.alloc<T>(k * k);
        int64_t curr_Y_cols = k, curr_X_cols = k;
        int64_t Y_i = 0, X_i = 0, R_i = -1, R_ii = 0, S_i = 0, S_ii = k;      // element offsets (the reference's moving pointers)
        const T norm_A = A.fro_nrm();                                                                           // :272
        const T sq_tol = tol * tol;
        const T threshold = std::sqrt(1 - sq_tol) * norm_A;
        const T sqrt_eps = std::sqrt(std::numeric_limits<T>::epsilon());
        auto elem = [&](const T* p) { T v; blas::copy_to_host(1, p, &v, q); return v; };
        toc(allocation_t);

        tic();
        RandBLAS::DenseDist D(n, k);                                                                            // :298-299
        state = RandBLAS::fill_dense(D, Y_od.p + Y_i, state, q);
        toc(sketching_t);
        tic();
        A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, k, n, (T)1.0, Y_od.p + Y_i, n, (T)0.0, X_ev.p + X_i, m);   // :311
        toc(gemm_A_t);
        if (use_cqrrt) {                                                                                        // :318-319
            tic();
            panel_qr(m, X_ev.p + X_i, R_11_trans, k, true, state);
            toc(qr_t);
        } else {
            tic();
            if (sharded) sharded_householder_qr(m, X_ev.p + X_i, R_qr, k);
            const bool fused = sharded || (fuse_geqrf_ungqr && lapack::geqrf_q(m, k, X_ev.p + X_i, m, R_qr, k, q));   // :333 + :342 in one pass
            if (!fused) lapack::geqrf(m, k, X_ev.p + X_i, m, tau, q);                                           // :333
            toc(qr_t);
            tic();
            if (!fused) lapack::ungqr(m, k, k, X_ev.p + X_i, m, tau, q);                                        // :342
            toc(ungqr_t);
        }
        ++iter_od;
        ++iter;
        if (timing) { q.sync(); loop_t0 = clk::now(); }
        while (1) {
            if (iter % 2 != 0) {
                tic();
                A(Side::Left, Layout::ColMajor, Op::Trans, Op::NoTrans, n, k, m, (T)1.0, X_ev.p + X_i, m, (T)0.0, Y_od.p + Y_i, n);   // :364
                toc(gemm_A_t);
                tic();
                curr_X_cols += k;                                                                               // :371-375
                X_ev.ensure(curr_X_cols);
                X_i = m * (curr_X_cols - k);
                toc(allocation_t);
                tic();
                if (iter != 1) {                                                                                // :384-394
                    blas::Scratch w2(q);
                    Y_orth_buf = w2.alloc<T>(k * iter_ev * k);
                    blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, k, iter_ev * k, n, (T)1.0, Y_od.p + Y_i, n, Y_od.p, n, (T)0.0, R.p + R_i, n, q);
                    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, n, k, iter_ev * k, (T)-1.0, Y_od.p, n, R.p + R_i, n, (T)1.0, Y_od.p + Y_i, n, q);
                    blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, k, iter_ev * k, n, (T)1.0, Y_od.p + Y_i, n, Y_od.p, n, (T)0.0, Y_orth_buf, k, q);
                    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, n, k, iter_ev * k, (T)-1.0, Y_od.p, n, Y_orth_buf, k, (T)1.0, Y_od.p + Y_i, n, q);
                }
                toc(reorth_t);
                if (use_cqrrt) {                                                                                // :402-410
                    tic();
                    lapack::laset(MatrixType::General, k, k, (T)0, (T)0, R_11_trans, k, q);
                    panel_qr(n, Y_od.p + Y_i, R_11_trans, k, false, state);
                    toc(qr_t);
                    tic();
                    util::transposition(k, k, R_11_trans, k, R.p + R_ii, n, 1, q);
                    toc(r_cpy_t);
                } else {
                    tic();
                    const bool fused = fuse_geqrf_ungqr && lapack::geqrf_q(n, k, Y_od.p + Y_i, n, R_qr, k, q);  // :420 + :444 in one pass
                    if (!fused) lapack::geqrf(n, k, Y_od.p + Y_i, n, tau, q);                                   // :420
                    toc(qr_t);
                    tic();
                    if (fused) util::transposition(k, k, R_qr, k, R.p + R_ii, n, 1, q);
                    else util::transposition(k, k, Y_od.p + Y_i, n, R.p + R_ii, n, 1, q);                       // :432 (upper triangle, transposed)
                    toc(r_cpy_t);
                    tic();
                    if (!fused) lapack::ungqr(n, k, k, Y_od.p + Y_i, n, tau, q);                                // :444
                    toc(ungqr_t);
                }
                if (std::abs(elem(R.p + R_ii + (n + 1) * (k - 1))) < sqrt_eps) break;                           // :455-458
                tic();
                R.ensure(curr_X_cols);                                                                          // :461-484
                R_i = (iter_ev + 1) * k;
                R_ii = (n * k * (iter_ev + 1)) + k + (k * iter_ev);
                toc(allocation_t);
                ++iter_ev;
            } else {
                tic();
                A(Side::Left, Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, k, n, (T)1.0, Y_od.p + Y_i, n, (T)0.0, X_ev.p + X_i, m);   // :494
                toc(gemm_A_t);
                tic();
                curr_Y_cols += k;                                                                               // :501-505
                Y_od.ensure(curr_Y_cols);
                Y_i = n * (curr_Y_cols - k);
                toc(allocation_t);
                tic();
                {                                                                                               // :515-522
                    blas::Scratch w2(q);
                    const int64_t ldx = iter_od * k;
                    X_orth_buf = w2.alloc<T>(ldx * k);
                    blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, iter_od * k, k, m, (T)1.0, X_ev.p, m, X_ev.p + X_i, m, (T)0.0, S.p + S_i, n + k, q);
                    reduce_m(S.p + S_i, iter_od * k, k, n + k);
                    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, k, iter_od * k, (T)-1.0, X_ev.p, m, S.p + S_i, n + k, (T)1.0, X_ev.p + X_i, m, q);
                    blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, iter_od * k, k, m, (T)1.0, X_ev.p, m, X_ev.p + X_i, m, (T)0.0, X_orth_buf, ldx, q);
                    reduce_m(X_orth_buf, iter_od * k, k, ldx);
                    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, k, iter_od * k, (T)-1.0, X_ev.p, m, X_orth_buf, ldx, (T)1.0, X_ev.p + X_i, m, q);
                }
                toc(reorth_t);
                if (use_cqrrt) {                                                                                // :530-531
                    tic();
                    panel_qr(m, X_ev.p + X_i, S.p + S_ii, n + k, true, state);
                    toc(qr_t);
                } else {
                    tic();
                    if (sharded) sharded_householder_qr(m, X_ev.p + X_i, R_qr, k);
                    const bool fused = sharded || (fuse_geqrf_ungqr && lapack::geqrf_q(m, k, X_ev.p + X_i, m, R_qr, k, q));  // :552 + :570 in one pass
                    if (!fused) lapack::geqrf(m, k, X_ev.p + X_i, m, tau, q);                                   // :552
                    toc(qr_t);
                    tic();
                    if (fused) lapack::lacpy(MatrixType::Upper, k, k, R_qr, k, S.p + S_ii, n + k, q);
                    else lapack::lacpy(MatrixType::Upper, k, k, X_ev.p + X_i, m, S.p + S_ii, n + k, q);         // :561
                    toc(s_cpy_t);
                    tic();
                    if (!fused) lapack::ungqr(m, k, k, X_ev.p + X_i, m, tau, q);                                // :570
                    toc(ungqr_t);
                }
                if (std::abs(elem(S.p + S_ii + ((n + k) + 1) * (k - 1))) < sqrt_eps) break;                     // :595-598
                tic();
                S.ensure(curr_Y_cols);                                                                          // :604-630
                S_i = (n + k) * k * iter_od;
                S_ii = (n + k) * k * iter_od + k + (iter_od * k);
                toc(allocation_t);
                ++iter_od;
            }
            tic();
            if (iter % 2 != 0) {                                                                                // :641-642 lantr(Fro, Upper)
                blas::Scratch w2(q);
                const int64_t nn = iter_ev * k;
                T* Tri = w2.alloc<T>(nn * nn);
                lapack::laset(MatrixType::General, nn, nn, (T)0, (T)0, Tri, nn, q);
                lapack::lacpy(MatrixType::Upper, nn, nn, R.p, n, Tri, nn, q);
                norm_R = lapack::lange(Norm::Fro, nn, nn, Tri, nn, q);
            }
            toc(norm_t);
            if (iter >= max_iters) break;                                                                       // :650-653
            ++iter;
            if (norm_R > threshold) break;                                                                      // :659-662
        }
        if (timing) { q.sync(); main_loop_t = (long)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - loop_t0).count(); }
        norm_R_end = norm_R;
        num_krylov_iters = (int)iter;
        end_cols = num_krylov_iters * k / 2;                                                                    // :668-669
        end_rows = (iter % 2 == 0) ? end_cols + k : end_cols;
        tic();
        T* U_hat = ws.alloc<T>(end_rows * end_cols);
        T* VT_hat = ws.alloc<T>(end_cols * end_cols);
        Sigma = blas::device_malloc<T>(std::min(end_cols, end_rows), q);                                        // :678-680
        U = blas::device_malloc<T>(m * end_cols, q);
        V = blas::device_malloc<T>(n * end_cols, q);
        toc(allocation_t);
        tic();
        if (iter % 2 != 0) lapack::gesdd(Job::SomeVec, end_rows, end_cols, R.p, n, Sigma, U_hat, end_rows, VT_hat, end_cols, q);       // :685
        else lapack::gesdd(Job::SomeVec, end_rows, end_cols, S.p, n + k, Sigma, U_hat, end_rows, VT_hat, end_cols, q);                 // :688
        blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, end_cols, end_rows, (T)1.0, X_ev.p, m, U_hat, end_rows, (T)0.0, U, m, q);   // :696
        blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, n, end_cols, end_cols, (T)1.0, Y_od.p, n, VT_hat, end_cols, (T)0.0, V, n, q);    // :698
        singular_triplets_found = end_cols;
        q.sync();
        toc(get_factors_t);
        if (timing) {                                                                                           // :730-734
            const long total_t = (long)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - total_t0).count();
            const long t_rest = total_t - (allocation_t + get_factors_t + ungqr_t + reorth_t + qr_t + gemm_A_t + sketching_t + r_cpy_t + s_cpy_t + norm_t);
            times = {allocation_t, get_factors_t, ungqr_t, reorth_t, qr_t, gemm_A_t, main_loop_t, sketching_t, r_cpy_t, s_cpy_t, norm_t, t_rest, total_t};
        }
        return 0;
    }

    blas::Queue& q;
    Subroutines::QR_explicit qr_exp;
    // (not in the reference) qr_exp = geqrf_ungqr: run each geqrf + ungqr pair (:333-342, :420-444, :552-570) as ONE pass over the panel
    // (lapack::geqrf_q: Cholesky-QR twice + the sign vector of the Householder reconstruction; same Q and R to rounding, the two calls when
    // the panel is not tall / well conditioned enough).  (A row-sharded operator runs CQRRT panels anyway.)
    bool fuse_geqrf_ungqr = true;
    bool verbose;
    bool timing;
    T tol;
    int num_krylov_iters;
    int max_krylov_iters;
    std::vector<long> times;
    T norm_R_end;
    int64_t singular_triplets_found;
};

}  // namespace RandLAPACK
