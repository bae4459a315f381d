// CQRRPTalg / CQRRPT (reference: RandLAPACK/drivers/rl_cqrrpt.hh:20-391): Cholesky QR with randomized
// pivoting for tall matrices.  A (m x n, device) -> Q in place, R (n x n), J (1-based pivots), rank.
#pragma once
#include <chrono>
#include <cmath>
#include <limits>
#include <vector>
#include <cstdlib>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"
#include "rl_util.hh"
#include "rl_hqrrp.hh"
#include "rl_bqrrp.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class CQRRPTalg {
public:
    virtual ~CQRRPTalg() {}
    virtual int call(int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr, int64_t* J, T d_factor,
                     RandBLAS::RNGState<RNG>& state) = 0;
};

struct CQRRPTSubroutines {
    enum QRCP { hqrrp, bqrrp, geqp3 };
};

template <typename T, typename RNG = RandBLAS::DefaultRNG>
class CQRRPT : public CQRRPTalg<T, RNG> {
public:
    using Subroutines = CQRRPTSubroutines;

    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    CQRRPT(bool time_subroutines, T ep) : CQRRPT(blas::default_queue(), time_subroutines, ep) {}                                         // rl_cqrrpt.hh:52-55
    CQRRPT(blas::Queue& queue, bool time_subroutines, T ep) : q(queue) {
        timing = time_subroutines;
        eps = ep;
        nnz = 2;                       // reference default (rl_cqrrpt.hh constructor)
        qrcp = Subroutines::QRCP::geqp3;
        bqrrp_block_ratio = 1;         // rl_cqrrpt.hh:59-63
        nb_alg = 64;
        oversampling = 10;
        use_cholqr = 0;
        panel_pivoting = 1;
        orthogonalization = false;
        rank = 0;
        fold_pivoting = true;
        split_qrcp = true;
        split_cols = 0;
    }

    /// A (m x n, lda, DEVICE) is overwritten by Q (first `rank` columns orthonormal), R (n x n, ldr, DEVICE) receives
    /// the rank x n upper-trapezoidal factor, J (n, DEVICE int64, 1-based) the pivots.  Return codes as the
    /// reference: 0 ok (also for an all-zero sketch, :256-261), 1 when R_sk has a zero on its diagonal (:296-301).
    /// SURVEY.md A.6 is the behavioural spec.  qrcp: geqp3 (default), hqrrp or bqrrp, as in the reference.
    int call(int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr, int64_t* J, T d_factor,
             RandBLAS::RNGState<RNG>& state) override {
        randlapack_require(m >= 0) << "m=" << m << " must be >= 0";                                      // :161-168
        randlapack_require(n >= 0) << "n=" << n << " must be >= 0";
        randlapack_require(lda >= m) << "lda=" << lda << " < m=" << m << " (lda must be >= m for ColMajor)";
        randlapack_require(ldr >= n) << "ldr=" << ldr << " < n=" << n << " (ldr must be >= n)";
        randlapack_require(d_factor >= (T)1.0) << "d_factor=" << d_factor << " must be >= 1.0";
        randlapack_require(!(A == nullptr && m > 0 && n > 0)) << "A buffer is null but m=" << m << " and n=" << n << " imply a nonempty matrix";
        randlapack_require(!(R == nullptr && n > 0)) << "R buffer is null but n=" << n << " > 0";
        randlapack_require(!(J == nullptr && n > 0)) << "J buffer is null but n=" << n << " > 0";
        using clk = std::chrono::steady_clock;
        auto stamp = [&]() { if (timing) q.sync(); return clk::now(); };
        auto t_total0 = stamp();
        blas::Phases ph;                 // profiler phases named like the reference's timers (rl_cqrrpt_gpu.hh:162-181): saso, qrcp, rank_reveal, a_mod_piv, a_mod_trsm, cholqr

        int64_t k = n;
        T* W_early = nullptr;            // split QRCP: the scratch copy of A_pre, allocated before the factorization of the sketch ...
        bool half_solved = false;        // ... whose left half is solved beside the second half of that factorization
        std::vector<T> diag_early;       // split QRCP: diag(R_sk) as read on the side queue
        auto ldw_of = [](int64_t rows) { return (rows % 512 == 0) ? rows + 32 : rows; };
        const int64_t d = (int64_t)(d_factor * n);                                                          // :198 (truncation)
        const T eps_initial_rank_estimation = 2 * std::pow(std::numeric_limits<T>::epsilon(), (T)0.95);   // :200
        int64_t new_rank;
        if (n == 0 || m == 0) { rank = 0; return 0; }

        randlapack_require(nnz >= 1 && nnz <= d) << "nnz=" << nnz << " nonzeros per column do not fit a sketch of d=" << d << " rows (RandBLAS::SparseDist requires vec_nnz <= d)";
        blas::Scratch ws(q);
        T* A_hat = ws.alloc<T>(d * n);
        T* tau = ws.alloc<T>(n);

        // ---- sketch: S = SparseSkOp(SparseDist(d, m, nnz), state); state = S.next_state; A_hat = S * A  (:214-222)
        auto t0 = stamp();
        ph("saso");
        {
            // Row-block sharding (one process per GPU): S is the operator for the GLOBAL row count; every rank applies its
            // column window of S to its rows and the d x n partial sketches are summed over the ranks.  Everything that
            // follows on the sketch (QRCP, rank estimate) is replicated, the m-long operations stay local, and the only
            // other exchange is the k x k Gram matrix of the CholQR step.
            int64_t m_glob = m, row0 = 0;
            q.shard_extent(m, m_glob, row0);
            RandBLAS::SparseDist DS(d, m_glob, nnz);
            RandBLAS::SparseSkOp<T, RNG> S(DS, state, q);
            state = S.next_state;
            if (sketch_override)   // parity-test hook: both sides factor the SAME precomputed sketch
                lapack::lacpy(MatrixType::General, d, n, sketch_override, d, A_hat, d, q);
            else if (q.world() > 1) {
                RandBLAS::sketch_rows(S, n, (T)1.0, A, lda, row0, m, (T)0.0, A_hat, d, q);
                q.allreduce_sum(A_hat, d * n);
            } else
                RandBLAS::sketch_general(Layout::ColMajor, Op::NoTrans, Op::NoTrans, d, n, m, (T)1.0, S, 0, 0, A, lda,
                                         (T)0.0, A_hat, d, q);
            if (sketch_export) lapack::lacpy(MatrixType::General, d, n, A_hat, d, sketch_export, d, q);
        }
        auto t1 = stamp();
        ph("qrcp");
        // ---- QRCP of the sketch (:247)
        if (qrcp == Subroutines::QRCP::hqrrp) {                                                             // :230-231
            blas::LocalOnly replicated(q);    // the sketch is replicated: every rank runs the single-device QRCP and gets the same pivots
            hqrrp(d, n, A_hat, d, J, tau, nb_alg, oversampling, panel_pivoting, use_cholqr, state, q);
        } else if (qrcp == Subroutines::QRCP::bqrrp) {                                                      // :232-245
            if (n <= 2000) bqrrp_block_ratio = 1.0;
            else if (n <= 8000) bqrrp_block_ratio = 0.5;
            else bqrrp_block_ratio = (T)1 / (T)32;
            blas::LocalOnly replicated(q);    // (as for hqrrp: a replicated sub-problem)
            RandLAPACK::BQRRP<T, RNG> bq(q, false, (int64_t)(n * bqrrp_block_ratio));
            bq.qrcp_wide = BQRRPSubroutines::QRCPWide::luqr;      // the reference object's defaults (rl_bqrrp.hh:101-103)
            bq.qr_tall = BQRRPSubroutines::QRTall::geqrf;
            bq.apply_trans_q = BQRRPSubroutines::ApplyTransQ::ormqr;
            bq.call(d, n, A_hat, d, (T)1.0, tau, J, state);
        } else {
            // Split QRCP (not in the reference; same factorization): after the first n / 2 steps of geqp3 the leading n / 2 rows of R_sk and
            // the leading n / 2 pivots are final, and the left half of A_pre = (A P) R_sk^-1 depends on nothing else -- so the first solve
            // starts on it while the second half of the sketch (geqp3 of the trailing block: a latency-bound kernel that keeps a quarter
            // of the CUs waiting on its exchanges) is factored on a side stream, packed into fewer workgroups.  C3: DESIGN 4.13.
            // Only where it pays and cannot change a result that tests pin bit for bit: one rank, untimed, tall inputs, geqp3.
            const int64_t h = n / 2;
            bool split = split_qrcp && fold_pivoting && !timing && q.world() == 1 && m >= ((int64_t)1 << 18) && n >= 512 && h % 256 == 0 && d > h;
            if (split) {
                W_early = ws.try_alloc<T>(ldw_of(m) * n);
                split = W_early != nullptr;
            }
            if (!split) lapack::geqp3(d, n, A_hat, d, J, tau, q);
            else {
                // (allocated BEFORE the solve below is enqueued: that call releases its own scratch -- the packed triangle its kernel is still
                //  reading -- when it returns, and the arena is ordered by the main stream only; the side stream writes J2 beside that kernel)
                rlhip_path_note(q.ctx(), 13, 1);
                int64_t* J2 = ws.alloc<int64_t>(n - h);
                T* R_lead = ws.alloc<T>(h * h);                                          // the leading h x h triangle of R_sk for the speculative half solve
                lapack::geqp3_steps(d, n, h, A_hat, d, J, tau, q);
                // the side queue is ordered HERE, behind the first half of the factorization and its copy-back and before the half solve is
                // enqueued: the second half (side queue) then runs beside the solve, and never beside the kernels that still write A_hat
                // (the ordering used to rest on a host wait inside the solve's call)
                blas::Queue side(q, blas::Queue::CachedSide{});
                side.wait_for(q);
                // The half solve is SPECULATIVE: it goes out right behind the first half of the factorization, without a look at the leading
                // diagonal of R_sk (a host round trip with the device idle).  A leading block that is singular or graded beyond the solve's
                // conditioning guard closes the gate of the launch (nothing is written, `half_solved` stays false); one that merely fails the
                // rank criterion below produces a left half nobody reads -- the rank decision then takes the reference's in-place order.
                // (the triangle is packed into SCRATCH, not into the caller's R: when the sketch turns out rank deficient below h, or all zero,
                //  rows of R that the one-piece order and the reference leave untouched stay untouched -- ADVICE r4)
                lapack::laset(MatrixType::General, h, h, (T)0, (T)0, R_lead, h, q);
                lapack::lacpy(MatrixType::Upper, h, h, A_hat, d, R_lead, h, q);
                half_solved = blas::trsm_gather_range(Diag::NonUnit, m, n, (T)1.0, R_lead, h, A, lda, J, W_early, ldw_of(m), 0, h, q);
                {
                    // the second half is packed into few, full workgroups (up to 16 columns each, ~100 KiB of LDS): it leaves the other CUs to
                    // the solve and loses little itself (768 x 512: 3.5 ms with 4 columns per workgroup, 4.9 with 16; C3: 71.3 / 70.3 / 69.6 ms
                    // with 4 / 8 / 16).  split_cols > 0 overrides.
                    int64_t cols = (100 * 1024) / ((d - h) * (int64_t)sizeof(T));
                    cols = cols < 4 ? 4 : (cols > 16 ? 16 : cols);
                    if (split_cols > 0) cols = split_cols;
                    side.set_qrcp_cols(half_solved ? (int)cols : 0);
                    lapack::geqp3(d - h, n - h, A_hat + h + h * d, d, J2, tau + h, side);
                    side.set_qrcp_cols(0);                                              // (the cached side queue goes back to its default)
                    // the diagonal of R_sk is final here (the column swap below moves entries of the finished rows right of column h only):
                    // read on the SIDE queue, the rank decision does not make the host wait for the half solve on the main stream -- whose
                    // successors (the swaps, the second half of the solve) are then enqueued behind it while it runs
                    diag_early.resize(n);
                    lapack::get_diag(n < d ? n : d, A_hat, d, diag_early.data(), side);
                    q.wait_for(side);
                    util::col_swap(h, n - h, n - h, A_hat + h * d, d, J2, q);          // the finished rows follow the trailing block's pivots
                    util::col_swap(n - h, n - h, &J[h], J2, q);                        // J[h:] <- J[h:][J2]
                }
            }
        }
        auto t2 = stamp();
        ph("rank_reveal");

        std::vector<T> diag(n);
        if (!diag_early.empty()) diag = diag_early;
        else lapack::get_diag(n < d ? n : d, A_hat, d, diag.data(), q);
        if (!diag[0]) { rank = 0; return 0; }                                                               // :256-261
        for (int64_t i = 0; i < n; ++i) {                                                                   // :267-272
            if (std::abs(diag[i]) / std::abs(diag[0]) < eps_initial_rank_estimation) { k = i; break; }
        }
        rank = k;
        new_rank = k;
        auto t3 = stamp();
        ph("a_mod_piv");

        lapack::lacpy(MatrixType::Upper, k, k, A_hat, d, R, ldr, q);                                        // :281
        // Full-rank sketches (the common case) take ONE pass over A for "permute, then precondition": the first solve reads the
        // pivoted columns of A directly and writes A_pre into a scratch copy W, the Gram matrix is formed from W, and the second solve
        // reads W and writes Q into A.  Same arithmetic per entry as col_swap + in-place trsm; saves the 2 x 8 m n bytes of the
        // separate permutation pass (C3: 3.9 ms of 79).  `fold_pivoting = false` (or a rank-deficient sketch) keeps the reference's
        // statement order below.
        // The scratch matrix has its own leading dimension: with ldw = m a column stride that is a multiple of 4 KiB (m = 2^20: 8 MiB)
        // puts the 384 column pieces a Gram tile reads per K-step on the same memory channels -- 20.7 against 18.7 ms for C3's Gram matrix.
        T* W = nullptr;
        const int64_t ldw = ldw_of(m);
        if (fold_pivoting && k == n && m >= 16384) W = W_early ? W_early : ws.try_alloc<T>(ldw * n);   // no room for a second m x n matrix: the in-place order below
        if (W) {
            auto t4 = stamp();
            ph("a_mod_trsm");
            for (int64_t i = 0; i < k; ++i)                                                                 // :296-301 diag_is_nonzero
                if (diag[i] == (T)0) { util::col_swap(m, n, k, A, lda, J, q); return 1; }                   // (A leaves permuted, as in the reference)
            // (split QRCP: the left half of W is solved already; the right half is the same launch with other bounds.  A right half outside
            //  the fused kernel's domain -- a badly conditioned diagonal block -- takes the whole solve again.)
            if (!(half_solved && blas::trsm_gather_range(Diag::NonUnit, m, n, (T)1.0, R, ldr, A, lda, J, W, ldw, n / 2, n, q)))
                blas::trsm_gather(Diag::NonUnit, m, k, (T)1.0, R, ldr, A, lda, J, W, ldw, q);               // :288 + :302
            auto t5 = stamp();
            ph("cholqr");
            blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, k, m, (T)1.0, W, ldw, (T)0.0, R, ldr, q);   // :310
            if (q.world() > 1) {
                T* G = ws.alloc<T>(k * k);
                lapack::laset(MatrixType::General, k, k, (T)0, (T)0, G, k, q);
                lapack::lacpy(MatrixType::Upper, k, k, R, ldr, G, k, q);
                q.allreduce_sum(G, k * k);
                lapack::lacpy(MatrixType::Upper, k, k, G, k, R, ldr, q);
            }
            if (lapack::potrf(Uplo::Upper, k, R, ldr, q)) {                                                 // :311, :319-331
                std::vector<T> rd(k);
                lapack::get_diag(k, R, ldr, rd.data(), q);
                T running_max = rd[0], running_min = rd[0];
                const T cond_threshold = std::sqrt(eps / std::numeric_limits<T>::epsilon());
                for (int64_t i = 0; i < k; ++i) {
                    T curr = std::abs(rd[i]);
                    running_max = std::max(running_max, curr);
                    running_min = std::min(running_min, curr);
                    if ((running_min * cond_threshold < running_max) && i > 1) { new_rank = i - 1; break; }
                }
            }
            rank = new_rank;                                                                                 // :335
            if (new_rank == n) {
                blas::trsm_gather(Diag::NonUnit, m, n, (T)1.0, R, ldr, W, ldw, (int64_t const*)nullptr, A, lda, q);   // :338, W -> A
            } else {       // the Cholesky factorization stopped early: A takes A_pre and the leading columns are solved in place, as in the reference
                lapack::lacpy(MatrixType::General, m, n, W, ldw, A, lda, q);
                blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, new_rank, (T)1.0, R, ldr, A, lda, q);
            }
            auto t6 = stamp();
            return finish(m, n, A, lda, A_hat, d, R, ldr, new_rank, state, t_total0, t0, t1, t2, t3, t4, t5, t6);
        }
        // :287-288 (the reference permutes with a scratch copy of J because lapmt uses it as workspace; the device
        // kernel leaves its index vector untouched, so J itself is passed)
        util::col_swap(m, n, k, A, lda, J, q);
        auto t4 = stamp();
        ph("a_mod_trsm");

        for (int64_t i = 0; i < k; ++i)                                                                     // :296-301 diag_is_nonzero
            if (diag[i] == (T)0) return 1;
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, k, (T)1.0, R, ldr, A, lda, q);   // :302
        auto t5 = stamp();
        ph("cholqr");

        blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, k, m, (T)1.0, A, lda, (T)0.0, R, ldr, q);     // :310
        if (q.world() > 1) {                    // Gram matrix of the sharded rows: sum the upper triangles over the ranks
            T* G = ws.alloc<T>(k * k);
            lapack::laset(MatrixType::General, k, k, (T)0, (T)0, G, k, q);
            lapack::lacpy(MatrixType::Upper, k, k, R, ldr, G, k, q);
            q.allreduce_sum(G, k * k);
            lapack::lacpy(MatrixType::Upper, k, k, G, k, R, ldr, q);
        }
        if (lapack::potrf(Uplo::Upper, k, R, ldr, q)) {                                                     // :311
            // a-posteriori rank estimate from the (partially factored) diagonal                               :319-331
            std::vector<T> rd(k);
            lapack::get_diag(k, R, ldr, rd.data(), q);
            T running_max = rd[0], running_min = rd[0];
            const T cond_threshold = std::sqrt(eps / std::numeric_limits<T>::epsilon());
            for (int64_t i = 0; i < k; ++i) {
                T curr = std::abs(rd[i]);
                running_max = std::max(running_max, curr);
                running_min = std::min(running_min, curr);
                if ((running_min * cond_threshold < running_max) && i > 1) { new_rank = i - 1; break; }
            }
        }
        rank = new_rank;                                                                                     // :335
        blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m, new_rank, (T)1.0, R, ldr, A, lda, q);   // :338
        auto t6 = stamp();
        return finish(m, n, A, lda, A_hat, d, R, ldr, new_rank, state, t_total0, t0, t1, t2, t3, t4, t5, t6);
    }

    using clk_tp = std::chrono::steady_clock::time_point;
    // everything after the second solve (rl_cqrrpt.hh:341-384): R = R_chol * R_sk or the orthogonal completion, timing vector
    int finish(int64_t m, int64_t n, T* A, int64_t lda, T* A_hat, int64_t d, T* R, int64_t ldr, int64_t new_rank, RandBLAS::RNGState<RNG>& state,
               clk_tp t_total0, clk_tp t0, clk_tp t1, clk_tp t2, clk_tp t3, clk_tp t4, clk_tp t5, clk_tp t6) {
        using clk = std::chrono::steady_clock;
        auto stamp = [&]() { if (timing) q.sync(); return clk::now(); };
        if (!orthogonalization) {
            // R <- R_chol * R_sk (rows 0..new_rank-1, all n columns)                                           :345
            blas::trmm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, new_rank, n, (T)1.0, A_hat, d, R, ldr, q);
        } else if (new_rank != n) {                                                                         // :347-367
            // complete the orthonormal set: Gaussian trailing columns, projected against Q, Householder-orthogonalised.
            // (the reference passes &A[new_rank*lda] to fill_dense as an m x cols buffer with ld m and DISCARDS the returned
            //  state, :351-352 -- both kept: the fill goes through a packed scratch and `state` is left alone)
            const int64_t cols_to_fill = n - new_rank;
            blas::Scratch w2(q);
            T* Gs = w2.alloc<T>(m * cols_to_fill);
            T* temp = w2.alloc<T>(new_rank * cols_to_fill);
            T* tau_orth = w2.alloc<T>(cols_to_fill);
            T* Gc = &A[new_rank * lda];
            if (q.world() > 1) {
                // row-sharded: every rank draws ITS rows of the one global Gaussian block, the projection coefficients Q^T G sum over the
                // ranks, and the Householder step is the one-exchange TSQR of rl_orth.hh
                int64_t m_glob = m, row0 = 0;
                q.shard_extent(m, m_glob, row0);
                RandBLAS::DenseDist Dg(m_glob, cols_to_fill);
                (void)RandBLAS::fill_dense_rows(Dg, row0, m, Gs, state, q);
                lapack::lacpy(MatrixType::General, m, cols_to_fill, Gs, m, Gc, lda, q);
                blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, new_rank, cols_to_fill, m, (T)1.0, A, lda, Gc, lda, (T)0.0, temp, new_rank, q);
                q.allreduce_sum(temp, new_rank * cols_to_fill);
                blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, cols_to_fill, new_rank, (T)-1.0, A, lda, temp, new_rank, (T)1.0, Gc, lda, q);
                detail::tsqr_q(q, m, cols_to_fill, Gc, lda);
            } else {
                RandBLAS::DenseDist Dg(m, cols_to_fill);
                (void)RandBLAS::fill_dense(Dg, Gs, state, q);
                lapack::lacpy(MatrixType::General, m, cols_to_fill, Gs, m, Gc, lda, q);
                blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, new_rank, cols_to_fill, m, (T)1.0, A, lda, Gc, lda, (T)0.0, temp, new_rank, q);
                blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, cols_to_fill, new_rank, (T)-1.0, A, lda, temp, new_rank, (T)1.0, Gc, lda, q);
                lapack::geqrf(m, cols_to_fill, Gc, lda, tau_orth, q);
                lapack::ungqr(m, cols_to_fill, cols_to_fill, Gc, lda, tau_orth, q);
            }
        }
        if (timing) {                                                                                        // :370-384
            auto t7 = stamp();
            auto us = [](clk::time_point a, clk::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
            long saso = us(t0, t1), qrcp_t = us(t1, t2), rr = us(t2, t3), piv = us(t3, t4), trsm_t = us(t4, t5), chol = us(t5, t6);
            long total = us(t_total0, t7);
            times = {saso, qrcp_t, rr, chol, piv, trsm_t, total - (saso + qrcp_t + rr + chol + piv + trsm_t), total};
        }
        return 0;
    }

    blas::Queue& q;
    bool timing;
    T eps;
    int64_t rank;
    std::vector<long> times;   // {saso, qrcp, rank_reveal, cholqr, a_mod_piv, a_mod_trsm, rest, total} in microseconds
    int64_t nnz;
    Subroutines::QRCP qrcp;
    double bqrrp_block_ratio;      // rl_cqrrpt.hh:133-137
    int64_t nb_alg;
    int64_t oversampling;
    int64_t panel_pivoting;
    int64_t use_cholqr;
    bool orthogonalization;
    // (not in the reference) the column pivoting is folded into the first preconditioning solve instead of a separate pass over A;
    // false restores the reference's statement order col_swap -> trsm -> syrk -> trsm, all in place
    bool fold_pivoting;
    // (not in the reference) geqp3 of the sketch in two halves with the left half of the first solve beside the second one; false keeps the
    // one-piece order.  split_cols: columns per workgroup of the second half (0 = auto, <= 16)
    bool split_qrcp;
    int64_t split_cols;
    // testing hooks (not in the reference): a d x n sketch to use instead of S*A / a buffer receiving the sketch
    const T* sketch_override = nullptr;
    T* sketch_export = nullptr;
};

}  // namespace RandLAPACK
