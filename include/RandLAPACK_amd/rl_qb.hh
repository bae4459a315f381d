// QBalg / QB (reference: RandLAPACK/comps/rl_qb.hh:18-268): blocked randomized QB, A ~= Q * B, adaptive stop.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include "rl_rf.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class QBalg {                                                     // rl_qb.hh:18-34
public:
    virtual ~QBalg() {}
    virtual int call(int64_t m, int64_t n, T* A, int64_t& k, int64_t b_sz, T tol, T*& Q, T*& BT,
                     RandBLAS::RNGState<RNG>& state) = 0;
};

template <typename T, typename RNG>
class QB : public QBalg<T, RNG> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    QB(RangeFinder<T, RNG>& rf_obj, Stabilization<T>& orth_obj, bool verb, bool orth) : QB(blas::default_queue(), rf_obj, orth_obj, verb, orth) {}   // rl_qb.hh:58-63
    QB(blas::Queue& queue, RangeFinder<T, RNG>& rf_obj, Stabilization<T>& orth_obj, bool verb, bool orth)
        : q(queue), rf(rf_obj), orth(orth_obj) {
        verbose = verb;
        orth_check = orth;
    }

    /// Q (m x k) and BT (n x k) are allocated HERE on the device (blas::device_malloc) and owned by the caller
    /// afterwards (blas::device_free), mirroring the reference's calloc/free ownership (rl_qb.hh:155-159); incoming
    /// non-null Q/BT are freed first.  k is in/out.  Return codes 0/2/3/4/5/6 as in the reference.
    ///
    /// Deviations that do not change results: Q/BT are sized for the requested k up front instead of growing by
    /// realloc (:180-182); when ONE block covers k (b_sz >= k) the m x n copy of A (:162,171) and the final,
    /// never-read rank-b update of that copy (:260) are skipped -- A itself is then read-only input.
    int call(int64_t m, int64_t n, T* A, int64_t& k, int64_t b_sz, T tol, T*& Q, T*& BT,
             RandBLAS::RNGState<RNG>& state) override {
        int64_t curr_sz = 0, next_sz = 0;
        tol = std::max(tol, 100 * std::numeric_limits<T>::epsilon());                                    // :149
        T norm_B = 0, prev_err = 0, approx_err = 0;
        if (Q) blas::device_free(Q, q);
        if (BT) blas::device_free(BT, q);
        const int64_t k_in = k;
        Q = blas::device_malloc<T>(m * k_in, q);
        BT = blas::device_malloc<T>(n * k_in, q);
        // (the reference callocs: here every column a block writes is written in full, and the columns NOT reached -- an early exit --
        //  are zeroed on the way out: `done` below; a 51 + 41 MB memset per call otherwise, 22 us of an 11 ms shard step)
        blas::Scratch ws(q);
        T* QtQi = ws.alloc<T>(k_in * std::max<int64_t>(1, std::min(b_sz, k_in)));
        const bool single_block = (b_sz >= k_in);
        T* A_cpy = A;
        T* A_own = nullptr;
        // ||A||_F (:168) is only consumed after the first block's B_i (:225); it is obtained from the first GEMM
        // that streams A (inside rf.call) instead of a separate 8*m*n-byte pass -- identical value, one pass less.
        T norm_A = 0;
        bool have_norm = false;
        if (!single_block) {
            A_own = blas::device_malloc<T>(m * n, q);
            lapack::lacpy(MatrixType::General, m, n, A, m, A_own, m, q);                                  // :171
            A_cpy = A_own;
        }
        if constexpr (sizeof(T) == 8) {
            q.norm_req = blas::Queue::NormRequest();
            q.norm_req.ptr = A_cpy; q.norm_req.rows = m; q.norm_req.cols = n; q.norm_req.ld = m;
            q.norm_req.defer = true;                               // read together with ||B_1||_F below: one host round trip for both
        }
        blas::RowsSharded sh(q, true);                             // Q_i, Q: rows sharded
        auto done = [&](int code) {
            q.norm_req = blas::Queue::NormRequest();
            if (A_own) blas::device_free(A_own, q);
            const int64_t kept = std::max<int64_t>(std::min(k, k_in), 0);      // columns [kept, k_in) hold nothing the caller may read: zero, as calloc would leave them
            if (kept < k_in) {
                blas::device_memset(Q + m * kept, 0, m * (k_in - kept), q);
                blas::device_memset(BT + n * kept, 0, n * (k_in - kept), q);
            }
            return code;
        };

        while (curr_sz < k) {
            b_sz = std::min(b_sz, k - curr_sz);                                                           // :175
            next_sz = curr_sz + b_sz;
            T* Q_i = Q + m * curr_sz;
            T* BT_i = BT + n * curr_sz;
            if (rf.call(m, n, A_cpy, b_sz, Q_i, state)) { k = curr_sz; q.norm_req = blas::Queue::NormRequest(); return done(6); }   // :190-196
            const bool norm_rides = !have_norm && q.norm_req.done;  // ||A||_F came out of rf.call's product; its value is collected below
            if (!have_norm && !norm_rides) norm_A = lapack::lange(Norm::Fro, m, n, A_cpy, m, q);   // A_cpy still equals A here
            if (orth_check && util::orthogonality_check(m, b_sz, Q_i, verbose, q)) { k = curr_sz; return done(4); }   // :198-206
            if (curr_sz != 0) {                                                                           // :209-215
                blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, curr_sz, b_sz, m, T(1), Q, m, Q_i, m, T(0), QtQi, next_sz, q);
                if (q.world() > 1) q.allreduce_sum(QtQi, next_sz * b_sz);
                blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, b_sz, curr_sz, T(-1), Q, m, QtQi, next_sz, T(1), Q_i, m, q);
                orth.call(m, b_sz, Q_i);
            }
            blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, n, b_sz, m, T(1), A_cpy, m, Q_i, m, T(0), BT_i, n, q);   // :218
            if (q.world() > 1) q.allreduce_sum(BT_i, n * b_sz);     // B_i^T = sum_g A_g^T Q_g : the n x b exchange
            T norm_B_i = lapack::lange(Norm::Fro, n, b_sz, BT_i, n, q);                                   // :221
            if (!have_norm) {
                if (norm_rides) norm_A = (T)q.collect_norm(true);   // (the stream has just been drained by lange: no second wait; row-sharded: the global norm)
                else if (q.world() > 1) {                              // ||A||_F^2 = sum over the row blocks
                    double ssq = (double)norm_A * (double)norm_A;
                    q.allreduce_sum_host(&ssq, 1);
                    norm_A = (T)std::sqrt(ssq);
                }
                q.norm_req = blas::Queue::NormRequest();
                have_norm = true;
            }
            norm_B = std::hypot(norm_B, norm_B_i);
            prev_err = approx_err;
            approx_err = std::sqrt(std::abs(norm_A - norm_B)) * (std::sqrt(norm_A + norm_B) / norm_A);    // :225
            if ((curr_sz > 0) && (approx_err > prev_err)) { k = curr_sz; return done(2); }                // :228-234
            if (orth_check && util::orthogonality_check(m, next_sz, Q, verbose, q)) { k = curr_sz; return done(5); }   // :236-244
            curr_sz += b_sz;                                                                              // :247
            if (approx_err < tol) { k = curr_sz; return done(0); }                                        // :250-256
            if (curr_sz < k)   // the reference also updates after the last block; that result is never read (:260)
                blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, n, b_sz, T(-1), Q_i, m, BT_i, n, T(1), A_cpy, m, q);
        }
        return done(3);                                                                                   // :267
    }

    blas::Queue& q;
    RangeFinder<T, RNG>& rf;
    Stabilization<T>& orth;
    bool verbose;
    bool orth_check;
};

}  // namespace RandLAPACK
