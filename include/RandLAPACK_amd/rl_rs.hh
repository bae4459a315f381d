// RowSketcher / RS (reference: RandLAPACK/comps/rl_rs.hh:16-178): Gaussian sketching operator refined by p
// alternating passes with A and A^T, stabilised every q passes.
#pragma once
#include <vector>
#include "rl_orth.hh"
#include "rl_randblas.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class RowSketcher {                                               // rl_rs.hh:16-29
public:
    virtual ~RowSketcher() {}
    virtual int call(int64_t m, int64_t n, const T*& A, int64_t k, T*& Omega, RandBLAS::RNGState<RNG>& state) = 0;
};

template <typename T, typename RNG>
class RS : public RowSketcher<T, RNG> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    RS(Stabilization<T>& stab_obj, int64_t p, int64_t q_, bool verb, bool cond) : RS(blas::default_queue(), stab_obj, p, q_, verb, cond) {}           // rl_rs.hh:55-62
    RS(blas::Queue& queue, Stabilization<T>& stab_obj, int64_t p, int64_t q_, bool verb, bool cond)
        : q(queue), Stab_Obj(stab_obj) {
        verbose = verb;
        cond_check = cond;
        passes_over_data = p;
        passes_per_stab = q_;
    }

    /// Omega (n x k, device, caller allocated) <- sketching operator for Y = A * Omega.
    /// returns 0, or 1 if a stabilisation step failed.                      (rl_rs.hh:117-178)
    int call(int64_t m, int64_t n, const T*& A, int64_t k, T*& Omega, RandBLAS::RNGState<RNG>& state) override {
        const int64_t p = passes_over_data, qq = passes_per_stab;
        int64_t p_done = 0;
        blas::Scratch ws(q);
        T* Omega_1 = (p > 0) ? ws.alloc<T>(m * k) : nullptr;      // only touched when passes are requested
        if (p % 2 == 0) {
            RandBLAS::DenseDist D(n, k);
            state = RandBLAS::fill_dense(D, Omega, state, q);                                            // :132-135
        } else {
            // row-sharded: every rank regenerates ITS rows of the one global m x k operator, so the result does not
            // depend on how many ranks hold the matrix
            int64_t m_glob = m, row0 = 0;
            q.shard_extent(m, m_glob, row0);
            RandBLAS::DenseDist D(m_glob, k);
            if (m_glob == m) state = RandBLAS::fill_dense(D, Omega_1, state, q);                         // :137-139
            else state = RandBLAS::fill_dense_rows(D, row0, m, Omega_1, state, q);
            blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, n, k, m, T(1), A, m, Omega_1, m, T(0), Omega, n, q);  // :142
            if (q.world() > 1) q.allreduce_sum(Omega, n * k);      // A^T Omega_1 sums over the sharded rows
            ++p_done;
            { blas::RowsSharded rep(q, false);
              if ((p_done % qq == 0) && Stab_Obj.call(n, k, Omega)) return 1; }                           // :145-148
        }
        while (p - p_done > 0) {
            blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, k, n, T(1), A, m, Omega, n, T(0), Omega_1, m, q);  // :153
            ++p_done;
            { blas::RowsSharded sh(q, true);
              if (cond_check) cond_nums.push_back(util::cond_num_check(m, k, Omega_1, verbose, q));
              if ((p_done % qq == 0) && Stab_Obj.call(m, k, Omega_1)) return 1; }                         // :159-162
            blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, n, k, m, T(1), A, m, Omega_1, m, T(0), Omega, n, q);   // :165
            if (q.world() > 1) q.allreduce_sum(Omega, n * k);
            ++p_done;
            { blas::RowsSharded rep(q, false);
              if (cond_check) cond_nums.push_back(util::cond_num_check(n, k, Omega, verbose, q));
              if ((p_done % qq == 0) && Stab_Obj.call(n, k, Omega)) return 1; }                           // :171-172
        }
        return 0;
    }

    blas::Queue& q;
    Stabilization<T>& Stab_Obj;
    int64_t passes_over_data;
    int64_t passes_per_stab;
    bool verbose;
    bool cond_check;
    std::vector<T> cond_nums;
};

}  // namespace RandLAPACK
