// RangeFinder / RF (reference: RandLAPACK/comps/rl_rf.hh:17-137): Q = orth(A * Omega).
#pragma once
#include <vector>
#include "rl_rs.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class RangeFinder {                                               // rl_rf.hh:17-29
public:
    virtual ~RangeFinder() {}
    virtual int call(int64_t m, int64_t n, const T* A, int64_t k, T* Q, RandBLAS::RNGState<RNG>& state) = 0;
};

template <typename T, typename RNG>
class RF : public RangeFinder<T, RNG> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    RF(RowSketcher<T, RNG>& rs_obj, Stabilization<T>& orth_obj, bool verb, bool cond) : RF(blas::default_queue(), rs_obj, orth_obj, verb, cond) {}   // rl_rf.hh:45-50
    RF(blas::Queue& queue, RowSketcher<T, RNG>& rs_obj, Stabilization<T>& orth_obj, bool verb, bool cond)
        : q(queue), rs(rs_obj), orth(orth_obj) {
        verbose = verb;
        cond_check = cond;
    }

    /// Q (m x k, device, caller allocated) <- orthonormal basis of range(A * Omega).
    /// returns 0; 1 if the sketcher failed; 2 if the orthogonalisation failed.          (rl_rf.hh:107-137)
    int call(int64_t m, int64_t n, const T* A, int64_t k, T* Q, RandBLAS::RNGState<RNG>& state) override {
        blas::Scratch ws(q);                                       // (the reference leaks Omega on the failure paths)
        T* Omega = ws.alloc<T>(n * k);
        if (rs.call(m, n, A, k, Omega, state)) return 1;                                                  // :118-120
        blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, k, n, T(1), A, m, Omega, n, T(0), Q, m, q);   // :123
        blas::RowsSharded sh(q, true);                             // Q's rows are the sharded dimension
        if (cond_check) cond_nums.push_back(util::cond_num_check(m, k, Q, verbose, q));                  // :125-127
        if (orth.call(m, k, Q)) return 2;                                                                 // :129-132
        return 0;
    }

    blas::Queue& q;
    RowSketcher<T, RNG>& rs;
    Stabilization<T>& orth;
    bool verbose;
    bool cond_check;
    std::vector<T> cond_nums;
};

}  // namespace RandLAPACK
