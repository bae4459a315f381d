// RSVDalg / RSVD (reference: RandLAPACK/drivers/rl_rsvd.hh:15-154): A ~= U diag(S) V^T via QB + SVD of B^T.
#pragma once
#include "rl_exceptions.hh"
#include "rl_qb.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class RSVDalg {                                                   // rl_rsvd.hh:15-32
public:
    virtual ~RSVDalg() {}
    virtual int call(int64_t m, int64_t n, T* A, int64_t& k, T tol, T*& U, T*& S, T*& V,
                     RandBLAS::RNGState<RNG>& state) = 0;
};

template <typename T, typename RNG>
class RSVD : public RSVDalg<T, RNG> {
public:
    // the reference's signature (no queue): the process-wide default queue, as the reference's device drivers use Queue(0)
    RSVD(QBalg<T, RNG>& qb_obj, int64_t b_sz) : RSVD(blas::default_queue(), qb_obj, b_sz) {}                                            // rl_rsvd.hh:49-52
    RSVD(blas::Queue& queue, QBalg<T, RNG>& qb_obj, int64_t b_sz) : q(queue), QB_Obj(qb_obj) { block_sz = b_sz; }

    /// A (m x n, device, not modified) ~= U diag(S) V^T with U m x k, S k, V n x k (V is NOT transposed).
    /// U, S, V are allocated here on the device and owned by the caller (blas::device_free), like the
    /// reference's calloc'd outputs (rl_rsvd.hh:139-143).  Always returns 0; QB's return code is kept in
    /// `qb_return` (the reference discards it, :137).
    int call(int64_t m, int64_t n, T* A, int64_t& k, T tol, T*& U, T*& S, T*& V,
             RandBLAS::RNGState<RNG>& state) override {
        randlapack_require(m >= 0) << "m=" << m << " must be >= 0";                                    // :128-132
        randlapack_require(n >= 0) << "n=" << n << " must be >= 0";
        randlapack_require(k > 0) << "target rank k=" << k << " must be > 0";
        randlapack_require(tol >= (T)0) << "tol=" << tol << " must be >= 0";
        randlapack_require(!(A == nullptr && m > 0 && n > 0))
            << "A buffer is null but m=" << m << " and n=" << n << " imply a nonempty matrix";

        T* Q = nullptr;
        T* BT = nullptr;
        qb_return = QB_Obj.call(m, n, A, k, block_sz, tol, Q, BT, state);                                // :137

        const int64_t kk = std::max<int64_t>(k, 1);
        U = blas::device_malloc<T>(m * kk, q);                                                           // :141-143
        S = blas::device_malloc<T>(kk, q);
        V = blas::device_malloc<T>(n * kk, q);
        if (k > 0) {
            blas::Scratch ws(q);
            T* UT_buf = ws.alloc<T>(k * k);
            svd_info = lapack::gesdd(Job::SomeVec, n, k, BT, n, S, V, n, UT_buf, k, q);                  // :146
            blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, k, k, T(1), Q, m, UT_buf, k, T(0), U, m, q);   // :148
        }
        blas::device_free(Q, q);                                                                          // :150-152
        blas::device_free(BT, q);
        return 0;
    }

    blas::Queue& q;
    QBalg<T, RNG>& QB_Obj;
    int64_t block_sz;
    int qb_return = 0;
    int64_t svd_info = 0;
};

}  // namespace RandLAPACK
