// CQRRPT_GPU_alg / CQRRPT_GPU (reference: RandLAPACK/drivers/rl_cqrrpt_gpu.hh:23-146, call :149-390).  The reference's class takes
// HOST matrices: it sketches, pivots and permutes on the CPU and moves A to the device only for trsm / syrk / potrf / trsm / trmm
// (:280-376).  Same interface here -- host A (m x n, lda), host R (n x n, ldr), host J, the same public members with the same
// defaults -- but every stage runs on the device (SASO sketch, QRCP of the sketch, column permutation, Cholesky QR): the class
// uploads A and R, runs CQRRPT on the queue and downloads Q, R and J.  Callers whose data already lives in HBM use CQRRPT
// (rl_cqrrpt.hh) directly and skip the two PCIe passes; a benchmark must not time this adaptor's transfers as factorization time.
//
// Reference quirks NOT reproduced: R's trailing block is addressed with n*k where ldr*k is meant
// (:257, wrong for ldr > n); `rank` is reset to the a-priori estimate k after the a-posteriori estimate has been computed (:355,
// against its own comment); nnz has no default (:72-80 leave it uninitialised -- here 2, the CPU class's default); a zero on the
// diagonal of R_sk is not guarded (the CPU class returns 1, rl_cqrrpt.hh:296-301, and so does this one).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#include "rl_cqrrpt.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class CQRRPT_GPU_alg {
public:
    virtual ~CQRRPT_GPU_alg() {}
    virtual int call(int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr, int64_t* J, T d_factor, RandBLAS::RNGState<RNG>& state) = 0;
};

template <typename T, typename RNG = RandBLAS::DefaultRNG>
class CQRRPT_GPU : public CQRRPT_GPU_alg<T, RNG> {
public:
    // the reference's signature (rl_cqrrpt_gpu.hh:62-76); the queue-taking overload is for multi-stream callers
    CQRRPT_GPU(bool verb, bool time_subroutines, T ep) : CQRRPT_GPU(blas::default_queue(), verb, time_subroutines, ep) {}
    CQRRPT_GPU(blas::Queue& queue, bool verb, bool time_subroutines, T ep) : q(queue) {
        verbosity = verb;
        timing = time_subroutines;
        eps = ep;
        no_hqrrp = 1;
        nb_alg = 64;
        oversampling = 10;
        use_cholqr = 0;
        panel_pivoting = 1;
        nnz = 2;
        num_threads = 1;
        rank = 0;
    }

    /// A (m x n, lda), R (n x n, ldr), J (n): HOST buffers, as in the reference.  On exit A holds Q (m x rank, explicit), R the
    /// rank x n upper-trapezoidal factor, J the 1-based pivots.  Returns 0, or 1 when the sketch's R factor is singular.
    int call(int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr, int64_t* J, T d_factor, RandBLAS::RNGState<RNG>& state) override {
        randlapack_require(m >= 0 && n >= 0) << "CQRRPT_GPU: m=" << m << ", n=" << n << " must be >= 0";
        randlapack_require(lda >= m) << "lda=" << lda << " < m=" << m;
        randlapack_require(ldr >= n) << "ldr=" << ldr << " < n=" << n;
        if (m == 0 || n == 0) { rank = 0; return 0; }
        using clk = std::chrono::steady_clock;
        const auto t_begin = clk::now();
        // the matrices keep their host leading dimensions on the device: one contiguous transfer each way, entries between
        // columns (rows m..lda of A, n..ldr of R) make the round trip untouched, as with the reference's cudaMemcpy of lda*n / ldr*n
        const int64_t lenA = lda * (n - 1) + m, lenR = ldr * (n - 1) + n;
        T* A_dev = blas::device_malloc<T>(lenA, q);
        blas::Scratch ws(q);
        T* R_dev = ws.alloc<T>(lenR);
        int64_t* J_dev = ws.alloc<int64_t>(n);
        blas::copy_to_device(lenA, A, A_dev, q);
        blas::copy_to_device(lenR, R, R_dev, q);

        CQRRPT<T, RNG> alg(q, timing, eps);
        alg.nnz = nnz;
        alg.qrcp = no_hqrrp ? CQRRPTSubroutines::QRCP::geqp3 : CQRRPTSubroutines::QRCP::hqrrp;        // :218-222
        alg.nb_alg = nb_alg;
        alg.oversampling = oversampling;
        alg.panel_pivoting = panel_pivoting;
        alg.use_cholqr = use_cholqr;
        const int64_t d = (int64_t)(d_factor * n);
        T* sk_dev = sketch_export_host ? ws.alloc<T>(d * n) : nullptr;
        alg.sketch_export = sk_dev;
        int rc = 0;
        try {
            rc = alg.call(m, n, A_dev, lda, R_dev, ldr, J_dev, d_factor, state);
        } catch (...) {
            blas::device_free(A_dev, q);
            throw;
        }
        rank = alg.rank;
        blas::copy_to_host(lenA, A_dev, A, q);
        blas::copy_to_host(lenR, R_dev, R, q);
        blas::copy_to_host(n, J_dev, J, q);
        if (sk_dev) blas::copy_to_host(d * n, sk_dev, sketch_export_host, q);
        q.sync();
        blas::device_free(A_dev, q);
        if (timing) {
            // {saso, qrcp, rank_reveal, cholqr, a_mod_piv, a_mod_trsm, rest, total} (:371); `total` and `rest` include the transfers
            times = alg.times;
            if (times.size() == 8) {
                const long total = (long)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t_begin).count();
                times[6] += total - times[7];
                times[7] = total;
            }
        }
        if (verbosity)
            std::printf("CQRRPT_GPU: %lld x %lld, d = %lld, nnz = %lld, rank %lld, rc %d\n", (long long)m, (long long)n,
                        (long long)(d_factor * n), (long long)nnz, (long long)rank, rc);
        return rc;
    }

public:
    bool verbosity;
    bool timing;
    T eps;
    int64_t rank;
    // 8 entries
    std::vector<long> times;
    // tuning SASOS
    int num_threads;     // kept for source compatibility: the device sketch has no thread count
    int64_t nnz;
    // HQRRP-related
    int no_hqrrp;
    int64_t nb_alg;
    int64_t oversampling;
    int64_t panel_pivoting;
    int64_t use_cholqr;

    // ---- not in the reference
    blas::Queue& q;
    T* sketch_export_host = nullptr;   // testing hook: HOST buffer (d x n) receiving the sketch that was factored
};

}  // namespace RandLAPACK
