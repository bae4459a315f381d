// BQRRP_GPU_alg / BQRRPGPUSubroutines / BQRRP_GPU (reference: RandLAPACK/drivers/rl_bqrrp_gpu.hh:27-149, call :152-934): the
// device-resident blocked QR with randomized pivoting.  Interface as the reference's: ALL data lives on the device, the d x n
// sketch A_sk is an INPUT (the reference has no device sketching, rl_bqrrp_gpu.hh:56-58; here the caller may also produce it with
// RandBLAS::fill_dense + blas::gemm on the queue), `call` returns A in GEQP3 format, tau, 1-based J, and sets `rank`.
//
// Subroutines (rl_bqrrp_gpu.hh:62-70): qrcp_wide = LU-QR only, rank_est = naive, col_perm, qr_tall in {cholqr, geqrf} (default
// geqrf, :84), apply_trans_q = ormqr.  The loop is detail::bqrrp_factor (rl_bqrrp.hh), the same code BQRRP runs:
//   reference device step                                   here
//   transposition_gpu + lapack::getrf + LUQRCP_piv_process   rlhip_transpose + device getrf (LAPACK-identical pivots) + luqrcp_piv
//   col_swap_gpu through a second copy of A_sk / A / J       in-place cycle-following col_swap (no m x n copy; the three copy timers read 0)
//   all_of on the panel's first column                       any_abs_gt over ALL rows (the CPU class's test, rl_bqrrp.hh:373-379)
//   naive_rank_est (<= tol on the device)                    the CPU class's `<` on the diagonal (rl_bqrrp.hh:421-427)
//   cholqr: trsm, syrk, potrf, trsm, orhr_col_gpu, signs     same calls; the compact-WY T that orhr_col returns is kept and reused
//   cusolver ormqr                                           compact-WY apply on the MFMA GEMMs (from T for cholqr panels, from
//                                                            (V, tau) through larft for geqrf panels): one operator either way
#pragma once
#include <chrono>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>
#include <vector>
#include "rl_bqrrp.hh"

namespace RandLAPACK {

template <typename T, typename RNG>
class BQRRP_GPU_alg {
public:
    virtual ~BQRRP_GPU_alg() {}
    virtual int call(int64_t m, int64_t n, T* A, int64_t lda, T* A_sk, int64_t d, T* tau, int64_t* J) = 0;
};

// outside the class to keep symbols short, as in the reference (rl_bqrrp_gpu.hh:44-47)
struct BQRRPGPUSubroutines {
    enum QRTall { cholqr, geqrf };
};

template <typename T, typename RNG = RandBLAS::DefaultRNG>
class BQRRP_GPU : public BQRRP_GPU_alg<T, RNG> {
public:
    using GPUSubroutine = BQRRPGPUSubroutines;

    // the reference's signature (rl_bqrrp_gpu.hh:76-84); the queue-taking overload is for multi-stream callers
    BQRRP_GPU(bool time_subroutines, int64_t b_sz) : BQRRP_GPU(blas::default_queue(), time_subroutines, b_sz) {}
    BQRRP_GPU(blas::Queue& queue, bool time_subroutines, int64_t b_sz) : q(queue) {
        randlapack_require(b_sz > 0) << "BQRRP_GPU block size b_sz=" << b_sz << " must be > 0";
        timing = time_subroutines;
        tol = std::numeric_limits<T>::epsilon();
        block_size = b_sz;
        qr_tall = GPUSubroutine::QRTall::geqrf;
        rank = 0;
    }

    /// A (m x n, lda), A_sk (d x n, ld d; OVERWRITTEN), tau (n), J (n): device buffers.  d >= block_size.  Returns 0.
    int call(int64_t m, int64_t n, T* A, int64_t lda, T* A_sk, int64_t d, T* tau, int64_t* J) override {
        randlapack_require(m >= 0 && n >= 0 && lda >= m) << "BQRRP_GPU: bad dimensions m=" << m << " n=" << n << " lda=" << lda;
        randlapack_require(q.world() == 1) << "BQRRP_GPU is the single-device class; the row-sharded factorization is BQRRP::call on a sharded queue";
        const int64_t mn = std::min(m, n);
        if (mn == 0) { rank = 0; return 0; }
        randlapack_require(d >= std::min(block_size, mn)) << "BQRRP_GPU: sampling dimension d=" << d << " is below the block size " << block_size;
        using clk = std::chrono::steady_clock;
        auto us = [](clk::time_point a, clk::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        if (timing) q.sync();
        const auto t_begin = clk::now();
        detail::BqrrpOpts<T> P{block_size, /*internal_nb*/ block_size, tol, BQRRPSubroutines::QRCPWide::luqr,
                               qr_tall == GPUSubroutine::QRTall::cholqr ? BQRRPSubroutines::QRTall::cholqr : BQRRPSubroutines::QRTall::geqrf,
                               BQRRPSubroutines::ApplyTransQ::gemqrt, cholqr_fallback, cholqr_cond_limit_inv, timing,
                               lookahead, lookahead_min_elems, lookahead_min_block};
        detail::BqrrpLaps L;
        // "Preallocation" (rl_bqrrp_gpu.hh:213-334: eleven cudaMallocAsync + workspace queries): here one reservation in the
        // queue's stream-ordered arena at the top of the loop function -- nothing to time separately, the entry reads ~0.
        const long prealloc = 0;
        detail::bqrrp_factor(q, P, m, n, A, lda, A_sk, d, tau, J, rank, cholqr_fallbacks, L);
        lookaheads = L.lookaheads;
        if (timing) {
            q.sync();
            const long total = us(t_begin, clk::now());
            const long copy_A_sk = 0, copy_A = 0, copy_J = 0;            // pointer swaps in the reference, nothing here (in-place permutations)
            const long rest = total - (prealloc + L.qrcp_main + L.qrcp_piv + L.piv_A + L.upd_J + L.precond + L.qr_tall + L.recon + L.apply + L.upd_sk);
            // the reference's 15 entries in its order (rl_bqrrp_gpu.hh:829-834)
            times = {prealloc, L.qrcp_main, copy_A_sk, L.qrcp_piv, copy_A, L.piv_A, copy_J, L.upd_J, L.precond, L.qr_tall, L.recon, L.apply,
                     L.upd_sk, rest, total};
            if (print_timing) print_times();
        }
        return 0;
    }

    /// the reference prints this block from inside call() whenever timing is on (rl_bqrrp_gpu.hh:836-868); here on request
    void print_times(std::ostream& os = std::cout) const {
        if (times.size() != 15) return;
        static const char* names[15] = {"Preallocation", "QRCP_wide main", "Copy(A_sk)", "QRCP_wide piv", "Copy(A)", "Piv(A)", "Copy(J)", "J updating",
                                        "Preconditioning", "QR_tall", "Householder reconstruction", "Apply QT", "Sample updating", "Other routines", "Total"};
        os << "\n/------------BQRRP TIMING RESULTS BEGIN------------/\n";
        for (int i = 0; i < 15; ++i) os << std::left << std::setw(38) << (std::string(names[i]) + " time:") << times[(size_t)i] << " us\n";
        const double tot = (double)std::max<long>(times[14], 1);
        for (int i = 0; i < 14; ++i)
            os << std::left << std::setw(38) << (std::string(names[i]) + " takes") << std::fixed << std::setprecision(2) << 100.0 * (double)times[(size_t)i] / tot << "% of runtime\n";
        os << "/-------------BQRRP TIMING RESULTS END-------------/\n\n";
    }

public:
    bool timing;
    RandBLAS::RNGState<RNG> state;
    int64_t rank;
    int64_t block_size;
    // 15 entries - time in microseconds of the portions of the algorithm (layout above)
    std::vector<long> times;
    // naive rank estimation parameter
    T tol;
    // core subroutine option, controlled by the user
    GPUSubroutine::QRTall qr_tall;

    // ---- not in the reference
    blas::Queue& q;
    bool print_timing = false;        // write the timing block to stdout at the end of a timed call, as the reference always does
    bool cholqr_fallback = true;      // qr_tall = cholqr: Householder refactorization of a panel whose Cholesky QR is unsafe (see BQRRP)
    int64_t cholqr_fallbacks = 0;
    bool lookahead = true;            // (not in the reference) see BQRRP::lookahead
    double lookahead_min_elems = 2.5e8;
    int64_t lookahead_min_block = 256;
    int64_t lookaheads = 0;
    T cholqr_cond_limit_inv = std::pow(std::numeric_limits<T>::epsilon(), (T)0.25);
};

}  // namespace RandLAPACK
