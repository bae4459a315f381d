// A caller of the two device classes of the reference's GPU path, written against their documented interface
// (RandLAPACK/drivers/rl_bqrrp_gpu.hh:27-149, rl_cqrrpt_gpu.hh:23-146) -- NOT lifted from the reference's tests:
//   1. BQRRP_GPU<T, RNG>: construct with (time_subroutines, block size), set the public `.qr_tall`, hand over device A and a device
//      sketch, read `.rank` and the 15-entry `.times`; the result is verified through the GEQP3 contract A[:, J] = Q R with Q rebuilt
//      from (V, tau) -- and through the BQRRP_GPU_alg base class pointer.
//   2. CQRRPT_GPU<T, RNG>: HOST matrices with lda > m and ldr > n, public members (nnz, no_hqrrp, ...), `.times` (8 entries); verified
//      with plain host loops (A[:, J] = Q R, Q^T Q = I, padding rows untouched).
// Built by tests/cxx/Makefile with the host compiler against librlhip.so, run by tests/test_gpu_cxx.py (-m gpu).
#include "RandLAPACK_amd.hh"

#include <cmath>
#include <cstdio>
#include <limits>
#include <memory>
#include <numeric>
#include <vector>

using RNG = r123::Philox4x32;
using GPUSubroutines = RandLAPACK::BQRRPGPUSubroutines;

static int g_fail = 0;
#define EXPECT(cond, ...)                                                       \
    do {                                                                        \
        if (!(cond)) { std::printf("FAILED line %d: ", __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++g_fail; } \
    } while (0)

template <typename T>
struct dev_array {
    T* p;
    explicit dev_array(int64_t n) : p(blas::device_malloc<T>(n)) {}
    dev_array(dev_array const&) = delete;
    ~dev_array() { blas::device_free(p); }
};

// ||A[:, J] - Q R||_F / ||A||_F and ||Q^T Q - I||_F on the device, from the GEQP3-format output
template <typename T>
static void geqp3_format_residuals(int64_t m, int64_t n, const T* A_orig, const T* A_out, const T* tau, const int64_t* J, T& rel_res, T& orth) {
    const int64_t k = std::min(m, n);
    dev_array<T> Q(m * k), R(k * n), AP(m * n), G(k * k);
    lapack::lacpy(lapack::MatrixType::General, m, k, A_out, m, Q.p, m);
    lapack::ungqr(m, k, k, Q.p, m, tau);
    lapack::laset(lapack::MatrixType::General, k, n, (T)0, (T)0, R.p, k);
    lapack::lacpy(lapack::MatrixType::Upper, k, n, A_out, m, R.p, k);
    lapack::lacpy(lapack::MatrixType::General, m, n, A_orig, m, AP.p, m);
    RandLAPACK::util::col_swap(m, n, n, AP.p, m, J);
    const T norm_A = lapack::lange(lapack::Norm::Fro, m, n, AP.p, m);
    blas::gemm(blas::Layout::ColMajor, blas::Op::NoTrans, blas::Op::NoTrans, m, n, k, (T)-1, Q.p, m, R.p, k, (T)1, AP.p, m);
    rel_res = lapack::lange(lapack::Norm::Fro, m, n, AP.p, m) / norm_A;
    blas::gemm(blas::Layout::ColMajor, blas::Op::Trans, blas::Op::NoTrans, k, k, m, (T)1, Q.p, m, Q.p, m, (T)0, G.p, k);
    lapack::add_diag(k, (T)-1, G.p, k);
    orth = lapack::lange(lapack::Norm::Fro, k, k, G.p, k);
}

template <typename T>
static void run_bqrrp_gpu(int64_t m, int64_t n, int64_t b_sz, int64_t d, GPUSubroutines::QRTall which, const char* label) {
    auto state = RandBLAS::RNGState<RNG>();
    dev_array<T> A(m * n), A_keep(m * n), A_sk(d * n), S(d * m), tau(n);
    dev_array<int64_t> J(n);
    // a Gaussian test matrix and its Gaussian sketch, both produced on the device
    state = RandBLAS::fill_dense(RandBLAS::DenseDist(m, n), A.p, state);
    state = RandBLAS::fill_dense(RandBLAS::DenseDist(d, m), S.p, state);
    blas::gemm(blas::Layout::ColMajor, blas::Op::NoTrans, blas::Op::NoTrans, d, n, m, (T)1, S.p, d, A.p, m, (T)0, A_sk.p, d);
    blas::device_copy_vector(m * n, A.p, A_keep.p);

    RandLAPACK::BQRRP_GPU<T, RNG> alg(true, b_sz);
    EXPECT(alg.qr_tall == GPUSubroutines::QRTall::geqrf, "default qr_tall must be geqrf");
    EXPECT(alg.tol == std::numeric_limits<T>::epsilon() && alg.block_size == b_sz && alg.timing, "constructor defaults");
    alg.qr_tall = which;
    RandLAPACK::BQRRP_GPU_alg<T, RNG>& base = alg;       // the abstract interface is what generic callers hold
    const int rc = base.call(m, n, A.p, m, A_sk.p, d, tau.p, J.p);
    EXPECT(rc == 0, "call returned %d", rc);
    EXPECT(alg.rank == std::min(m, n), "rank %lld on a full-rank Gaussian matrix", (long long)alg.rank);

    EXPECT(alg.times.size() == 15, "times has %zu entries", alg.times.size());
    if (alg.times.size() == 15) {
        const long sum = std::accumulate(alg.times.begin(), alg.times.begin() + 14, 0L);
        EXPECT(sum == alg.times[14], "the first 14 entries (%ld) must add up to the total (%ld)", sum, alg.times[14]);
        EXPECT(alg.times[9] > 0 && alg.times[11] > 0, "qr_tall / apply_transq were not timed");
        const bool chol = which == GPUSubroutines::QRTall::cholqr;
        EXPECT((alg.times[8] > 0) == chol && (alg.times[10] > 0) == chol, "preconditioning / reconstruction entries do not match qr_tall");
        std::printf("%-28s total %8ld us  qrcp_main %7ld  qr_tall %7ld  apply %7ld  sample_update %6ld\n", label, alg.times[14], alg.times[1],
                    alg.times[9], alg.times[11], alg.times[12]);
    }
    std::vector<int64_t> Jh((size_t)n);
    blas::copy_to_host(n, J.p, Jh.data());
    blas::default_queue().sync();
    std::vector<char> seen((size_t)n, 0);
    bool perm = true;
    for (int64_t j : Jh) { if (j < 1 || j > n || seen[(size_t)(j - 1)]) { perm = false; break; } seen[(size_t)(j - 1)] = 1; }
    EXPECT(perm, "J is not a permutation of 1..n");
    T res = 0, orth = 0;
    if (perm) geqp3_format_residuals(m, n, A_keep.p, A.p, tau.p, J.p, res, orth);
    const T atol = std::pow(std::numeric_limits<T>::epsilon(), std::is_same<T, double>::value ? (T)0.75 : (T)0.60);
    EXPECT(res <= atol, "||AP - QR|| / ||A|| = %.3e > %.3e", (double)res, (double)atol);
    EXPECT(orth / std::sqrt((T)n) <= atol, "||Q'Q - I|| / sqrt(n) = %.3e", (double)(orth / std::sqrt((T)n)));
}

static void run_cqrrpt_gpu() {
    const int64_t m = 900, n = 60, lda = m + 5, ldr = n + 3;
    std::vector<double> A((size_t)(lda * n), -1.0), A0, R((size_t)(ldr * n), 0.0);
    std::vector<int64_t> J((size_t)n, 0);
    // host input: a graded matrix, so that the pivoting has something to do
    uint64_t s = 12345;
    auto unif = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < m; ++i) A[(size_t)(i + j * lda)] = unif() * std::pow(0.9, (double)((j * 7) % n));
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = n; i < ldr; ++i) R[(size_t)(i + j * ldr)] = 42.0;         // padding below R: must survive the call
    A0 = A;
    auto state = RandBLAS::RNGState<RNG>();
    RandLAPACK::CQRRPT_GPU<double, RNG> alg(false, true, std::pow(std::numeric_limits<double>::epsilon(), 0.85));
    EXPECT(alg.no_hqrrp == 1 && alg.nb_alg == 64 && alg.oversampling == 10 && alg.use_cholqr == 0 && alg.panel_pivoting == 1, "constructor defaults");
    alg.nnz = 4;
    RandLAPACK::CQRRPT_GPU_alg<double, RNG>& base = alg;
    const int rc = base.call(m, n, A.data(), lda, R.data(), ldr, J.data(), 1.25, state);
    EXPECT(rc == 0, "CQRRPT_GPU returned %d", rc);
    EXPECT(alg.rank == n, "rank %lld", (long long)alg.rank);
    EXPECT(alg.times.size() == 8, "times has %zu entries", alg.times.size());
    if (alg.times.size() == 8) EXPECT(std::accumulate(alg.times.begin(), alg.times.begin() + 7, 0L) == alg.times[7], "times do not add up");
    EXPECT(state.counter[0] != 0, "the RNG state was not advanced");
    bool pad_ok = true;
    for (int64_t j = 0; j < n; ++j) {
        for (int64_t i = m; i < lda && j < n - 1; ++i) pad_ok = pad_ok && A[(size_t)(i + j * lda)] == -1.0;
        for (int64_t i = n; i < ldr && j < n - 1; ++i) pad_ok = pad_ok && R[(size_t)(i + j * ldr)] == 42.0;
    }
    EXPECT(pad_ok, "entries between the columns (rows m..lda of A, n..ldr of R) were modified");
    double res2 = 0, nrm2 = 0, orth2 = 0;
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < m; ++i) {
            double qr = 0;
            for (int64_t l = 0; l <= std::min<int64_t>(j, alg.rank - 1); ++l) qr += A[(size_t)(i + l * lda)] * R[(size_t)(l + j * ldr)];
            const double a = A0[(size_t)(i + (J[(size_t)j] - 1) * lda)];
            res2 += (a - qr) * (a - qr);
            nrm2 += a * a;
        }
    for (int64_t a = 0; a < n; ++a)
        for (int64_t b = 0; b < n; ++b) {
            double g = 0;
            for (int64_t i = 0; i < m; ++i) g += A[(size_t)(i + a * lda)] * A[(size_t)(i + b * lda)];
            g -= (a == b);
            orth2 += g * g;
        }
    const double atol = std::pow(std::numeric_limits<double>::epsilon(), 0.75);
    EXPECT(std::sqrt(res2 / nrm2) <= atol, "||AP - QR|| / ||A|| = %.3e", std::sqrt(res2 / nrm2));
    EXPECT(std::sqrt(orth2) <= atol, "||Q'Q - I|| = %.3e", std::sqrt(orth2));
    std::printf("CQRRPT_GPU %lld x %lld (lda %lld, ldr %lld): rank %lld, residual %.2e, orthogonality %.2e\n", (long long)m, (long long)n,
                (long long)lda, (long long)ldr, (long long)alg.rank, std::sqrt(res2 / nrm2), std::sqrt(orth2));
}

int main() {
    try {
        run_bqrrp_gpu<double>(1500, 700, 128, 128, GPUSubroutines::QRTall::cholqr, "BQRRP_GPU<double> cholqr");
        run_bqrrp_gpu<double>(1500, 700, 128, 160, GPUSubroutines::QRTall::geqrf, "BQRRP_GPU<double> geqrf");
        run_bqrrp_gpu<float>(1200, 500, 100, 100, GPUSubroutines::QRTall::cholqr, "BQRRP_GPU<float> cholqr");
        run_cqrrpt_gpu();
    } catch (std::exception const& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf(g_fail ? "FAILED (%d checks)\n" : "PASSED\n", g_fail);
    return g_fail ? 1 : 0;
}
