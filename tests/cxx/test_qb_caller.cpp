// Drop-in check of the C++ object layer: a CALLER written for the reference compiles and passes against RandLAPACK_amd.hh.
//
// The function `qb_low_exact_rank_checks` below is the body of the reference's own QB test
// (/root/reference/test/comps/test_qb.cc:126-176, `test_QB2_low_exact_rank`) -- the statements a user of RandLAPACK writes: build the
// algorithm objects without any queue argument, call QB, form A - QB, Q'Q - I and A_k - QB with blas:: / lapack:: / util:: free
// functions, compare the norms with eps^0.625.  What differs from the reference file, and nothing else:
//   * the include line (RandLAPACK_amd.hh instead of RandLAPACK.hh + gtest),
//   * where the arrays live: `buf<T>` hands out page-locked host memory the device can address (rlhip_malloc_host), so that the
//     caller's host-side statements (std::fill on the singular values, :154) keep working,
//   * the two arrays the callee allocates (Q, BT) are released with blas::device_free instead of free() -- they are device memory,
//   * one blas::default_queue().sync() before the caller's std::fill: the device calls are asynchronous, the reference's are not.
// GoogleTest is not in the image: ASSERT_NEAR is a three-line macro here.  Built by tests/cxx/Makefile with plain g++ against
// librlhip.so; run by tests/test_gpu_cxx.py (-m gpu).
#include "RandLAPACK_amd.hh"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <limits>
#include <vector>

using namespace blas;
using namespace lapack;

static int g_failures = 0;
#define ASSERT_NEAR(val, ref, tol)                                                                                   \
    do {                                                                                                             \
        if (!(std::abs((double)(val) - (double)(ref)) <= (double)(tol))) {                                           \
            std::printf("ASSERT_NEAR failed at line %d: %.3e vs %.3e (tol %.3e)\n", __LINE__, (double)(val), (double)(ref), (double)(tol)); \
            ++g_failures;                                                                                            \
            return;                                                                                                  \
        }                                                                                                            \
    } while (0)

// array with std::vector's .data() whose storage both the host statements and the device kernels can address
template <typename T>
struct buf {
    T* p = nullptr;
    explicit buf(int64_t n) {
        void* v = nullptr;
        blas::check(rlhip_malloc_host(blas::default_queue().ctx(), &v, (size_t)(n > 0 ? n : 1) * sizeof(T)), "malloc_host");
        p = (T*)v;
        std::fill(p, p + n, (T)0);
    }
    buf(buf const&) = delete;
    ~buf() { rlhip_free_host(blas::default_queue().ctx(), p); }
    T* data() { return p; }
};

template <typename T>
struct QBTestData {
    int64_t row, col, rank;
    buf<T> A, BT_cpy, A_hat, A_k, A_cpy, A_cpy_2, A_cpy_3, s, S, U, VT;
    QBTestData(int64_t m, int64_t n, int64_t k)
        : row(m), col(n), rank(k), A(m * n), BT_cpy(k * n), A_hat(m * n), A_k(m * n), A_cpy(m * n), A_cpy_2(m * n), A_cpy_3(m * n), s(n), S(n * n),
          U(m * n), VT(n * n) {}
};

// exactly the reference's aggregate (test_qb.cc:58-79): every constructor is called WITHOUT a queue
template <typename T, typename RNG>
struct algorithm_objects {
    RandLAPACK::PLUL<T> Stab;
    RandLAPACK::RS<T, RNG> RS;
    RandLAPACK::CholQRQ<T> Orth_RF;
    RandLAPACK::RF<T, RNG> RF;
    RandLAPACK::CholQRQ<T> Orth_QB;
    RandLAPACK::QB<T, RNG> QB;

    algorithm_objects(bool verbose, bool cond_check, bool orth_check, int64_t p, int64_t passes_per_iteration)
        : Stab(cond_check, verbose),
          RS(Stab, p, passes_per_iteration, verbose, cond_check),
          Orth_RF(cond_check, verbose),
          RF(RS, Orth_RF, verbose, cond_check),
          Orth_QB(cond_check, verbose),
          QB(RF, Orth_QB, verbose, orth_check) {}
};

template <typename T>
static void svd_and_copy_computational_helper(QBTestData<T>& all_data) {                 // test_qb.cc:81-98
    auto m = all_data.row;
    auto n = all_data.col;
    blas::copy(m * n, all_data.A.data(), 1, all_data.A_cpy.data(), 1);
    blas::copy(m * n, all_data.A.data(), 1, all_data.A_cpy_2.data(), 1);
    blas::copy(m * n, all_data.A.data(), 1, all_data.A_cpy_3.data(), 1);
    lapack::gesdd(Job::SomeVec, m, n, all_data.A_cpy.data(), m, all_data.s.data(), all_data.U.data(), m, all_data.VT.data(), n);
}

template <typename T, typename RNG, typename alg_type>
static void qb_low_exact_rank_checks(int64_t block_sz, T tol, QBTestData<T>& all_data, alg_type& all_algs, RandBLAS::RNGState<RNG>& state) {
    auto m = all_data.row;
    auto n = all_data.col;
    auto k = all_data.rank;

    T* A_dat = all_data.A.data();
    T* A_hat_dat = all_data.A_hat.data();
    T* A_k_dat = all_data.A_k.data();

    T* U_dat = all_data.U.data();
    T* s_dat = all_data.s.data();
    T* S_dat = all_data.S.data();
    T* VT_dat = all_data.VT.data();

    T* Q = nullptr;
    T* BT = nullptr;

    // Regular QB2 call
    all_algs.QB.call(m, n, all_data.A.data(), k, block_sz, tol, Q, BT, state);

    // Reassing pointers because Q, B have been resized
    T* Q_dat = Q;
    T* BT_dat = BT;
    T* BT_cpy_dat = all_data.BT_cpy.data();

    std::cout << "Inner dimension of QB: " << std::left << std::setw(25) << k << "\n";

    buf<T> Ident_buf(k * k);                          // (reference: std::vector<T> Ident(k * k, 0.0))
    T* Ident = Ident_buf.data();
    T* Ident_dat = Ident;
    // Generate a reference identity
    RandLAPACK::util::eye(k, k, Ident);
    // Buffer for testing B
    blas::copy(k * n, BT_dat, 1, BT_cpy_dat, 1);

    // A_hat = Q * B
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, n, k, 1.0, Q_dat, m, BT_dat, n, 0.0, A_hat_dat, m);
    // TEST 1: A = A - Q * B = 0
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, n, k, -1.0, Q_dat, m, BT_dat, n, 1.0, A_dat, m);
    // TEST 2: Q'Q = I
    blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, k, m, 1.0, Q_dat, m, -1.0, Ident_dat, k);

    // zero out the trailing singular values
    blas::default_queue().sync();                     // (host statement on memory the device has been writing: wait for the stream)
    std::fill(s_dat + k, s_dat + n, 0.0);
    RandLAPACK::util::diag(n, n, all_data.s.data(), n, all_data.S.data());

    // TEST 3: Below is A_k - A_hat = A_k - QB
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, 1.0, U_dat, m, S_dat, n, 1.0, A_k_dat, m);
    // A_k * VT -  A_hat == 0
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, 1.0, A_k_dat, m, VT_dat, n, -1.0, A_hat_dat, m);

    T test_tol = std::pow(std::numeric_limits<T>::epsilon(), 0.625);
    // Test 1 Output
    T norm_test_1 = lapack::lange(Norm::Fro, m, n, A_dat, m);
    std::cout << "FRO NORM OF A - QB:    " << std::scientific << norm_test_1 << "\n";
    ASSERT_NEAR(norm_test_1, 0, test_tol);
    // Test 2 Output
    T norm_test_3 = lapack::lansy(lapack::Norm::Fro, Uplo::Upper, k, Ident_dat, k);
    std::cout << "FRO NORM OF Q'Q - I:   " << std::scientific << norm_test_3 << "\n";
    ASSERT_NEAR(norm_test_3, 0, test_tol);
    // Test 3 Output
    T norm_test_4 = lapack::lange(Norm::Fro, m, n, A_hat_dat, m);
    std::cout << "FRO NORM OF A_k - QB:  " << std::scientific << norm_test_4 << "\n";
    ASSERT_NEAR(norm_test_4, 0, test_tol);
    blas::device_free(Q, blas::default_queue());      // (reference: free(Q); free(BT);)
    blas::device_free(BT, blas::default_queue());
}

// TEST_F(TestQB, Polynomial_Decay_general1) (test_qb.cc:236-262) and its block-size / power-iteration variants
static void run_case(int64_t m, int64_t n, int64_t k, int64_t p, int64_t passes_per_iteration, int64_t block_sz) {
    double tol = std::pow(std::numeric_limits<double>::epsilon(), 0.75);
    auto state = RandBLAS::RNGState<r123::Philox4x32>();

    bool verbose = false;
    bool cond_check = true;
    bool orth_check = true;

    QBTestData<double> all_data(m, n, k);
    algorithm_objects<double, r123::Philox4x32> all_algs(verbose, cond_check, orth_check, p, passes_per_iteration);

    RandLAPACK::gen::mat_gen_info<double> m_info(m, n, RandLAPACK::gen::polynomial);
    m_info.cond_num = 2025;
    m_info.rank = k;
    m_info.exponent = 2.0;
    RandLAPACK::gen::mat_gen(m_info, all_data.A.data(), state);

    svd_and_copy_computational_helper(all_data);
    qb_low_exact_rank_checks<double, r123::Philox4x32>(block_sz, tol, all_data, all_algs, state);
}

int main() {
    try {
        run_case(100, 100, 50, 2, 1, 2);               // Polynomial_Decay_general1
        run_case(100, 100, 50, 5, 2, 10);              // the same test at another block size / power scheme (test_qb.cc:264-290)
        run_case(500, 200, 100, 2, 1, 20);             // a tall input (a block size at which the reference's PLUL-stabilised power scheme keeps
                                                       // Q orthonormal: at 1000 x 400, b = 50 both the reference path and this one return code 4)
    } catch (std::exception const& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf(g_failures ? "FAILED (%d)\n" : "PASSED\n", g_failures);
    return g_failures ? 1 : 0;
}
