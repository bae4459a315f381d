// A user program for the rangefinder / QB / RSVD object graph of RandLAPACK_amd.hh, written against the documented interface
// (RandLAPACK/comps/rl_{orth,rs,rf,qb}.hh, drivers/rl_rsvd.hh: constructor argument lists, call() signatures, return codes,
// callee-allocated outputs) -- its own scenarios and its own checks, not a transcription of any reference test:
//   1. planted rank: A = X * Y^T from two device-generated Gaussian factors; QB at rank r with several block sizes must reproduce A,
//      keep Q orthonormal, advance the RNG state and report 0 or 3 (tolerance met / requested rank exhausted);
//   2. tolerance-driven stop on a matrix with prescribed, geometrically decaying singular values: QB must stop early with code 0
//      and the true error ||A - Q B||_F / ||A||_F must respect the requested tolerance;
//   3. RSVD held through the abstract bases (RSVDalg <- QBalg <- RangeFinder <- RowSketcher / Stabilization): U, V orthonormal,
//      S positive and descending, S equal to the planted singular values, A = U S V^T;
//   4. the float instantiation of the same graph;
//   5. argument errors raise, k comes back unchanged.
// Every constructor is called WITHOUT a queue (the reference's signatures); matrices live in device memory, verification pulls
// k x k / n-sized pieces to the host.  Built by tests/cxx/Makefile with the host compiler against librlhip.so; run by
// tests/test_gpu_cxx.py (-m gpu).
#include "RandLAPACK_amd.hh"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <memory>
#include <string>
#include <vector>

using RNG = r123::Philox4x32;
using blas::Layout;
using blas::Op;

static int g_fail = 0;
#define EXPECT(cond, ...)                                                                                                  \
    do {                                                                                                                   \
        if (!(cond)) { std::printf("FAILED line %d: ", __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++g_fail; } \
    } while (0)

template <typename T>
struct dev_array {
    T* p;
    explicit dev_array(int64_t n) : p(blas::device_malloc<T>(n)) {}
    dev_array(dev_array const&) = delete;
    ~dev_array() { blas::device_free(p); }
    std::vector<T> host(int64_t n) const {
        std::vector<T> h((size_t)n);
        blas::copy_to_host(n, p, h.data());
        blas::default_queue().sync();
        return h;
    }
};

// ---- the object graphs a caller assembles -----------------------------------------------------------------------------------------
enum class StabKind { cholqr, householder, lu };

template <typename T>
static std::unique_ptr<RandLAPACK::Stabilization<T>> make_stab(StabKind kind) {
    const bool cond_check = false, verbose = false;
    switch (kind) {
        case StabKind::cholqr: return std::make_unique<RandLAPACK::CholQRQ<T>>(cond_check, verbose);
        case StabKind::householder: return std::make_unique<RandLAPACK::HQRQ<T>>(cond_check, verbose);
        default: return std::make_unique<RandLAPACK::PLUL<T>>(cond_check, verbose);
    }
}

template <typename T>
struct Pipeline {                                          // power-scheme stabiliser -> RS -> RF (own orth) -> QB (own re-orth)
    std::unique_ptr<RandLAPACK::Stabilization<T>> power_stab, rf_orth, qb_orth;
    RandLAPACK::RS<T, RNG> sketcher;
    RandLAPACK::RF<T, RNG> finder;
    RandLAPACK::QB<T, RNG> qb;
    Pipeline(StabKind power, StabKind orth, int64_t passes, int64_t passes_per_stab, bool orth_check)
        : power_stab(make_stab<T>(power)), rf_orth(make_stab<T>(orth)), qb_orth(make_stab<T>(orth)),
          sketcher(*power_stab, passes, passes_per_stab, /*verbose*/ false, /*cond_check*/ false),
          finder(sketcher, *rf_orth, /*verbose*/ false, /*cond_check*/ false),
          qb(finder, *qb_orth, /*verbose*/ false, orth_check) {}
};

// ---- measurements ------------------------------------------------------------------------------------------------------------------
template <typename T>
static T rel_residual_QB(int64_t m, int64_t n, int64_t k, const T* A, const T* Q, const T* BT) {          // ||A - Q BT^T||_F / ||A||_F
    dev_array<T> R(m * n);
    lapack::lacpy(lapack::MatrixType::General, m, n, A, m, R.p, m);
    const T nrm = lapack::lange(lapack::Norm::Fro, m, n, A, m);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, n, k, (T)-1, Q, m, BT, n, (T)1, R.p, m);
    return lapack::lange(lapack::Norm::Fro, m, n, R.p, m) / nrm;
}

template <typename T>
static T orth_defect(int64_t rows, int64_t k, const T* Q) {                                               // ||Q^T Q - I||_F
    dev_array<T> G(k * k);
    blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, k, k, rows, (T)1, Q, rows, Q, rows, (T)0, G.p, k);
    lapack::add_diag(k, (T)-1, G.p, k);
    return lapack::lange(lapack::Norm::Fro, k, k, G.p, k);
}

// A = X * Y^T with X (m x r), Y (n x r) Gaussian: exact rank r, generated entirely on the device
template <typename T>
static void planted_rank(int64_t m, int64_t n, int64_t r, T* A, RandBLAS::RNGState<RNG>& state) {
    dev_array<T> X(m * r), Y(n * r);
    state = RandBLAS::fill_dense(RandBLAS::DenseDist(m, r), X.p, state);
    state = RandBLAS::fill_dense(RandBLAS::DenseDist(n, r), Y.p, state);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, n, r, (T)1, X.p, m, Y.p, n, (T)0, A, m);
}

// A = Qx * diag(sigma) * Qy^T with Qx, Qy orthonormal (Householder Q factors of Gaussian blocks): singular values known exactly
template <typename T>
static void planted_spectrum(int64_t m, int64_t n, const std::vector<T>& sigma, T* A, RandBLAS::RNGState<RNG>& state) {
    const int64_t r = (int64_t)sigma.size();
    dev_array<T> X(m * r), Y(n * r), tau(r), D(r * r);
    state = RandBLAS::fill_dense(RandBLAS::DenseDist(m, r), X.p, state);
    state = RandBLAS::fill_dense(RandBLAS::DenseDist(n, r), Y.p, state);
    lapack::geqrf(m, r, X.p, m, tau.p);
    lapack::ungqr(m, r, r, X.p, m, tau.p);
    lapack::geqrf(n, r, Y.p, n, tau.p);
    lapack::ungqr(n, r, r, Y.p, n, tau.p);
    std::vector<T> Dh((size_t)(r * r), (T)0);
    for (int64_t i = 0; i < r; ++i) Dh[(size_t)(i + i * r)] = sigma[(size_t)i];
    blas::copy_to_device(r * r, Dh.data(), D.p);
    dev_array<T> XD(m * r);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, r, r, (T)1, X.p, m, D.p, r, (T)0, XD.p, m);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, n, r, (T)1, XD.p, m, Y.p, n, (T)0, A, m);
}

// ---- scenario 1 -------------------------------------------------------------------------------------------------------------------
template <typename T>
static void qb_planted_rank(int64_t m, int64_t n, int64_t r, int64_t block, StabKind power, StabKind orth, int64_t passes, const char* label) {
    auto state = RandBLAS::RNGState<RNG>(7);
    dev_array<T> A(m * n);
    planted_rank(m, n, r, A.p, state);
    const auto state_before = state;

    Pipeline<T> algs(power, orth, passes, /*passes_per_stab*/ 1, /*orth_check*/ true);
    T* Q = nullptr;
    T* BT = nullptr;
    int64_t k = r;
    const T tol = std::pow(std::numeric_limits<T>::epsilon(), (T)0.75);
    const int rc = algs.qb.call(m, n, A.p, k, block, tol, Q, BT, state);

    EXPECT(rc == 0 || rc == 3, "%s: QB returned %d", label, rc);
    EXPECT(k == r, "%s: k came back as %lld, planted rank %lld", label, (long long)k, (long long)r);
    EXPECT(Q != nullptr && BT != nullptr, "%s: outputs were not allocated", label);
    EXPECT(!(state.counter == state_before.counter), "%s: the RNG state did not advance", label);
    const T bound = std::pow(std::numeric_limits<T>::epsilon(), (T)0.625);
    const T res = rel_residual_QB(m, n, k, A.p, Q, BT), orth_err = orth_defect(m, k, Q);
    std::printf("QB %-34s rc %d  k %3lld  ||A-QB||/||A|| %.2e  ||Q'Q-I|| %.2e\n", label, rc, (long long)k, (double)res, (double)orth_err);
    EXPECT(res <= bound, "%s: residual %.3e above %.3e", label, (double)res, (double)bound);
    EXPECT(orth_err <= bound, "%s: Q is not orthonormal (%.3e)", label, (double)orth_err);

    // B^T = A^T Q (the definition, rl_qb.hh:218), recomputed from the outputs; for one block it must match what QB returned
    if (block >= r) {
        dev_array<T> BT2(n * k);
        blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, n, k, m, (T)1, A.p, m, Q, m, (T)0, BT2.p, n);
        const std::vector<T> ours = BT2.host(n * k);
        std::vector<T> theirs((size_t)(n * k));
        blas::copy_to_host(n * k, BT, theirs.data());
        blas::default_queue().sync();
        T dmax = 0, amax = 0;
        for (size_t i = 0; i < ours.size(); ++i) { dmax = std::max(dmax, std::abs(ours[i] - theirs[i])); amax = std::max(amax, std::abs(ours[i])); }
        EXPECT(dmax <= 64 * std::numeric_limits<T>::epsilon() * amax, "%s: BT is not A^T Q (%.2e of %.2e)", label, (double)dmax, (double)amax);
    }
    // a second call with the same objects and a FRESH state reproduces the factors bit for bit (deterministic kernels)
    T* Q2 = nullptr;
    T* BT3 = nullptr;
    int64_t k2 = r;
    auto state_again = state_before;
    const int rc2 = algs.qb.call(m, n, A.p, k2, block, tol, Q2, BT3, state_again);
    EXPECT(rc2 == rc && k2 == k, "%s: repeated call gave rc %d k %lld", label, rc2, (long long)k2);
    EXPECT(state_again.counter == state.counter, "%s: repeated call left a different RNG state", label);
    std::vector<T> h1((size_t)(n * k)), h2((size_t)(n * k));
    blas::copy_to_host(n * k, BT, h1.data());
    blas::copy_to_host(n * k, BT3, h2.data());
    blas::default_queue().sync();
    EXPECT(h1 == h2, "%s: two runs from the same state differ", label);
    blas::device_free(Q);
    blas::device_free(BT);
    blas::device_free(Q2);
    blas::device_free(BT3);
}

// ---- scenario 2 -------------------------------------------------------------------------------------------------------------------
static void qb_stops_at_tolerance() {
    using T = double;
    const int64_t m = 600, n = 300, r = 120, block = 8;
    std::vector<T> sigma((size_t)r);
    for (int64_t i = 0; i < r; ++i) sigma[(size_t)i] = std::pow(0.8, (T)i);       // sigma_i = 0.8^i
    auto state = RandBLAS::RNGState<RNG>(11);
    dev_array<T> A(m * n);
    planted_spectrum(m, n, sigma, A.p, state);

    Pipeline<T> algs(StabKind::householder, StabKind::cholqr, /*passes*/ 2, 1, /*orth_check*/ false);
    const T tol = 1e-3;
    T* Q = nullptr;
    T* BT = nullptr;
    int64_t k = r;
    const int rc = algs.qb.call(m, n, A.p, k, block, tol, Q, BT, state);
    const T res = rel_residual_QB(m, n, k, A.p, Q, BT);
    // smallest rank whose optimal truncation error is below tol: the randomized basis needs a few more columns, never fewer
    T total = 0;
    for (T s : sigma) total += s * s;
    int64_t k_opt = 0;
    for (T tail = total; k_opt < r && std::sqrt(tail / total) > tol; ++k_opt) tail -= sigma[(size_t)k_opt] * sigma[(size_t)k_opt];
    std::printf("QB tolerance stop: rc %d, k %lld (optimal %lld), true relative error %.2e for tol %.0e\n", rc, (long long)k, (long long)k_opt, res, tol);
    EXPECT(rc == 0, "expected the tolerance exit (0), got %d", rc);
    EXPECT(k < r && k % block == 0, "k = %lld is not an early, whole-block stop", (long long)k);
    EXPECT(k >= k_opt && k <= k_opt + 3 * block, "k = %lld is implausible next to the optimal rank %lld", (long long)k, (long long)k_opt);
    EXPECT(res <= 2 * tol, "true error %.3e does not respect the tolerance", res);
    EXPECT(orth_defect(m, k, Q) < 1e-10, "Q lost orthonormality over %lld blocks", (long long)(k / block));
    blas::device_free(Q);
    blas::device_free(BT);
}

// ---- scenario 3 / 4 ---------------------------------------------------------------------------------------------------------------
template <typename T>
static void rsvd_through_the_bases(int64_t m, int64_t n, int64_t r, int64_t block, const char* label) {
    std::vector<T> sigma((size_t)r);
    for (int64_t i = 0; i < r; ++i) sigma[(size_t)i] = (T)10 / (T)(1 + i);       // 10, 5, 3.33, ...: well separated
    auto state = RandBLAS::RNGState<RNG>(3);
    dev_array<T> A(m * n), A_keep(m * n);
    planted_spectrum(m, n, sigma, A.p, state);
    blas::device_copy_vector(m * n, A.p, A_keep.p);

    Pipeline<T> algs(StabKind::lu, StabKind::cholqr, /*passes*/ 2, 1, /*orth_check*/ false);
    RandLAPACK::RowSketcher<T, RNG>& as_sketcher = algs.sketcher;                  // the abstract interfaces are what generic code holds
    RandLAPACK::RangeFinder<T, RNG>& as_finder = algs.finder;
    RandLAPACK::QBalg<T, RNG>& as_qb = algs.qb;
    (void)as_sketcher;
    (void)as_finder;
    RandLAPACK::RSVD<T, RNG> driver(as_qb, block);
    RandLAPACK::RSVDalg<T, RNG>& rsvd = driver;

    T *U = nullptr, *S = nullptr, *V = nullptr;
    int64_t k = r;
    const T tol = std::pow(std::numeric_limits<T>::epsilon(), (T)0.75);
    const int rc = rsvd.call(m, n, A.p, k, tol, U, S, V, state);
    EXPECT(rc == 0 && k == r, "%s: RSVD rc %d, k %lld", label, rc, (long long)k);
    EXPECT(driver.qb_return == 0 || driver.qb_return == 3, "%s: QB inside RSVD returned %d", label, driver.qb_return);

    // the input is read-only for a single-block or multi-block call
    EXPECT(A.host(m * n) == A_keep.host(m * n), "%s: RSVD modified its input", label);

    std::vector<T> Sh((size_t)k);
    blas::copy_to_host(k, S, Sh.data());
    blas::default_queue().sync();
    const T eps = std::numeric_limits<T>::epsilon();
    T worst = 0;
    bool ordered = true;
    for (int64_t i = 0; i < k; ++i) {
        worst = std::max(worst, std::abs(Sh[(size_t)i] - sigma[(size_t)i]) / sigma[(size_t)i]);
        if (i && !(Sh[(size_t)i] <= Sh[(size_t)i - 1])) ordered = false;
        if (!(Sh[(size_t)i] > 0)) ordered = false;
    }
    // A ~= U diag(S) V^T
    dev_array<T> US(m * k), Sd(k * k);
    std::vector<T> Sdh((size_t)(k * k), (T)0);
    for (int64_t i = 0; i < k; ++i) Sdh[(size_t)(i + i * k)] = Sh[(size_t)i];
    blas::copy_to_device(k * k, Sdh.data(), Sd.p);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, k, k, (T)1, U, m, Sd.p, k, (T)0, US.p, m);
    const T res = rel_residual_QB(m, n, k, A.p, US.p, V);
    const T uo = orth_defect(m, k, U), vo = orth_defect(n, k, V);
    std::printf("RSVD %-32s k %3lld  max |dS|/S %.2e  ||A-USV'||/||A|| %.2e  ||U'U-I|| %.2e  ||V'V-I|| %.2e\n", label, (long long)k, (double)worst,
                (double)res, (double)uo, (double)vo);
    EXPECT(ordered, "%s: singular values are not positive and descending", label);
    EXPECT(worst <= 200 * eps * (T)r, "%s: singular values off by %.3e", label, (double)worst);
    const T bound = std::pow(eps, (T)0.625);
    const T orth_bound = std::sqrt(eps) * (T)(sizeof(T) == 4 ? 1 : 1e-3);   // single precision: Cholesky-QR's eps * cond^2 on a cond-16 spectrum
    EXPECT(res <= bound && uo <= std::max(bound, orth_bound) && vo <= std::max(bound, orth_bound), "%s: factor checks failed (%.2e %.2e %.2e)", label, (double)res, (double)uo, (double)vo);
    blas::device_free(U);
    blas::device_free(S);
    blas::device_free(V);
}

// ---- scenario 5 -------------------------------------------------------------------------------------------------------------------
static void argument_errors_raise() {
    Pipeline<double> algs(StabKind::cholqr, StabKind::cholqr, 0, 1, false);
    RandLAPACK::RSVD<double, RNG> rsvd(algs.qb, 4);
    auto state = RandBLAS::RNGState<RNG>();
    double *U = nullptr, *S = nullptr, *V = nullptr;
    int raised = 0;
    int64_t k = 0;                                                         // target rank must be positive
    try { rsvd.call(10, 10, nullptr, k, 0.0, U, S, V, state); } catch (std::exception const&) { ++raised; }
    k = 2;                                                                 // null A with a nonempty shape
    try { rsvd.call(10, 10, nullptr, k, 0.0, U, S, V, state); } catch (std::exception const&) { ++raised; }
    dev_array<double> A(100);
    try { rsvd.call(10, 10, A.p, k, -1.0, U, S, V, state); } catch (std::exception const&) { ++raised; }   // negative tolerance
    EXPECT(raised == 3, "%d of 3 invalid calls raised", raised);
    EXPECT(k == 2 && U == nullptr && S == nullptr && V == nullptr, "a rejected call touched its outputs");
    std::printf("argument checks: %d of 3 invalid calls raised\n", raised);
}

int main() {
    try {
        qb_planted_rank<double>(400, 250, 48, 48, StabKind::cholqr, StabKind::cholqr, 0, "one block, no power passes");
        qb_planted_rank<double>(400, 250, 48, 8, StabKind::householder, StabKind::cholqr, 2, "6 blocks, 2 passes, HQRQ stab");
        qb_planted_rank<double>(1500, 300, 60, 20, StabKind::lu, StabKind::householder, 3, "3 blocks, 3 passes, PLUL / HQRQ");
        qb_planted_rank<double>(300, 700, 40, 16, StabKind::cholqr, StabKind::cholqr, 1, "wide input, ragged last block");
        qb_stops_at_tolerance();
        rsvd_through_the_bases<double>(2000, 500, 32, 32, "double, one block");
        rsvd_through_the_bases<double>(900, 400, 30, 10, "double, three blocks");
        rsvd_through_the_bases<float>(1200, 300, 16, 16, "float, one block");
        argument_errors_raise();
    } catch (std::exception const& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf(g_fail ? "FAILED (%d)\n" : "PASSED\n", g_fail);
    return g_fail ? 1 : 0;
}
