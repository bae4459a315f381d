// Test-matrix generators, device flavour (reference: RandLAPACK/testing/rl_gen.hh:22-790).  Same names, same mat_gen_info fields,
// same spectra; the matrix is GENERATED IN HBM (accuracy studies at scale without host generation or PCIe).
//
// Device design: singular vectors are explicit Householder Q factors of Gaussian blocks (device geqrf + ungqr), and
// A = (Q_U diag(s)) Q_V^T is one MFMA GEMM -- the same matrix the reference builds by applying the two implicit Q's with ormqr
// to a matrix holding diag(s) (rl_gen.hh:79-86), since only the leading k columns of either Q meet nonzeros.
// The random stream is this library's own ("parity unpinned" for the stream); the ORDER in which the state is
// consumed is the reference's (U before V, rows before V in the spiked matrix).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"
#include "rl_util.hh"

namespace RandLAPACK::gen {

enum mat_type { polynomial, exponential, gaussian, step, spiked, adverserial, bad_cholqr, kahan, custom_input };     // :22-31

template <typename T>
struct mat_gen_info {                                                                                                // :35-58
    int64_t rows;
    int64_t cols;
    int64_t rank;
    mat_type m_type;
    T cond_num = 1.0;
    T scaling = 1.0;
    T exponent = 1.0;
    bool diag = false;
    bool check_true_rank = false;
    T theta = 1.0;
    T perturb = 1.0;
    char* filename = nullptr;
    int workspace_query_mod = 0;
    T frac_spectrum_one = 0.1;
    mat_gen_info(int64_t& m, int64_t& n, mat_type t) : rows(m), cols(n), rank(n), m_type(t) {}
};

namespace detail {
inline void scal_cols(int64_t m, int64_t n, double* A, int64_t lda, const double* s, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_scal_cols_f64(q.ctx(), m, n, A, lda, s), "scal_cols"); }
inline void scal_cols(int64_t m, int64_t n, float* A, int64_t lda, const float* s, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_scal_cols_f32(q.ctx(), m, n, A, lda, s), "scal_cols"); }
inline void scal_rows_idx(int64_t cnt, const int64_t* idx, int64_t n, double* A, int64_t lda, double a, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_scal_rows_idx_f64(q.ctx(), cnt, idx, n, A, lda, a), "scal_rows_idx"); }
inline void scal_rows_idx(int64_t cnt, const int64_t* idx, int64_t n, float* A, int64_t lda, float a, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_scal_rows_idx_f32(q.ctx(), cnt, idx, n, A, lda, a), "scal_rows_idx"); }
inline void kahan(int64_t m, int64_t n, double* A, int64_t lda, double th, double p, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_gen_kahan_f64(q.ctx(), m, n, A, lda, th, p), "gen_kahan"); }
inline void kahan(int64_t m, int64_t n, float* A, int64_t lda, float th, float p, blas::Queue& q = blas::default_queue()) { blas::check(rlhip_gen_kahan_f32(q.ctx(), m, n, A, lda, th, p), "gen_kahan"); }

/// Q (rows x k, DEVICE, ld rows) <- orthonormal basis of a Gaussian block: fill_dense, geqrf, ungqr; threads the state
template <typename T, typename RNG>
void gaussian_orthonormal(int64_t rows, int64_t k, T* Q, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    RandBLAS::DenseDist D(rows, k);
    state = RandBLAS::fill_dense(D, Q, state, q);
    blas::Scratch ws(q);
    T* tau = ws.alloc<T>(k);
    lapack::geqrf(rows, k, Q, rows, tau, q);
    lapack::ungqr(rows, k, k, Q, rows, tau, q);
}
}  // namespace detail

/// A (m x n, ld m, DEVICE) = U diag(S) V^T with U (m x k), V (n x k) orthonormalised Gaussians; S: k HOST singular values.   (:62-101)
template <typename T, typename RNG>
void gen_singvec(int64_t m, int64_t n, T* A, int64_t k, const T* S, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    if (k > std::min(m, n)) throw std::runtime_error("gen_singvec: rank exceeds min(m, n)");
    if (m == 0 || n == 0) return;
    if (k == 0) { lapack::laset(MatrixType::General, m, n, (T)0, (T)0, A, m, q); return; }
    blas::Scratch ws(q);
    T* U = ws.alloc<T>(m * k);
    T* V = ws.alloc<T>(n * k);
    T* s_dev = ws.alloc<T>(k);
    detail::gaussian_orthonormal(m, k, U, state, q);
    detail::gaussian_orthonormal(n, k, V, state, q);
    blas::copy_to_device(k, S, s_dev, q);
    detail::scal_cols(m, k, U, m, s_dev, q);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m, n, k, (T)1, U, m, V, n, (T)0, A, m, q);
}

/// s_i = 1 / (a (i + b)^p) with the first frac_spectrum_one of the values equal to one and s_k = 1 / cond                      (:105-132)
template <typename T>
std::vector<T> gen_poly_singvals(int64_t k, T frac_spectrum_one, T cond, T p) {
    std::vector<T> s((size_t)k);
    const int offset = (int)std::floor(k * frac_spectrum_one);
    const T first_entry = 1.0, last_entry = first_entry / cond, neg_invp = -((T)1.0) / p;
    const T a = std::pow((std::pow(last_entry, neg_invp) - std::pow(first_entry, neg_invp)) / (k - offset), p);
    const T b = std::pow(a * first_entry, neg_invp) - offset;
    std::fill(s.begin(), s.begin() + offset, (T)1.0);
    for (int i = offset; i < k; ++i) s[(size_t)i] = 1 / (a * std::pow(i + b, p));
    return s;
}
template <typename T>
std::vector<T> gen_exp_singvals(int64_t k, T cond) {                                                                        // :168-180
    std::vector<T> s((size_t)k);
    const int offset = (int)std::floor(k * 0.1);
    const T t = -std::log(1 / cond) / (k - offset);
    T cnt = 0.0;
    std::fill(s.begin(), s.begin() + offset, (T)1.0);
    for (int i = offset; i < k; ++i) s[(size_t)i] = std::exp(++cnt * -t);
    return s;
}
template <typename T>
std::vector<T> gen_step_singvals(int64_t k, T cond) {                                                                       // :215-225
    std::vector<T> s((size_t)k);
    const int offset = (int)(k / 4);
    std::fill(s.begin(), s.begin() + offset, (T)1.0);
    std::fill(s.begin() + offset, s.begin() + 2 * offset, (T)8.0 / cond);
    std::fill(s.begin() + 2 * offset, s.begin() + 3 * offset, (T)4.0 / cond);
    std::fill(s.begin() + 3 * offset, s.end(), (T)1.0 / cond);
    return s;
}
template <typename T>
std::vector<T> gen_bad_cholqr_singvals(int64_t k, int64_t n, T cond) {                                                       // :369-379
    std::vector<T> s((size_t)k, (T)1.0);
    const int offset = (int)k;                   // as in the reference: the decaying tail starts at k, i.e. is empty
    const T t = std::log(std::pow((T)10, 8) / cond) / (1 - (n - offset));
    T cnt = 0.0;
    for (int i = offset; i < k; ++i) s[(size_t)i] = (std::exp(t) / std::pow((T)10, 8)) * (std::exp(++cnt * -t));
    return s;
}

namespace detail {
/// the shared tail of gen_{poly,exp,step,bad_cholqr}_mat: diagonal k x k matrix in A (ld k), or the full factored form
template <typename T, typename RNG>
void from_singvals(int64_t m, int64_t n, T* A, int64_t k, std::vector<T> const& s, bool diagon, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    if (diagon) {                                                                                                             // :155-156
        blas::Scratch ws(q);
        T* s_dev = ws.alloc<T>(k);
        blas::copy_to_device(k, s.data(), s_dev, q);
        lapack::laset(MatrixType::General, k, k, (T)0, (T)0, A, k, q);
        util::diag(k, k, s_dev, k, A, q);
    } else
        gen_singvec(m, n, A, k, s.data(), state, q);
}
}  // namespace detail

template <typename T, typename RNG>
void gen_poly_mat(int64_t& m, int64_t& n, T* A, int64_t k, T frac_spectrum_one, T cond, T p, bool diagon, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    detail::from_singvals(m, n, A, k, gen_poly_singvals(k, frac_spectrum_one, cond, p), diagon, state, q);                    // :134-160
}
template <typename T, typename RNG>
void gen_exp_mat(int64_t& m, int64_t& n, T* A, int64_t k, T cond, bool diagon, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    detail::from_singvals(m, n, A, k, gen_exp_singvals(k, cond), diagon, state, q);                                           // :184-207
}
template <typename T, typename RNG>
void gen_step_mat(int64_t& m, int64_t& n, T* A, int64_t k, T cond, bool diagon, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    detail::from_singvals(m, n, A, k, gen_step_singvals(k, cond), diagon, state, q);                                          // :229-252
}
template <typename T, typename RNG>
void gen_bad_cholqr_mat(int64_t& m, int64_t& n, T* A, int64_t k, T cond, bool diagon, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    detail::from_singvals(m, n, A, k, gen_bad_cholqr_singvals(k, n, cond), diagon, state, q);                                 // :383-405
}

/// Stacked copies of an n x n orthogonal V with floor(n/2) sampled rows (without replacement) scaled by spike_scale.         (:257-305)
template <typename T, typename RNG>
void gen_spiked_mat(int64_t& m, int64_t& n, T* A, T spike_scale, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    const int64_t num_rows_sampled = n / 2;
    std::vector<int64_t> rows((size_t)std::max<int64_t>(num_rows_sampled, 1));
    state = RandBLAS::repeated_fisher_yates(num_rows_sampled, m, 1, rows.data(), state, q);
    blas::Scratch ws(q);
    T* V = ws.alloc<T>(n * n);
    int64_t* rows_dev = ws.alloc<int64_t>(std::max<int64_t>(num_rows_sampled, 1));
    detail::gaussian_orthonormal(n, n, V, state, q);
    for (int64_t size = 0; size < m;) {
        const int64_t h = std::min(n, m - size);
        lapack::lacpy(MatrixType::General, h, n, V, n, A + size, m, q);
        size += h;
    }
    if (num_rows_sampled > 0) {
        blas::copy_to_device(num_rows_sampled, rows.data(), rows_dev, q);
        detail::scal_rows_idx(num_rows_sampled, rows_dev, n, A, m, spike_scale, q);
    }
}

/// A = U V: U = orth(Gaussian with its first 10 rows scaled by sigma), V = triu(orth(Gaussian)) with V_ii *= 10e-3 for i >= 11   (:310-365)
template <typename T, typename RNG>
void gen_oleg_adversarial_mat(int64_t& m, int64_t& n, T* A, T sigma, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    const T scaling_factor_V = (T)10e-3;
    blas::Scratch ws(q);
    T* U = ws.alloc<T>(m * n);
    T* V = ws.alloc<T>(n * n);
    T* tau = ws.alloc<T>(n);
    RandBLAS::DenseDist DU(m, n), DV(n, n);
    state = RandBLAS::fill_dense(DU, U, state, q);
    state = RandBLAS::fill_dense(DV, V, state, q);
    const int64_t head = std::min<int64_t>(10, m);
    std::vector<int64_t> idx((size_t)head);
    for (int64_t i = 0; i < head; ++i) idx[(size_t)i] = i;
    int64_t* idx_dev = ws.alloc<int64_t>(head);
    blas::copy_to_device(head, idx.data(), idx_dev, q);
    detail::scal_rows_idx(head, idx_dev, n, U, m, sigma, q);
    lapack::geqrf(m, n, U, m, tau, q);
    lapack::ungqr(m, n, n, U, m, tau, q);
    lapack::geqrf(n, n, V, n, tau, q);
    lapack::ungqr(n, n, n, V, n, tau, q);
    util::get_U(n, n, V, n, q);
    if (n > 11) {
        std::vector<T> dg((size_t)n);
        lapack::get_diag(n, V, n, dg.data(), q);
        for (int64_t i = 11; i < n; ++i) dg[(size_t)i] *= scaling_factor_V;
        T* dg_dev = ws.alloc<T>(n);
        blas::copy_to_device(n, dg.data(), dg_dev, q);
        util::diag(n, n, dg_dev, n, V, q);
    }
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m, n, n, (T)1, U, m, V, n, (T)0, A, m, q);
}

template <typename T>
void gen_kahan_mat(int64_t m, int64_t n, T* A, T theta, T perturb, blas::Queue& q = blas::default_queue()) { detail::kahan(m, n, A, m, theta, perturb, q); }   // :408-434

/// Numerical rank from the singular values (misc/rl_util.hh:426-448; the reference's "return i - 1" on the first small value is kept)
template <typename T>
int64_t rank_check(int64_t m, int64_t n, const T* A, blas::Queue& q = blas::default_queue()) {
    blas::Scratch ws(q);
    T* cpy = ws.alloc<T>(m * n);
    T* s = ws.alloc<T>(n);
    T* vt = ws.alloc<T>(n * n);
    lapack::lacpy(MatrixType::General, m, n, A, m, cpy, m, q);
    lapack::gesvdj(m, n, cpy, m, s, vt, n, q);
    std::vector<T> sh((size_t)n);
    blas::copy_to_host(n, s, sh.data(), q);
    for (int64_t i = 0; i < n; ++i)
        if (sh[(size_t)i] <= 5 * std::numeric_limits<T>::epsilon() * sh[0]) return i - 1;
    return n;
}

/// Dispatcher (:712-772).  A: DEVICE buffer of info.rows x info.cols (info.rank x info.rank when info.diag), ld = rows.
template <typename T, typename RNG>
void mat_gen(mat_gen_info<T>& info, T* A, RandBLAS::RNGState<RNG>& state, blas::Queue& q = blas::default_queue()) {
    switch (info.m_type) {
        case polynomial: gen_poly_mat(info.rows, info.cols, A, info.rank, info.frac_spectrum_one, info.cond_num, info.exponent, info.diag, state, q); break;
        case exponential: gen_exp_mat(info.rows, info.cols, A, info.rank, info.cond_num, info.diag, state, q); break;
        case gaussian: {
            RandBLAS::DenseDist D(info.rows, info.cols);
            state = RandBLAS::fill_dense(D, A, state, q);
        } break;
        case step: gen_step_mat(info.rows, info.cols, A, info.rank, info.cond_num, info.diag, state, q); break;
        case spiked:
            gen_spiked_mat(info.rows, info.cols, A, info.scaling, state, q);
            if (info.check_true_rank) info.rank = rank_check(info.rows, info.cols, A, q);
            break;
        case adverserial:
            gen_oleg_adversarial_mat(info.rows, info.cols, A, info.scaling, state, q);
            if (info.check_true_rank) info.rank = rank_check(info.rows, info.cols, A, q);
            break;
        case bad_cholqr: gen_bad_cholqr_mat(info.rows, info.cols, A, info.rank, info.cond_num, info.diag, state, q); break;
        case kahan: gen_kahan_mat(info.rows, info.cols, A, info.theta, info.perturb, q); break;
        case custom_input: throw std::runtime_error("mat_gen: custom_input reads a host text file; load it on the host and copy_to_device");
        default: throw std::runtime_error(std::string("Unrecognized case."));
    }
}

}  // namespace RandLAPACK::gen
