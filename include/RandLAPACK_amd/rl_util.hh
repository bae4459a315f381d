// RandLAPACK::util helpers that steer control flow on the path (reference: RandLAPACK/misc/rl_util.hh),
// operating on device buffers.
#pragma once
#include <stdexcept>
#include <cmath>
#include <limits>
#include <vector>
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"

namespace RandLAPACK::util {

/// s_max / s_min of an m x n (m >= n) device matrix, inf if s_min == 0.      (rl_util.hh:403-424)
/// The reference takes an SVD of a copy with gesdd(NoVec); here the copy goes through the device Jacobi SVD.
template <typename T>
T cond_num_check(int64_t m, int64_t n, T const* A, bool verbose, blas::Queue& q = blas::default_queue()) {
    (void)verbose;
    if (q.reduce_over_rows())
        throw blas::Error("cond_num_check of a row-sharded matrix is not available (needs a distributed SVD)");
    blas::Scratch ws(q);
    T* cpy = ws.alloc<T>(m * n);
    T* s = ws.alloc<T>(n);
    T* vt = ws.alloc<T>(n * n);
    lapack::lacpy(MatrixType::General, m, n, A, m, cpy, m, q);
    lapack::gesvdj(m, n, cpy, m, s, vt, n, q);
    std::vector<T> sh(n);
    blas::copy_to_host(n, s, sh.data(), q);
    return (sh[n - 1] == 0) ? std::numeric_limits<T>::infinity() : sh[0] / sh[n - 1];
}

/// true when the columns of A have LOST orthonormality: ||A^T A - I||_F / sqrt(k) > 1e-10 (double) or 1e-2
/// (float).                                                                    (rl_util.hh:468-496)
/// As in the reference the Gram buffer's strictly lower triangle is left at zero.
template <typename T>
bool orthogonality_check(int64_t m, int64_t k, T const* A, bool verbose, blas::Queue& q = blas::default_queue()) {
    (void)verbose;
    blas::Scratch ws(q);
    T* G = ws.alloc<T>(k * k);
    lapack::laset(MatrixType::General, k, k, T(0), T(0), G, k, q);
    blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, k, m, T(1), A, m, T(0), G, k, q);
    if (q.reduce_over_rows()) q.allreduce_sum(G, k * k);
    if (k > 1) lapack::laset(MatrixType::Lower, k - 1, k, T(0), T(0), G + 1, k, q);   // keep the upper triangle only
    lapack::add_diag(k, T(-1), G, k, q);                                              // G - I
    T orth_err = lapack::lange(Norm::Fro, k, k, G, k, q);
    constexpr T tol = (sizeof(T) == sizeof(double)) ? (T)1.0e-10 : (T)1.0e-2;
    return orth_err / std::sqrt((T)k) > tol;
}


/// Forward column permutation (== lapack::lapmt(true, ...)): on exit column i of A holds former column idx[i]-1.
/// idx is a DEVICE vector of n 1-based indices and is left untouched (the reference's lapmt restores it).
/// k > n throws std::runtime_error like the reference.                               (rl_util.hh:151-164)
/// util::eye (misc/rl_util.hh:60-70): A (m x n, ld m, DEVICE) <- identity pattern
template <typename T>
void eye(int64_t m, int64_t n, T* A, blas::Queue& q = blas::default_queue()) { lapack::laset(MatrixType::General, m, n, (T)0, (T)1, A, m, q); }

/// util::diag (misc/rl_util.hh:84-96): overwrite the first k diagonal entries of S (m x n, ld m, DEVICE) with s (k, DEVICE)
template <typename T>
void diag(int64_t m, int64_t n, const T* s_vec, int64_t k, T* S, blas::Queue& q = blas::default_queue()) {
    if (k > std::min(m, n)) throw std::runtime_error("Invalid rank parameter.");
    if (k > 0) lapack::lacpy(MatrixType::General, 1, k, s_vec, 1, S, m + 1, q);     // a 1 x k matrix with ld 1 -> stride m + 1
}

/// util::get_L (misc/rl_util.hh:102-115): zero the strictly upper triangle of A (m x n, ld m), optionally set the diagonal to one
template <typename T>
void get_L(int64_t m, int64_t n, T* A, int overwrite_diagonal, blas::Queue& q = blas::default_queue()) {
    if (overwrite_diagonal) lapack::laset(MatrixType::Upper, m, n, (T)0, (T)1, A, m, q);
    else if (n > 1) lapack::laset(MatrixType::Upper, m, n - 1, (T)0, (T)0, A + m, m, q);
}

/// util::get_U (misc/rl_util.hh:119-131): zero the strictly lower triangle of A (m x n, lda)
template <typename T>
void get_U(int64_t m, int64_t n, T* A, int64_t lda, blas::Queue& q = blas::default_queue()) {
    if (m > 1) lapack::laset(MatrixType::Lower, m - 1, n, (T)0, (T)0, A + 1, lda, q);
}

/// util::diag_is_nonzero (misc/rl_util.hh:138-142) on a DEVICE matrix: exact comparison with zero, as the reference
template <typename T>
bool diag_is_nonzero(int64_t n, const T* R, int64_t ldr, blas::Queue& q = blas::default_queue()) {
    std::vector<T> d((size_t)std::max<int64_t>(n, 1));
    lapack::get_diag(n, R, ldr, d.data(), q);
    for (int64_t i = 0; i < n; ++i)
        if (d[(size_t)i] == (T)0) return false;
    return true;
}

/// util::transposition (misc/rl_util.hh:315-334): AT (ld ldat) = A^T; copy_upper_triangle != 0 moves only the upper triangle of the
/// leading n x n block (the reference ignores m in that mode).
template <typename T>
void transposition(int64_t m, int64_t n, const T* A, int64_t lda, T* AT, int64_t ldat, int copy_upper_triangle, blas::Queue& q = blas::default_queue()) {
    const int64_t mm = copy_upper_triangle ? n : m;
    if constexpr (sizeof(T) == 8) blas::check(rlhip_transpose_f64(q.ctx(), mm, n, (const double*)A, lda, (double*)AT, ldat, copy_upper_triangle), "transposition");
    else blas::check(rlhip_transpose_f32(q.ctx(), mm, n, (const float*)A, lda, (float*)AT, ldat, copy_upper_triangle), "transposition");
}

template <typename T>
void col_swap(int64_t m, int64_t n, int64_t k, T* A, int64_t lda, int64_t const* idx, blas::Queue& q = blas::default_queue()) {
    if (k > n) throw std::runtime_error("Invalid rank parameter.");
    if constexpr (sizeof(T) == 8) blas::check(rlhip_col_swap_f64(q.ctx(), m, n, k, (double*)A, lda, idx), "col_swap");
    else blas::check(rlhip_col_swap_f32(q.ctx(), m, n, k, (float*)A, lda, idx), "col_swap");
}
/// Integer-vector overload: the first k entries of A are permuted by the permutation idx of 1..k (rl_util.hh:174-198)
inline void col_swap(int64_t n, int64_t k, int64_t* A, int64_t const* idx, blas::Queue& q = blas::default_queue()) {
    if (k > n) throw std::runtime_error("Incorrect rank parameter.");
    blas::check(rlhip_col_swap_i64(q.ctx(), n, k, A, idx), "col_swap_i64");
}

}  // namespace RandLAPACK::util
