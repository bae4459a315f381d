// Small kernels behind the device test-matrix generators (include/RandLAPACK_amd/rl_gen.hh; reference
// RandLAPACK/testing/rl_gen.hh): column / row scalings and the Kahan matrix.  HBM-bound, one pass each.
#include "rlhip_internal.h"
#include "../../include/rlhip.h"

namespace {

template <typename T>
__global__ void scal_cols_kernel(int64_t m, int64_t n, T* __restrict__ A, int64_t lda, const T* __restrict__ s) {
    const int64_t total = m * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % m, j = e / m;
        A[i + j * lda] *= s[j];
    }
}

// A[idx[r], :] *= alpha for r < cnt (rows listed once each)
template <typename T>
__global__ void scal_rows_idx_kernel(int64_t cnt, const int64_t* __restrict__ idx, int64_t n, T* __restrict__ A, int64_t lda, T alpha) {
    const int64_t total = cnt * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e % cnt, j = e / cnt;
        A[idx[r] + j * lda] *= alpha;
    }
}

// Kahan matrix (rl_gen.hh:408-434): A = diag(sin^i) * C + diag(perturb * eps * (m - i)), C unit upper triangular with -cos above
// the diagonal: A[j, i] = -cos * sin^j (j < i), A[i, i] = sin^i + perturb * eps_double * (m - i), zero below.
template <typename T>
__global__ void kahan_kernel(int64_t m, int64_t n, T* __restrict__ A, int64_t lda, double sn, double cs, double perturb) {
    const int64_t total = m * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e % m, i = e / m;      // row j, column i
        double v = 0;
        if (j < i) v = -cs * pow(sn, (double)j);
        else if (j == i) v = pow(sn, (double)i) + perturb * 2.220446049250313e-16 * (double)(m - i);
        A[j + i * lda] = (T)v;
    }
}

inline unsigned grid_for(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 65536); }

}  // namespace

#define CAPI(T, SUF)                                                                                                              \
    extern "C" int rlhip_scal_cols_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, const T* s_dev) {                  \
        if (!c || m < 0 || n < 0 || lda < (m > 1 ? m : 1)) return -2;                                                              \
        if (m == 0 || n == 0) return 0;                                                                                            \
        hipLaunchKernelGGL(scal_cols_kernel<T>, dim3(grid_for(m * n)), dim3(256), 0, c->stream, m, n, A, lda, s_dev);              \
        RLHIP_LAUNCH_CHECK();                                                                                                      \
        return 0;                                                                                                                  \
    }                                                                                                                              \
    extern "C" int rlhip_scal_rows_idx_##SUF(rlhip_ctx* c, int64_t cnt, const int64_t* idx_dev, int64_t n, T* A, int64_t lda, T alpha) { \
        if (!c || cnt < 0 || n < 0) return -2;                                                                                     \
        if (cnt == 0 || n == 0) return 0;                                                                                          \
        hipLaunchKernelGGL(scal_rows_idx_kernel<T>, dim3(grid_for(cnt * n)), dim3(256), 0, c->stream, cnt, idx_dev, n, A, lda, alpha); \
        RLHIP_LAUNCH_CHECK();                                                                                                      \
        return 0;                                                                                                                  \
    }                                                                                                                              \
    extern "C" int rlhip_gen_kahan_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T theta, T perturb) {              \
        if (!c || m < 0 || n < 0 || lda < (m > 1 ? m : 1)) return -2;                                                              \
        if (m == 0 || n == 0) return 0;                                                                                            \
        hipLaunchKernelGGL(kahan_kernel<T>, dim3(grid_for(m * n)), dim3(256), 0, c->stream, m, n, A, lda, (double)sin((T)theta),    \
                           (double)cos((T)theta), (double)perturb);                                                                \
        RLHIP_LAUNCH_CHECK();                                                                                                      \
        return 0;                                                                                                                  \
    }
CAPI(double, f64)
CAPI(float, f32)

// ---- pieces of the symmetric (Nystrom) path: linops::ExplicitSymLinOp, REVD2's error estimator (drivers/rl_revd2.hh:34-63)
namespace {
// F (n x n, full) from the `uplo` triangle of A; the other triangle of A is never read (it may hold NaNs, test_revd2.cc:123-128)
template <typename T>
__global__ void symmetrize_kernel(int upper, int64_t n, const T* __restrict__ A, int64_t lda, T* __restrict__ F, int64_t ldf) {
    const int64_t total = n * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % n, j = e / n;
        const bool stored = upper ? (i <= j) : (i >= j);
        F[i + j * ldf] = stored ? A[i + j * lda] : A[j + i * lda];
    }
}
template <typename T>
__global__ void axpby_kernel(int64_t n, T alpha, const T* __restrict__ x, T beta, T* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        y[e] = (beta == (T)0) ? alpha * x[e] : alpha * x[e] + beta * y[e];
}
}  // namespace

#define CAPI2(T, SUF)                                                                                                             \
    extern "C" int rlhip_symmetrize_##SUF(rlhip_ctx* c, char uplo, int64_t n, const T* A, int64_t lda, T* F, int64_t ldf) {       \
        const int up = (uplo == 'U' || uplo == 'u') ? 1 : (uplo == 'L' || uplo == 'l') ? 0 : -1;                                  \
        if (!c || up < 0 || n < 0 || lda < (n > 1 ? n : 1) || ldf < (n > 1 ? n : 1)) return -2;                                   \
        if (n == 0) return 0;                                                                                                     \
        hipLaunchKernelGGL(symmetrize_kernel<T>, dim3(grid_for(n * n)), dim3(256), 0, c->stream, up, n, A, lda, F, ldf);          \
        RLHIP_LAUNCH_CHECK();                                                                                                     \
        return 0;                                                                                                                 \
    }                                                                                                                             \
    extern "C" int rlhip_axpby_##SUF(rlhip_ctx* c, int64_t n, T alpha, const T* x, T beta, T* y) {                                \
        if (!c || n < 0) return -2;                                                                                               \
        if (n == 0) return 0;                                                                                                     \
        hipLaunchKernelGGL(axpby_kernel<T>, dim3(grid_for(n)), dim3(256), 0, c->stream, n, alpha, x, beta, y);                    \
        RLHIP_LAUNCH_CHECK();                                                                                                     \
        return 0;                                                                                                                 \
    }
CAPI2(double, f64)
CAPI2(float, f32)
