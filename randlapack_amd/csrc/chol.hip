// Device Cholesky factorisation A = U^T U (Uplo::Upper), replacing lapack::potrf on the CholQR steps
// (RandLAPACK/comps/rl_orth.hh:81, drivers/rl_cqrrpt.hh:311, drivers/rl_bqrrp.hh:462).
//
// Blocked right-looking, NB = 32:
//   per block step ONE panel kernel -- every workgroup re-factors the 32x32 diagonal block in LDS
//   (cheaper than a separate launch + a pass through memory), workgroup 0 publishes it, and each
//   workgroup then forward-substitutes its own slice of the block row  U12 = U11^-T A12, one column
//   per lane -- followed by the trailing update A22 -= U12^T U12 on the MFMA syrk path.
// A non-positive (or NaN) pivot stores its 1-based global index in a device flag; later kernels see
// the flag and leave the matrix untouched, so the caller gets LAPACK's info and a partially factored
// matrix, as with dpotrf.
#include "rlhip_internal.h"

namespace {

constexpr int NB = 32;

template <typename T>
__global__ __launch_bounds__(256) void potrf_panel_kernel(int64_t n, int64_t j0, int jb, T* __restrict__ A,
                                                          int64_t lda, int* __restrict__ info) {
    __shared__ T sU[NB][NB + 1];
    __shared__ int s_bad;
    if (*info != 0) return;
    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    // load diagonal block (upper part; lower part zero)
    for (int e = tid; e < NB * NB; e += 256) {
        int i = e % NB, j = e / NB;
        T v = 0;
        if (i < jb && j < jb && i <= j) v = A[(j0 + i) + (j0 + j) * lda];
        sU[i][j] = v;
    }
    __syncthreads();
    // in-LDS upper Cholesky: for k: u_kk = sqrt(a_kk); row k /= u_kk; trailing a_ij -= u_ki u_kj
    for (int k = 0; k < jb; ++k) {
        T d = sU[k][k];
        if (!(d > T(0))) {  // also catches NaN
            if (tid == 0) s_bad = k + 1;
        }
        __syncthreads();
        if (s_bad) break;
        T r = sqrt(d);
        __syncthreads();
        if (tid < jb) {
            int j = tid;
            if (j == k) sU[k][k] = r;
            else if (j > k) sU[k][j] = sU[k][j] / r;
        }
        __syncthreads();
        for (int e = tid; e < NB * NB; e += 256) {
            int i = e % NB, j = e / NB;
            if (i > k && j >= i && j < jb) sU[i][j] -= sU[k][i] * sU[k][j];
        }
        __syncthreads();
    }
    if (s_bad) {
        if (blockIdx.x == 0 && tid == 0) *info = (int)(j0 + s_bad);
        return;
    }
    if (blockIdx.x == 0) {
        for (int e = tid; e < NB * NB; e += 256) {
            int i = e % NB, j = e / NB;
            if (i < jb && j < jb && i <= j) A[(j0 + i) + (j0 + j) * lda] = sU[i][j];
        }
    }
    // block row: column c of A12 (one per thread): solve U11^T x = a
    int64_t c = j0 + jb + (int64_t)blockIdx.x * 256 + tid;
    if (c < n) {
        T x[NB];
        T* col = A + j0 + c * lda;
#pragma unroll
        for (int i = 0; i < NB; ++i) x[i] = (i < jb) ? col[i] : T(0);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i < jb) {
                T s = x[i];
#pragma unroll
                for (int l = 0; l < NB; ++l)
                    if (l < i) s -= sU[l][i] * x[l];
                x[i] = s / sU[i][i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (i < jb) col[i] = x[i];
    }
}

__global__ void zero_int_kernel(int* p) { *p = 0; }

}  // namespace

namespace rlhip {

template <typename T>
int gemm_impl(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
              int64_t lda, const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri, double* ssqA_dev = nullptr,
              int* ssq_done = nullptr);

template <typename T>
int potrf_upper(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int* info_host) {
    *info_host = 0;
    if (n < 0) return -2;
    if (lda < (n > 1 ? n : 1)) return -4;
    if (n == 0) return 0;
    int* d_info = (int*)(c->d_mail + 8);
    hipLaunchKernelGGL(zero_int_kernel, dim3(1), dim3(1), 0, c->stream, d_info);
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        int jb = (int)((n - j0 < NB) ? (n - j0) : NB);
        int64_t rest = n - j0 - jb;
        unsigned blocks = (unsigned)((rest + 255) / 256);
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(potrf_panel_kernel<T>, dim3(blocks), dim3(256), 0, c->stream, n, j0, jb, A, lda, d_info);
        RLHIP_LAUNCH_CHECK();
        if (rest > 0) {
            // A22 -= U12^T U12 (upper tiles only).  If the flag is set the panel kernel returned early and
            // this update works on unmodified data; the result is discarded by the caller (info > 0).
            const T* U12 = A + j0 + (j0 + jb) * lda;
            T* A22 = A + (j0 + jb) + (j0 + jb) * lda;
            int rc = gemm_impl<T>(c, 1, 0, rest, rest, jb, T(-1), U12, lda, U12, lda, T(1), A22, lda, 1);
            if (rc) return rc;
        }
    }
    RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 8, d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(hipStreamSynchronize(c->stream));
    *info_host = *(int*)(c->h_mail + 8);
    return 0;
}

template int potrf_upper<double>(rlhip_ctx*, int64_t, double*, int64_t, int*);
template int potrf_upper<float>(rlhip_ctx*, int64_t, float*, int64_t, int*);

}  // namespace rlhip
