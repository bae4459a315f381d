// Thin SVD of a tall factor, drop-in for lapack::gesdd(Job::SomeVec, m, n, A, lda, S, U, ldu, VT, ldvt) at
// RandLAPACK/drivers/rl_rsvd.hh:146 (m = n_cols(A) of the data matrix, n = target rank k there).
//
// Pipeline (all on device):
//   1. Cholesky-QR twice:  A = Q1 R1, Q1 = Q2 R2  ->  A = Q (R2 R1)      [MFMA syrk + potrf + trsm]
//   2. X = (R2 R1)^T  (n x n, lower triangular)                          [LDS-tiled transpose]
//   3. one-sided Jacobi on X:  X = Ux S Vx^T   (converges in ~11 sweeps on R^T; 30+ on R itself)
//   4. A = Q R = Q X^T = (Q Vx) S Ux^T   ->   U_out = Q Vx [MFMA gemm],  VT_out = Ux^T [transpose]
// If either Cholesky fails or diag(R1) spans more than 1e7 (CholQR2 no longer guaranteed orthonormal)
// the routine falls back to Jacobi on A itself (slower, unconditionally accurate).
#include "rlhip_internal.h"
#include <limits>

namespace {

// out[j + i*ldo] = in[i + j*ldi]; 64x64 tiles through LDS, both sides coalesced
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int64_t m, int64_t n, const T* __restrict__ in, int64_t ldi,
                                                        T* __restrict__ out, int64_t ldo, int upper_only) {
    __shared__ T tile[64][65];
    const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int jj = ty; jj < 64; jj += 4) {
        int64_t i = i0 + tx, j = j0 + jj;
        T v = 0;
        if (i < m && j < n && (!upper_only || i <= j)) v = in[i + j * ldi];
        tile[jj][tx] = v;
    }
    __syncthreads();
    for (int ii = ty; ii < 64; ii += 4) {
        int64_t i = i0 + ii, j = j0 + tx;
        if (i < m && j < n && (!upper_only || i <= j)) out[j + i * ldo] = tile[tx][ii];
    }
}

// max|d_i| / min|d_i| over the diagonal of an n x n matrix -> out[0] (one workgroup)
template <typename T>
__global__ __launch_bounds__(256) void diag_ratio_kernel(int n, const T* __restrict__ R, int64_t ldr,
                                                         double* __restrict__ out) {
    __shared__ double smax[256], smin[256];
    double mx = 0, mn = 1e300;
    for (int i = threadIdx.x; i < n; i += 256) {
        double v = fabs((double)R[i + (int64_t)i * ldr]);
        mx = v > mx ? v : mx;
        mn = v < mn ? v : mn;
    }
    smax[threadIdx.x] = mx; smin[threadIdx.x] = mn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            smax[threadIdx.x] = smax[threadIdx.x] > smax[threadIdx.x + s] ? smax[threadIdx.x] : smax[threadIdx.x + s];
            smin[threadIdx.x] = smin[threadIdx.x] < smin[threadIdx.x + s] ? smin[threadIdx.x] : smin[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (smin[0] > 0) ? smax[0] / smin[0] : 1e300;
}

}  // namespace

namespace rlhip {

template <typename T>
int transpose(rlhip_ctx* c, int64_t m, int64_t n, const T* A, int64_t lda, T* AT, int64_t ldat, int upper_only) {
    if (m <= 0 || n <= 0) return 0;
    dim3 grid((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, c->stream, m, n, A, lda, AT, ldat, upper_only);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int gesdd_tall(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* S, T* U, int64_t ldu, T* VT, int64_t ldvt,
               int* sweeps_host) {
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (m < n) return -2;
    if (sweeps_host) *sweeps_host = 0;
    if (n == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* R1 = ws_alloc<T>(c, (size_t)n * n);
    T* R2 = ws_alloc<T>(c, (size_t)n * n);
    T* X = ws_alloc<T>(c, (size_t)n * n);
    T* VTx = ws_alloc<T>(c, (size_t)n * n);
    if (!R1 || !R2 || !X || !VTx) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    int rc = 0, info = 0;
    bool fallback = false;
    // ---- pass 1
    rc = laset<T>(c, 2, n, n, T(0), T(0), R1, n);
    if (!rc) rc = syrk<T>(c, Upper, 1, n, m, T(1), A, lda, T(0), R1, n);
    if (!rc) rc = potrf_upper<T>(c, n, R1, n, &info);
    if (rc) { rlhip_ws_release(c, mark); return rc; }
    if (info) fallback = true;
    if (!fallback) {
        double* d_ratio = (double*)(c->d_mail + 24);
        hipLaunchKernelGGL(diag_ratio_kernel<T>, dim3(1), dim3(256), 0, c->stream, (int)n, R1, (int64_t)n, d_ratio);
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 24, d_ratio, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(hipStreamSynchronize(c->stream));
        double ratio = *(double*)(c->h_mail + 24);
        const double lim = (sizeof(T) == 8) ? 1e7 : 1e3;
        if (!(ratio < lim)) fallback = true;
    }
    if (!fallback) {
        rc = trsm_right_upper<T>(c, NonUnit, m, n, T(1), R1, n, A, lda);
        // ---- pass 2
        if (!rc) rc = laset<T>(c, 2, n, n, T(0), T(0), R2, n);
        if (!rc) rc = syrk<T>(c, Upper, 1, n, m, T(1), A, lda, T(0), R2, n);
        if (!rc) rc = potrf_upper<T>(c, n, R2, n, &info);
        if (rc) { rlhip_ws_release(c, mark); return rc; }
        if (info) {
            // undo pass 1 on A (A = Q1 R1) and take the robust route
            rc = trmm_right_upper<T>(c, NonUnit, m, n, T(1), R1, n, A, lda);
            if (rc) { rlhip_ws_release(c, mark); return rc; }
            fallback = true;
        }
    }
    if (fallback) {
        // Jacobi on A directly: A -> U_A, then copy out
        int sw = 0;
        int jinfo = gesvdj<T>(c, m, n, A, lda, S, VT, ldvt, &sw);
        if (sweeps_host) *sweeps_host = sw;
        if (jinfo < 0) { rlhip_ws_release(c, mark); return jinfo; }
        rc = lacpy<T>(c, 2, m, n, A, lda, U, ldu);
        rlhip_ws_release(c, mark);
        return rc ? rc : jinfo;
    }
    rc = trsm_right_upper<T>(c, NonUnit, m, n, T(1), R2, n, A, lda);   // A now holds Q (orthonormal)
    // R = R2 * R1 (upper triangles only): zero the strictly lower parts first, then R2 <- R2 * R1
    if (!rc) rc = laset<T>(c, 1, n - 1, n, T(0), T(0), R2 + 1, n);      // 'L' incl. diag of the (n-1) x n block below row 0
    if (!rc) rc = trmm_right_upper<T>(c, NonUnit, n, n, T(1), R1, n, R2, n);
    if (!rc) rc = laset<T>(c, 2, n, n, T(0), T(0), X, n);
    if (!rc) rc = transpose<T>(c, n, n, R2, n, X, n, 1);               // X = R^T (lower triangular)
    if (rc) { rlhip_ws_release(c, mark); return rc; }
    int sw = 0;
    int jinfo = gesvdj<T>(c, n, n, X, n, S, VTx, n, &sw);              // X = Ux S VTx
    if (sweeps_host) *sweeps_host = sw;
    if (jinfo < 0) { rlhip_ws_release(c, mark); return jinfo; }
    // U_out = Q * Vx = Q * VTx^T ;  VT_out = Ux^T
    rc = gemm<T>(c, 0, 1, m, n, n, T(1), A, lda, VTx, n, T(0), U, ldu);
    if (!rc) rc = transpose<T>(c, n, n, X, n, VT, ldvt, 0);
    rlhip_ws_release(c, mark);
    return rc ? rc : jinfo;
}

template int transpose<double>(rlhip_ctx*, int64_t, int64_t, const double*, int64_t, double*, int64_t, int);
template int transpose<float>(rlhip_ctx*, int64_t, int64_t, const float*, int64_t, float*, int64_t, int);
template int gesdd_tall<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*, double*, int64_t, double*, int64_t, int*);
template int gesdd_tall<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*, float*, int64_t, float*, int64_t, int*);

}  // namespace rlhip
