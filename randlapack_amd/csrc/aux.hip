// HBM-bound helper kernels: norms, copies, fills.  Lanes always run along the contiguous (row)
// direction of the column-major operands so every wavefront touches whole 512-byte segments.
#include <cstring>
#include "rlhip_internal.h"

namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-level sum, result valid in thread 0
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T* smem /* NT/64 elements */) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) smem[w] = v;
    __syncthreads();
    T s = 0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < NT / 64; ++i) s += smem[i];
    }
    __syncthreads();
    return s;
}

// stage 1: one partial sum of squares per block (fixed grid -> deterministic)
template <typename T>
__global__ __launch_bounds__(256) void ssq_partial_kernel(int64_t m, int64_t n, const T* __restrict__ A,
                                                          int64_t lda, double* __restrict__ partial) {
    __shared__ double sm[4];
    double acc = 0;
    // each block walks columns in a strided fashion; threads walk rows
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
        const T* col = A + j * lda;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
            double v = (double)col[i];
            acc += v * v;
        }
    }
    double s = block_sum<double, 256>(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// scaled variant (dlassq's safeguard, taken only when the plain sum over- or underflowed): partial sums of (x * inv_scale)^2, and the
// largest |x| of the matrix by an integer max on the bit patterns (non-negative doubles order like unsigned integers)
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(int64_t m, int64_t n, const T* __restrict__ A, int64_t lda,
                                                     unsigned long long* __restrict__ out) {
    double mx = 0;
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
        const T* col = A + j * lda;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
            const double v = fabs((double)col[i]);
            if (v > mx || v != v) mx = v;                      // NaN propagates
        }
    }
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off, 64); if (o > mx || o != o) mx = o; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
}
template <typename T>
__global__ __launch_bounds__(256) void ssq_scaled_partial_kernel(int64_t m, int64_t n, const T* __restrict__ A, int64_t lda,
                                                                 double scale, double* __restrict__ partial) {
    __shared__ double sm[4];
    double acc = 0;
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
        const T* col = A + j * lda;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
            const double v = (double)col[i] / scale;
            acc += v * v;
        }
    }
    double s = block_sum<double, 256>(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void ssq_final_kernel(int np, const double* __restrict__ partial,
                                                        double* __restrict__ out) {
    __shared__ double sm[4];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;             // four loads in flight per thread (fixed order: deterministic)
    int i = threadIdx.x;
    for (; i + 768 < np; i += 1024) { a0 += partial[i]; a1 += partial[i + 256]; a2 += partial[i + 512]; a3 += partial[i + 768]; }
    for (; i < np; i += 256) a0 += partial[i];
    double s = block_sum<double, 256>((a0 + a1) + (a2 + a3), sm);
    if (threadIdx.x == 0) out[0] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void lacpy_kernel(int uplo, int64_t m, int64_t n, const T* __restrict__ A,
                                                    int64_t lda, T* __restrict__ B, int64_t ldb) {
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
            bool ok = (uplo == 2) || (uplo == 0 && i <= j) || (uplo == 1 && i >= j);
            if (ok) B[i + j * ldb] = A[i + j * lda];
        }
    }
}

// LAPACK laset semantics: 'U' sets the strictly upper part to offd, 'L' the strictly lower part, 'G'
// everything; the first min(m,n) diagonal entries are set to diag in all three cases.
template <typename T>
__global__ __launch_bounds__(256) void laset_kernel(int uplo, int64_t m, int64_t n, T offd, T diag,
                                                    T* __restrict__ A, int64_t lda) {
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
            if (i == j) A[i + j * lda] = diag;
            else if (uplo == 2 || (uplo == 0 && i < j) || (uplo == 1 && i > j)) A[i + j * lda] = offd;
        }
    }
}

template <typename T>
__global__ void add_diag_kernel(int64_t n, T alpha, T* __restrict__ A, int64_t lda) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i + i * lda] += alpha;
}

inline dim3 grid2d(int64_t m, int64_t n) {
    int64_t gx = (m + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    int64_t gy = n;
    if (gy > 1024) gy = 1024;
    if (gy < 1) gy = 1;
    return dim3((unsigned)gx, (unsigned)gy);
}

}  // namespace

namespace rlhip {

template <typename T>
int lange_fro(rlhip_ctx* c, int64_t m, int64_t n, const T* A, int64_t lda, T* result_host) {
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (m == 0 || n == 0) { *result_host = T(0); return 0; }
    dim3 grid = grid2d(m, n);
    // at most ~2048 partial sums: the one-workgroup second stage walked 16384 of them one dependent load at a time (17 us of a 30 us norm
    // of the 20000 x 256 factor B^T of the RSVD tail)
    if ((int64_t)grid.x * grid.y > 2048) grid.y = (unsigned)(2048 / grid.x < 1 ? 1 : 2048 / grid.x);
    int np = (int)(grid.x * grid.y);
    size_t mark = rlhip_ws_mark(c);
    double* partial = ws_alloc<double>(c, np);
    if (!partial) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    hipLaunchKernelGGL(ssq_partial_kernel<T>, grid, dim3(256), 0, c->stream, m, n, A, lda, partial);
    RLHIP_LAUNCH_CHECK();
    double* d_out = (double*)c->d_mail;
    hipLaunchKernelGGL(ssq_final_kernel, dim3(1), dim3(256), 0, c->stream, np, partial, d_out);
    RLHIP_LAUNCH_CHECK();
    RLHIP_CHECK(hipMemcpyAsync(c->h_mail, d_out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(rlhip_stream_sync(c));
    double ssq = *(double*)c->h_mail;
    if (!(ssq > 0.0) || ssq > 1.7e308) {
        // all zeros, NaN, or the plain sum of squares over- / underflowed (entries beyond ~1e154 or below ~1e-154): redo it the way
        // LAPACK's dlassq does, relative to the largest entry -- one more pass, taken only here
        unsigned long long* d_mx = (unsigned long long*)(c->d_mail + 1);
        RLHIP_CHECK(hipMemsetAsync(d_mx, 0, sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(absmax_kernel<T>, grid, dim3(256), 0, c->stream, m, n, A, lda, d_mx);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 1, d_mx, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        double mx;
        memcpy(&mx, c->h_mail + 1, sizeof(double));
        if (mx != mx || mx > 1.7e308) { rlhip_ws_release(c, mark); *result_host = (T)mx; return 0; }     // NaN / inf entries: that is the norm
        if (mx == 0.0) { rlhip_ws_release(c, mark); *result_host = T(0); return 0; }
        hipLaunchKernelGGL(ssq_scaled_partial_kernel<T>, grid, dim3(256), 0, c->stream, m, n, A, lda, mx, partial);
        hipLaunchKernelGGL(ssq_final_kernel, dim3(1), dim3(256), 0, c->stream, np, partial, d_out);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail, d_out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        rlhip_ws_release(c, mark);
        *result_host = (T)(mx * sqrt(*(double*)c->h_mail));
        return 0;
    }
    rlhip_ws_release(c, mark);
    *result_host = (T)sqrt(ssq);
    return 0;
}

template <typename T>
int lacpy(rlhip_ctx* c, int uplo, int64_t m, int64_t n, const T* A, int64_t lda, T* B, int64_t ldb) {
    if (m <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(lacpy_kernel<T>, grid2d(m, n), dim3(256), 0, c->stream, uplo, m, n, A, lda, B, ldb);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int laset(rlhip_ctx* c, int uplo, int64_t m, int64_t n, T offd, T diag, T* A, int64_t lda) {
    if (m <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(laset_kernel<T>, grid2d(m, n), dim3(256), 0, c->stream, uplo, m, n, offd, diag, A, lda);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int add_diag(rlhip_ctx* c, int64_t n, T alpha, T* A, int64_t lda) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_diag_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, alpha, A, lda);
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template int add_diag<double>(rlhip_ctx*, int64_t, double, double*, int64_t);
template int add_diag<float>(rlhip_ctx*, int64_t, float, float*, int64_t);

template int lange_fro<double>(rlhip_ctx*, int64_t, int64_t, const double*, int64_t, double*);
template int lange_fro<float>(rlhip_ctx*, int64_t, int64_t, const float*, int64_t, float*);
template int lacpy<double>(rlhip_ctx*, int, int64_t, int64_t, const double*, int64_t, double*, int64_t);
template int lacpy<float>(rlhip_ctx*, int, int64_t, int64_t, const float*, int64_t, float*, int64_t);
template int laset<double>(rlhip_ctx*, int, int64_t, int64_t, double, double, double*, int64_t);
template int laset<float>(rlhip_ctx*, int, int64_t, int64_t, float, float, float*, int64_t);

}  // namespace rlhip
