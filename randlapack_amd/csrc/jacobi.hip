// Thin SVD by one-sided (Hestenes) Jacobi, entirely on the GPU.
//
// Replaces lapack::gesdd(Job::SomeVec, ...) in the RSVD tail (RandLAPACK/drivers/rl_rsvd.hh:146).  The
// reference runs that on the host CPU; here the small factor never leaves HBM.  One-sided Jacobi is
// chosen because (a) it is made of independent column-pair rotations -> one workgroup per pair, wavefront
// shuffle reductions for the three inner products, and (b) it computes small singular values to high
// RELATIVE accuracy (better than bidiagonalisation-based gesdd), so the parity tolerance on sigma is
// met with margin.
//
// Ordering: round-robin tournament (N-1 rounds of N/2 disjoint pairs per sweep), one launch per round;
// the launches of a sweep are independent of the data, so the host just enqueues them.  A device
// counter accumulates the number of rotations applied in a sweep; the host reads it once per sweep.
// (Column exchanges a la de Rijk were tried and REJECTED: with the tournament ordering they slow
// convergence from ~11 to >30 sweeps on graded 256-column factors.)  Callers should hand in a
// pre-conditioned factor: Jacobi on R^T of a QR factorisation converges in ~11 sweeps regardless of
// grading, on R itself it can take 30+ (measured).
#include "rlhip_internal.h"
#include <cmath>
#include <limits>

namespace {

template <typename T>
__device__ __forceinline__ T wsum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void jacobi_round_kernel(int64_t m, int n, int N, int round, T* __restrict__ A,
                                                           int64_t lda, T* __restrict__ V, int64_t ldv, T tol,
                                                           unsigned* __restrict__ nrot) {
    // pair of this workgroup (circle method)
    const int s = blockIdx.x;
    int p, q;
    if (s == 0) { p = N - 1; q = round % (N - 1); }
    else { p = (round + s) % (N - 1); q = (round - s + (N - 1)) % (N - 1); }
    if (p > q) { int t = p; p = q; q = t; }
    if (q >= n) return;  // phantom column of an odd-sized problem

    __shared__ T red[3][4];
    __shared__ T rot[3];
    T* __restrict__ ap = A + (int64_t)p * lda;
    T* __restrict__ aq = A + (int64_t)q * lda;
    T aa = 0, bb = 0, ab = 0;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        T x = ap[i], y = aq[i];
        aa += x * x; bb += y * y; ab += x * y;
    }
    aa = wsum(aa); bb = wsum(bb); ab = wsum(ab);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = aa; red[1][w] = bb; red[2][w] = ab; }
    __syncthreads();
    if (threadIdx.x == 0) {
        T alpha = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        T beta = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        T gamma = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        T cs = 1, sn = 0, swap = 0;
        T lim = tol * sqrt(alpha) * sqrt(beta);
        if (fabs(gamma) > lim && lim >= T(0) && alpha > T(0) && beta > T(0)) {
            T zeta = (beta - alpha) / (T(2) * gamma);
            T t = (zeta >= 0 ? T(1) : T(-1)) / (fabs(zeta) + sqrt(T(1) + zeta * zeta));
            cs = T(1) / sqrt(T(1) + t * t);
            sn = cs * t;
            atomicAdd(nrot, 1u);
        }
        rot[0] = cs; rot[1] = sn; rot[2] = swap;
    }
    __syncthreads();
    const T cs = rot[0], sn = rot[1];
    const bool swap = rot[2] != T(0);
    if (cs == T(1) && sn == T(0) && !swap) return;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        T x = ap[i], y = aq[i];
        T xn = cs * x - sn * y, yn = sn * x + cs * y;
        ap[i] = swap ? yn : xn;
        aq[i] = swap ? xn : yn;
    }
    T* __restrict__ vp = V + (int64_t)p * ldv;
    T* __restrict__ vq = V + (int64_t)q * ldv;
    for (int i = threadIdx.x; i < n; i += 256) {
        T x = vp[i], y = vq[i];
        T xn = cs * x - sn * y, yn = sn * x + cs * y;
        vp[i] = swap ? yn : xn;
        vq[i] = swap ? xn : yn;
    }
}

// column norms -> S (unsorted), one workgroup per column
template <typename T>
__global__ __launch_bounds__(256) void colnorm_kernel(int64_t m, const T* __restrict__ A, int64_t lda,
                                                      T* __restrict__ S) {
    __shared__ T red[4];
    const T* col = A + (int64_t)blockIdx.x * lda;
    // scaled accumulation is unnecessary here: entries are bounded by the singular values themselves
    T acc = 0;
    for (int64_t i = threadIdx.x; i < m; i += 256) acc += col[i] * col[i];
    acc = wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) S[blockIdx.x] = sqrt(red[0] + red[1] + red[2] + red[3]);
}

// rank[j] = position of column j in descending order of S (stable)
template <typename T>
__global__ void rank_kernel(int n, const T* __restrict__ S, int* __restrict__ rank) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    T sj = S[j];
    int r = 0;
    for (int i = 0; i < n; ++i) {
        T si = S[i];
        r += (si > sj) || (si == sj && i < j);
    }
    rank[j] = r;
}

// scatter normalised columns into sorted position: Uout[:, rank[j]] = A[:, j] / S[j]; VT[rank[j], :] = V[:, j]^T
template <typename T>
__global__ __launch_bounds__(256) void finalize_kernel(int64_t m, int n, const T* __restrict__ A, int64_t lda,
                                                       const T* __restrict__ V, int64_t ldv,
                                                       const T* __restrict__ S, const int* __restrict__ rank,
                                                       T* __restrict__ Uout, int64_t ldu, T* __restrict__ Sout,
                                                       T* __restrict__ VT, int64_t ldvt) {
    const int j = blockIdx.x;
    const int r = rank[j];
    const T s = S[j];
    const T inv = (s > T(0)) ? T(1) / s : T(0);
    const T* col = A + (int64_t)j * lda;
    T* dst = Uout + (int64_t)r * ldu;
    for (int64_t i = threadIdx.x; i < m; i += 256) dst[i] = col[i] * inv;
    const T* vcol = V + (int64_t)j * ldv;
    for (int i = threadIdx.x; i < n; i += 256) VT[r + (int64_t)i * ldvt] = vcol[i];
    if (threadIdx.x == 0) Sout[r] = s;
}

__global__ void zero_u32_kernel(unsigned* p) { *p = 0; }

}  // namespace

namespace rlhip {

template <typename T>
int lacpy(rlhip_ctx* c, int uplo, int64_t m, int64_t n, const T* A, int64_t lda, T* B, int64_t ldb);
template <typename T>
int laset(rlhip_ctx* c, int uplo, int64_t m, int64_t n, T offdiag, T diag, T* A, int64_t lda);

template <typename T>
int gesvdj(rlhip_ctx* c, int64_t m, int64_t n64, T* A, int64_t lda, T* S, T* VT, int64_t ldvt,
           int* sweeps_host) {
    if (m < 0) return -2;
    if (n64 < 0) return -3;
    if (m < n64) return -2;  // tall only (the path's factor is n x k with n >= k)
    if (lda < (m > 1 ? m : 1)) return -5;
    if (ldvt < (n64 > 1 ? n64 : 1)) return -8;
    if (sweeps_host) *sweeps_host = 0;
    if (n64 == 0) return 0;
    const int n = (int)n64;
    const int N = (n % 2) ? n + 1 : n;
    size_t mark = rlhip_ws_mark(c);
    T* V = ws_alloc<T>(c, (size_t)n * n);
    T* W = ws_alloc<T>(c, (size_t)m * n);
    T* Sraw = ws_alloc<T>(c, (size_t)n);
    int* rank = ws_alloc<int>(c, (size_t)n);
    if (!V || !W || !Sraw || !rank) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    unsigned* d_nrot = (unsigned*)(c->d_mail + 16);
    int rc = laset<T>(c, 2, n, n, T(0), T(1), V, n);
    if (rc) { rlhip_ws_release(c, mark); return rc; }
    const T tol = std::sqrt((T)m) * std::numeric_limits<T>::epsilon();
    const int max_sweeps = 60;
    int sweep = 0;
    int info = 0;
    if (n > 1) {
        for (; sweep < max_sweeps; ++sweep) {
            hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, c->stream, d_nrot);
            for (int round = 0; round < N - 1; ++round) {
                hipLaunchKernelGGL(jacobi_round_kernel<T>, dim3(N / 2), dim3(256), 0, c->stream, m, n, N, round, A,
                                   lda, V, (int64_t)n, tol, d_nrot);
            }
            RLHIP_LAUNCH_CHECK();
            RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 16, d_nrot, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            RLHIP_CHECK(hipStreamSynchronize(c->stream));
            unsigned nrot = *(unsigned*)(c->h_mail + 16);
            if (nrot == 0) { ++sweep; break; }
        }
        if (sweep >= max_sweeps) info = 1;
    }
    if (sweeps_host) *sweeps_host = sweep;
    hipLaunchKernelGGL(colnorm_kernel<T>, dim3(n), dim3(256), 0, c->stream, m, A, lda, Sraw);
    hipLaunchKernelGGL(rank_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, Sraw, rank);
    hipLaunchKernelGGL(finalize_kernel<T>, dim3(n), dim3(256), 0, c->stream, m, n, A, lda, V, (int64_t)n, Sraw, rank,
                       W, m, S, VT, ldvt);
    RLHIP_LAUNCH_CHECK();
    rc = lacpy<T>(c, 2, m, n, W, m, A, lda);
    rlhip_ws_release(c, mark);
    return rc ? rc : info;
}

template int gesvdj<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*, double*, int64_t, int*);
template int gesvdj<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*, float*, int64_t, int*);

}  // namespace rlhip
