// Householder reconstruction and block-reflector application for the BQRRP / HQRRP part of the path.
//
//   orhr_col   lapack::orhr_col (drivers/rl_bqrrp.hh:480, rl_hqrrp.hh:537), util::rl_orhr_col (misc/rl_util.hh:339-379),
//              the reference's CUDA orhr_col_gpu = 3*n launches of BLAS-1/2 kernels (rl_cuda_kernels.cuh:772-803).
//              Here: LAPACK's dorhr_col organisation -- sign-modified LU without pivoting of the top n x n block
//              (blocked, NB = 32: one fused "diagonal block + block row" kernel and one MFMA update per step),
//              ONE wide triangular solve V2 = Q2 * U^-1 for all rows below (MFMA path of tri.hip), and the
//              T factors as  T = (-U * diag(D)) * V1^-T  per nb-block.
//   gemqrt_lt  lapack::gemqrt(Side::Left, Op::Trans, ...) (rl_bqrrp.hh:543): C <- Q^T C with the compact-WY
//              blocks (V, T).  Three dense contractions per block on the MFMA GEMMs:
//                  W = V^T C ;  W <- T^T W ;  C <- C - V W
//              V's top nb x nb block is read through a cleaned unit-lower copy, the rest of V in place.
//   larft_gram T from (V, tau) in LAPACK geqrf format, via  T^-1 = striu(V^T V) + diag(1/tau)  and one triangular
//              solve -- lets the same block apply serve lapack::ormqr (rl_bqrrp.hh:545).
#include "rlhip_internal.h"
#include <vector>
#include <limits>
#include <cmath>
#include <cstdlib>

namespace rlhip {
template <typename T>
int gemm_impl(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
              const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri, double* ssqA_dev = nullptr, int* ssq_done = nullptr);
}

namespace {

constexpr int LB = 32;

// Sign-modified LU step (dlaorhr_col_getrfnp semantics) on the diagonal block [j0, j0+jb) and its block row:
//   for i: D(i) = -sign(a_ii) (1 if a_ii == 0); a_ii -= D(i); column below /= a_ii; trailing -= col * row
// Every workgroup re-factors the jb x jb block in LDS; workgroup 0 writes it back (+ D); every workgroup then
// computes its slice of U12 = L11^-1 A12 (one column per thread).
template <typename T>
__global__ __launch_bounds__(256) void lunp_panel_kernel(int64_t n, int64_t j0, int jb, T* __restrict__ A, int64_t lda,
                                                         T* __restrict__ D) {
    __shared__ T sA[LB][LB + 1];
    __shared__ T sD[LB];
    const int tid = threadIdx.x;
    for (int e = tid; e < LB * LB; e += 256) {
        int i = e % LB, j = e / LB;
        sA[i][j] = (i < jb && j < jb) ? A[(j0 + i) + (j0 + j) * lda] : T(0);
    }
    __syncthreads();
    for (int k = 0; k < jb; ++k) {
        if (tid == 0) {
            T a = sA[k][k];
            T dd = (a == T(0)) ? T(1) : ((a > T(0)) ? T(-1) : T(1));
            sD[k] = dd;
            sA[k][k] = a - dd;
        }
        __syncthreads();
        const T piv = sA[k][k];
        if (tid > k && tid < jb) sA[tid][k] = sA[tid][k] / piv;
        __syncthreads();
        for (int e = tid; e < LB * LB; e += 256) {
            int i = e % LB, j = e / LB;
            if (i > k && j > k && i < jb && j < jb) sA[i][j] -= sA[i][k] * sA[k][j];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int e = tid; e < LB * LB; e += 256) {
            int i = e % LB, j = e / LB;
            if (i < jb && j < jb) A[(j0 + i) + (j0 + j) * lda] = sA[i][j];
        }
        if (tid < jb) D[j0 + tid] = sD[tid];
    }
    // U12: column c of A12, forward substitution with unit-lower L11
    int64_t c = j0 + jb + (int64_t)blockIdx.x * 256 + tid;
    if (c < n) {
        T x[LB];
        T* col = A + j0 + c * lda;
#pragma unroll
        for (int i = 0; i < LB; ++i) { const T t = col[(i < jb) ? i : (jb - 1)]; x[i] = (i < jb) ? t : T(0); }   // clamped, not branched
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            if (i < jb) {
                T s = x[i];
#pragma unroll
                for (int l = 0; l < LB; ++l)
                    if (l < i) s -= sA[i][l] * x[l];
                x[i] = s;
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i)
            if (i < jb) col[i] = x[i];
    }
}

// T(0:i+1, j) = -D(j) * U(jb0 : jb0+i+1, j) for the columns of one nb-block (upper triangle), zero below
template <typename T>
__global__ void tfac_init_kernel(int64_t n, int64_t nb, const T* __restrict__ A, int64_t lda, const T* __restrict__ D,
                                 T* __restrict__ Tm, int64_t ldt) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nb * n) return;
    int64_t i = idx % nb, j = idx / nb;            // T is nb x n
    int64_t jb0 = (j / nb) * nb;                   // block start
    int64_t jl = j - jb0;                          // column inside the block
    T v = 0;
    if (i <= jl) v = -D[j] * A[(jb0 + i) + j * lda];
    Tm[i + j * ldt] = v;
}

// unit-lower-triangular nb x nb block of V -> dense copy with explicit ones / zeros (dst ld = nb)
template <typename T>
__global__ void unit_lower_copy_kernel(int64_t nb, const T* __restrict__ V, int64_t ldv, T* __restrict__ out, int transpose) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nb * nb) return;
    int64_t i = idx % nb, j = idx / nb;
    T v = (i > j) ? V[i + j * ldv] : ((i == j) ? T(1) : T(0));
    if (transpose) out[j + i * nb] = v; else out[i + j * nb] = v;
}

// R(j, i) *= D(j) for j <= i (row scaling of an upper-triangular n x n matrix)   rl_bqrrp.hh:485-487
template <typename T>
__global__ void row_sign_kernel(int64_t n, T* __restrict__ R, int64_t ldr, const T* __restrict__ D) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int64_t j = idx % n, i = idx / n;
    if (j <= i) R[j + i * ldr] *= D[j];
}

// tau(i) = T(i % nb, i)                                                            rl_bqrrp.hh:490-491
template <typename T>
__global__ void tau_from_t_kernel(int64_t k, int64_t nb, const T* __restrict__ Tm, int64_t ldt, T* __restrict__ tau) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) tau[i] = Tm[(i % nb) + i * ldt];
}

// M = striu(G) + diag(1/tau); tau == 0 (H = I) -> a huge diagonal so that the column of T vanishes
template <typename T>
__global__ void larft_m_kernel(int64_t k, T* __restrict__ G, int64_t ldg, const T* __restrict__ tau) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= k * k) return;
    int64_t i = idx % k, j = idx / k;
    T v = G[i + j * ldg];
    if (i > j) v = 0;
    else if (i == j) v = (tau[i] != T(0)) ? T(1) / tau[i] : T(1e300);
    G[i + j * ldg] = v;
}

// flag |= any(|x[i]| > thr), i < n                                                  rl_bqrrp.hh:373-379
template <typename T>
__global__ void any_abs_gt_kernel(int64_t n, const T* __restrict__ x, T thr, int* __restrict__ flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = (i < n) && (fabs(x[i]) > thr);
    if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ void zero_int2(int* p) { *p = 0; }

}  // namespace

namespace rlhip {

template <typename T> int laset(rlhip_ctx*, int, int64_t, int64_t, T, T, T*, int64_t);
template <typename T> int lacpy(rlhip_ctx*, int, int64_t, int64_t, const T*, int64_t, T*, int64_t);
template <typename T> int trsm_right_upper(rlhip_ctx*, int, int64_t, int64_t, T, const T*, int64_t, T*, int64_t);
template <typename T> int gemm(rlhip_ctx*, int, int, int64_t, int64_t, int64_t, T, const T*, int64_t, const T*, int64_t, T, T*, int64_t);

template <typename T>
int lunp_blk(rlhip_ctx* c, int64_t n, T* A, int64_t lda, T* D);

template <typename T>
static int lunp_top(rlhip_ctx* c, int64_t n, T* A, int64_t lda, T* D);

// A (m x n, orthonormal columns) -> V (unit lower trapezoidal, in place), T (nb x n), D (n)
template <typename T>
int orhr_col(rlhip_ctx* c, int64_t m, int64_t n, int64_t nb, T* A, int64_t lda, T* Tm, int64_t ldt, T* D) {
    if (m < 0) return -2;
    if (n < 0 || n > m) return -3;
    if (nb < 1) return -4;
    if (lda < (m > 1 ? m : 1)) return -6;
    if (nb > n) nb = n;
    if (ldt < (nb > 1 ? nb : 1)) return -8;
    if (n == 0) return 0;
    {
        const int rc1 = lunp_top<T>(c, n, A, lda, D);
        if (rc1) return rc1;
    }
    // (2) V2 = Q2 * U^-1 for the rows below the top block
    if (m > n) {
        int rc = trsm_right_upper<T>(c, 0, m - n, n, T(1), A, lda, A + n, lda);
        if (rc) return rc;
    }
    // (3) T = (-U diag(D)) V1^-T, block by block
    hipLaunchKernelGGL(tfac_init_kernel<T>, dim3((unsigned)((nb * n + 255) / 256)), dim3(256), 0, c->stream, n, nb, A, lda, D,
                       Tm, ldt);
    RLHIP_LAUNCH_CHECK();
    size_t mark = rlhip_ws_mark(c);
    T* Lt = ws_alloc<T>(c, (size_t)nb * nb);
    if (!Lt) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    for (int64_t jb0 = 0; jb0 < n; jb0 += nb) {
        const int64_t jnb = (n - jb0 < nb) ? (n - jb0) : nb;
        hipLaunchKernelGGL(unit_lower_copy_kernel<T>, dim3((unsigned)((jnb * jnb + 255) / 256)), dim3(256), 0, c->stream, jnb,
                           A + jb0 + jb0 * lda, lda, Lt, 1);   // Lt = V1^T (unit upper)
        RLHIP_LAUNCH_CHECK();
        int rc = trsm_right_upper<T>(c, 1, jnb, jnb, T(1), Lt, jnb, Tm + jb0 * ldt, ldt);
        if (rc) { rlhip_ws_release(c, mark); return rc; }
    }
    rlhip_ws_release(c, mark);
    return 0;
}

// (1) of orhr_col: the sign-modified LU of the top n x n block (in place: unit lower L below, U on and above the diagonal), D = its sign
// vector -- one launch of the block-pipelined kernel when it fits (qr_blk.hip), else 32-column panels
template <typename T>
static int lunp_top(rlhip_ctx* c, int64_t n, T* A, int64_t lda, T* D) {
    // n <= one panel (ABRIK's 32-column Krylov blocks): ONE launch of the LDS panel kernel (~10 us) instead of the cooperative launch (42 us)
    const int rb = (n <= LB) ? 0 : lunp_blk<T>(c, n, A, lda, D);
    if (rb < 0) return rb;
    for (int64_t j0 = (rb == 1) ? n : 0; j0 < n; j0 += LB) {
        const int jb = (int)((n - j0 < LB) ? (n - j0) : LB);
        const int64_t rest = n - j0 - jb;
        unsigned blocks = (unsigned)((rest + 255) / 256);
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(lunp_panel_kernel<T>, dim3(blocks), dim3(256), 0, c->stream, n, j0, jb, A, lda, D);
        RLHIP_LAUNCH_CHECK();
        if (rest > 0) {
            // L21 = A21 * U11^-1 (rows j0+jb .. n-1)
            int rc = trsm_right_upper<T>(c, 0, rest, jb, T(1), A + j0 + j0 * lda, lda, A + (j0 + jb) + j0 * lda, lda);
            if (rc) return rc;
            // A22 -= L21 * U12
            rc = gemm<T>(c, 0, 0, rest, rest, jb, T(-1), A + (j0 + jb) + j0 * lda, lda, A + j0 + (j0 + jb) * lda, lda, T(1),
                         A + (j0 + jb) + (j0 + jb) * lda, lda);
            if (rc) return rc;
        }
    }
    return 0;
}

// C (m x n) <- Q^T C,  Q = H_1 ... H_k in compact-WY blocks of width nb: V (m x k, unit lower trapezoidal, only the
// strictly lower part is read), T (nb x k).  Side::Left, Op::Trans.
template <typename T>
int gemqrt_lt(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, int64_t nb, const T* V, int64_t ldv, const T* Tm, int64_t ldt,
              T* C, int64_t ldc) {
    if (m < 0) return -3;
    if (n < 0) return -4;
    if (k < 0 || k > m) return -5;
    if (nb < 1) return -6;
    if (k == 0 || n == 0 || m == 0) return 0;
    if (nb > k) nb = k;
    size_t mark = rlhip_ws_mark(c);
    T* V1 = ws_alloc<T>(c, (size_t)nb * nb);
    T* W = ws_alloc<T>(c, (size_t)nb * n);
    T* W2 = ws_alloc<T>(c, (size_t)nb * n);
    if (!V1 || !W || !W2) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    int rc = 0;
    for (int64_t i = 0; i < k && !rc; i += nb) {
        const int64_t ib = (k - i < nb) ? (k - i) : nb;
        const int64_t mr = m - i - ib;                      // rows below the block's triangle
        const T* Vi = V + i + i * ldv;
        T* Ci = C + i;
        hipLaunchKernelGGL(unit_lower_copy_kernel<T>, dim3((unsigned)((ib * ib + 255) / 256)), dim3(256), 0, c->stream, ib, Vi, ldv,
                           V1, 0);
        RLHIP_LAUNCH_CHECK();
        // W = V1^T C1 + V2^T C2
        rc = gemm<T>(c, 1, 0, ib, n, ib, T(1), V1, ib, Ci, ldc, T(0), W, ib);
        if (!rc && mr > 0) rc = gemm<T>(c, 1, 0, ib, n, mr, T(1), Vi + ib, ldv, Ci + ib, ldc, T(1), W, ib);
        // W2 = T_i^T W   (T_i upper triangular, strictly lower part is zero)
        if (!rc) rc = gemm<T>(c, 1, 0, ib, n, ib, T(1), Tm + i * ldt, ldt, W, ib, T(0), W2, ib);
        // C1 -= V1 W2 ; C2 -= V2 W2
        if (!rc) rc = gemm<T>(c, 0, 0, ib, n, ib, T(-1), V1, ib, W2, ib, T(1), Ci, ldc);
        if (!rc && mr > 0) rc = gemm<T>(c, 0, 0, mr, n, ib, T(-1), Vi + ib, ldv, W2, ib, T(1), Ci + ib, ldc);
    }
    rlhip_ws_release(c, mark);
    return rc;
}

// The same apply for ONE compact-WY block (k reflectors), cut where the first k rows of C are final:
//   head:  W2 = T^T (V1^T C1 + V2^T C2),  C1 -= V1 W2     -- after it the block row C1 (BQRRP's R12, rl_bqrrp.hh:547) does not change any more
//   tail:  C2 -= V2 W2                                     -- the big product; nothing on the path to the NEXT panel's pivots reads C2
// BQRRP's look-ahead (rl_bqrrp.hh, detail::bqrrp_factor) runs the sketch down-date and the next QRCP of the sketch beside the tail.
// W2 is the caller's k x n buffer (ld k).  Same kernels, same order of operations as gemqrt_lt with nb >= k: bitwise the same C.
template <typename T>
int gemqrt_lt_head(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, const T* V, int64_t ldv, const T* Tm, int64_t ldt, T* C, int64_t ldc, T* W2) {
    if (m < 0) return -3;
    if (n < 0) return -4;
    if (k < 0 || k > m) return -5;
    if (k == 0 || n == 0 || m == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* V1 = ws_alloc<T>(c, (size_t)k * k);
    T* W = ws_alloc<T>(c, (size_t)k * n);
    if (!V1 || !W) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    const int64_t mr = m - k;
    hipLaunchKernelGGL(unit_lower_copy_kernel<T>, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, c->stream, k, V, ldv, V1, 0);
    int rc = 0;
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) rc = RLHIP_ERR_HIP(le);
    }
    if (!rc) rc = gemm<T>(c, 1, 0, k, n, k, T(1), V1, k, C, ldc, T(0), W, k);
    if (!rc && mr > 0) rc = gemm<T>(c, 1, 0, k, n, mr, T(1), V + k, ldv, C + k, ldc, T(1), W, k);
    if (!rc) rc = gemm<T>(c, 1, 0, k, n, k, T(1), Tm, ldt, W, k, T(0), W2, k);
    if (!rc) rc = gemm<T>(c, 0, 0, k, n, k, T(-1), V1, k, W2, k, T(1), C, ldc);
    rlhip_ws_release(c, mark);
    return rc;
}
template <typename T>
int gemqrt_lt_tail(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, const T* V, int64_t ldv, const T* W2, T* C, int64_t ldc) {
    if (m < 0) return -3;
    if (n < 0) return -4;
    if (k < 0 || k > m) return -5;
    const int64_t mr = m - k;
    if (k == 0 || n == 0 || mr <= 0) return 0;
    // the tiled kernel: its workgroups retire continuously, so the look-ahead's kernels on the side stream find CUs (the persistent kernel
    // would hold all of them until the product is done); same rate for this NN shape (132 TFLOP/s fp32 either way)
    const int keep = c->avoid_persistent;
    c->avoid_persistent = 1;
    const int rc = gemm<T>(c, 0, 0, mr, n, k, T(-1), V + k, ldv, W2, k, T(1), C + k, ldc);
    c->avoid_persistent = keep;
    return rc;
}

// C (m x n) <- C Q,  Q = I - V T V^T one compact-WY block (V: n x k unit lower trapezoidal, T: k x k upper).  Side::Right,
// Op::NoTrans: HQRRP's update of the sketching matrix G (NoFLA_Apply_Q_WY_rnfc_blk_var4, rl_hqrrp.hh:178-206).
template <typename T>
int gemqrt_rn(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, const T* V, int64_t ldv, const T* Tm, int64_t ldt, T* C, int64_t ldc) {
    if (m < 0) return -3;
    if (n < 0) return -4;
    if (k < 0 || k > n) return -5;
    if (k == 0 || n == 0 || m == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* V1 = ws_alloc<T>(c, (size_t)k * k);
    T* W = ws_alloc<T>(c, (size_t)m * k);
    T* W2 = ws_alloc<T>(c, (size_t)m * k);
    if (!V1 || !W || !W2) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    hipLaunchKernelGGL(unit_lower_copy_kernel<T>, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, c->stream, k, V, ldv, V1, 0);
    RLHIP_LAUNCH_CHECK();
    const int64_t nr = n - k;
    int rc = gemm<T>(c, 0, 0, m, k, k, T(1), C, ldc, V1, k, T(0), W, m);                                   // W = C1 V1 + C2 V2
    if (!rc && nr > 0) rc = gemm<T>(c, 0, 0, m, k, nr, T(1), C + k * ldc, ldc, V + k, ldv, T(1), W, m);
    if (!rc) rc = gemm<T>(c, 0, 0, m, k, k, T(1), W, m, Tm, ldt, T(0), W2, m);                              // W2 = W T
    if (!rc) rc = gemm<T>(c, 0, 1, m, k, k, T(-1), W2, m, V1, k, T(1), C, ldc);                             // C1 -= W2 V1^T
    if (!rc && nr > 0) rc = gemm<T>(c, 0, 1, m, nr, k, T(-1), W2, m, V + k, ldv, T(1), C + k * ldc, ldc);   // C2 -= W2 V2^T
    rlhip_ws_release(c, mark);
    return rc;
}
template int gemqrt_rn<double>(rlhip_ctx*, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*, int64_t);
template int gemqrt_rn<float>(rlhip_ctx*, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*, int64_t);

// T (k x k upper, ldt) from V (m x k unit lower trapezoidal) and tau (k): one compact-WY block for all k reflectors
template <typename T>
int larft_gram(rlhip_ctx* c, int64_t m, int64_t k, const T* V, int64_t ldv, const T* tau, T* Tm, int64_t ldt) {
    if (k <= 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* V1 = ws_alloc<T>(c, (size_t)k * k);
    T* G = ws_alloc<T>(c, (size_t)k * k);
    if (!V1 || !G) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    hipLaunchKernelGGL(unit_lower_copy_kernel<T>, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, c->stream, k, V, ldv, V1, 0);
    RLHIP_LAUNCH_CHECK();
    int rc = gemm<T>(c, 1, 0, k, k, k, T(1), V1, k, V1, k, T(0), G, k);
    if (!rc && m > k) rc = gemm<T>(c, 1, 0, k, k, m - k, T(1), V + k, ldv, V + k, ldv, T(1), G, k);
    if (rc) { rlhip_ws_release(c, mark); return rc; }
    hipLaunchKernelGGL(larft_m_kernel<T>, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, c->stream, k, G, k, tau);
    RLHIP_LAUNCH_CHECK();
    rc = laset<T>(c, 2, k, k, T(0), T(1), Tm, ldt);                       // T = I * M^-1
    if (!rc) rc = trsm_right_upper<T>(c, 0, k, k, T(1), G, k, Tm, ldt);
    rlhip_ws_release(c, mark);
    return rc;
}

template <typename T>
int row_sign(rlhip_ctx* c, int64_t n, T* R, int64_t ldr, const T* D) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(row_sign_kernel<T>, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, c->stream, n, R, ldr, D);
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template <typename T>
int tau_from_t(rlhip_ctx* c, int64_t k, int64_t nb, const T* Tm, int64_t ldt, T* tau) {
    if (k <= 0) return 0;
    hipLaunchKernelGGL(tau_from_t_kernel<T>, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, k, nb, Tm, ldt, tau);
    RLHIP_LAUNCH_CHECK();
    return 0;
}
// returns 1 in *any_host if some |x[i]| > thr
template <typename T>
int any_abs_gt(rlhip_ctx* c, int64_t n, const T* x, T thr, int* any_host) {
    *any_host = 0;
    if (n <= 0) return 0;
    int* d_flag = (int*)(c->d_mail + 48);
    hipLaunchKernelGGL(zero_int2, dim3(1), dim3(1), 0, c->stream, d_flag);
    hipLaunchKernelGGL(any_abs_gt_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, x, thr, d_flag);
    RLHIP_LAUNCH_CHECK();
    RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 48, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(rlhip_stream_sync(c));
    *any_host = *(int*)(c->h_mail + 48);
    return 0;
}

#define INST(T)                                                                                                          \
    template int orhr_col<T>(rlhip_ctx*, int64_t, int64_t, int64_t, T*, int64_t, T*, int64_t, T*);                       \
    template int gemqrt_lt<T>(rlhip_ctx*, int64_t, int64_t, int64_t, int64_t, const T*, int64_t, const T*, int64_t, T*, int64_t); \
    template int gemqrt_lt_head<T>(rlhip_ctx*, int64_t, int64_t, int64_t, const T*, int64_t, const T*, int64_t, T*, int64_t, T*); \
    template int gemqrt_lt_tail<T>(rlhip_ctx*, int64_t, int64_t, int64_t, const T*, int64_t, const T*, T*, int64_t); \
    template int larft_gram<T>(rlhip_ctx*, int64_t, int64_t, const T*, int64_t, const T*, T*, int64_t);                  \
    template int row_sign<T>(rlhip_ctx*, int64_t, T*, int64_t, const T*);                                                \
    template int tau_from_t<T>(rlhip_ctx*, int64_t, int64_t, const T*, int64_t, T*);                                     \
    template int any_abs_gt<T>(rlhip_ctx*, int64_t, const T*, T, int*);
INST(double)
INST(float)


// geqrf of a TALL-SKINNY matrix through the BLAS-3 route: Cholesky-QR twice (Q orthonormal to rounding when cond(A) <~ 1e8),
// then Householder reconstruction (orhr_col) turns Q into the unit-lower V and tau, and R = D R2 R1 goes above the diagonal --
// the geqrf output format, the same reflectors Householder QR would produce (they are unique for a given sign convention), at
// GEMM speed instead of one grid-wide step per column.  *done = 0 (A restored to its input up to rounding) when a Cholesky breaks
// down or the second R factor is not close to the identity, i.e. when CholQR cannot be trusted: the caller then runs the
// Householder pipeline.  rl_orth.hh:157 (HQRQ), rl_abrik.hh:333,420,552 (ABRIK panels), rl_hqrrp.hh GEQRF_mod_WY.
template <typename T>
__global__ void r2_identity_dev_kernel(int n, const T* __restrict__ R, int64_t ldr, T* __restrict__ out) {
    // out[0] = max over the upper triangle of |R - I|  (one workgroup)
    __shared__ T red[256];
    T v = 0;
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e % n, j = e / n;
        if (i <= j) { T d = fabs(R[i + (int64_t)j * ldr] - (i == j ? T(1) : T(0))); v = d > v ? d : v; }
    }
    red[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + st] ? red[threadIdx.x] : red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// Cholesky-QR twice, in place: on success A = Q (orthonormal columns) and R2 = the upper-triangular R with A_in = Q R2; *good = false
// (A restored to A_in up to rounding, or untouched) when a Cholesky factorization breaks down or the second R factor is not close to the
// identity -- the first Q was then too far from orthonormal for the second pass to repair it (cond(A) beyond ~1e7 in fp64).
// ---- skinny panels (n <= 64: ABRIK's Krylov blocks): X <- X R^-1 with a thread per row and R in LDS, and the whole Cholesky-QR-twice
// sequence behind ONE host read.  The generic route reads LAPACK's info after each Cholesky factorization and the identity defect after the
// second (three stream drains of ~25 us around ~250 us of kernels); here the factorizations are enqueued (potrf_upper_enqueue: info on the
// device), the first solve tests that word itself and leaves X alone after a breakdown, and the three verdicts come back together.
constexpr int SKN = 64;
// n = N exactly (16 / 32 / 64: the block sizes ABRIK is run with): every register index is a compile-time constant and nothing is predicated.
// (A run-time bound n <= N inside the unrolled loops -- predicated loads and stores -- sent the row to scratch: 149 us instead of ~25 at
// 200000 x 32; other widths take the generic route.)
template <typename T, int N>
__global__ __launch_bounds__(256) void skinny_trsm_kernel(int64_t m, const T* __restrict__ R, int64_t ldr, T* __restrict__ X, int64_t ldx,
                                                          const int* __restrict__ skip) {
    __shared__ T sR[N * N];                         // sR[j * N + i] = R(j, i) for j < i (row j of R, contiguous); the diagonal holds 1 / R(j, j)
    if (skip && *skip != 0) return;
    for (int e = threadIdx.x; e < N * N; e += 256) {
        const int j = e / N, i = e % N;             // entry (j, i); below the diagonal: never read
        T v = R[j + (int64_t)i * ldr];
        if (j == i) v = T(1) / v;
        sR[e] = v;
    }
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    T x[N];
    T* p = X + r;
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = p[(int64_t)j * ldx];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        x[j] *= sR[j * N + j];
#pragma unroll
        for (int i = j + 1; i < N; ++i) x[i] -= x[j] * sR[j * N + i];
    }
#pragma unroll
    for (int j = 0; j < N; ++j) p[(int64_t)j * ldx] = x[j];
}
template <typename T>
static int skinny_trsm(rlhip_ctx* c, int64_t m, int64_t n, const T* R, int64_t ldr, T* X, int64_t ldx, const int* skip) {
    const unsigned grid = (unsigned)((m + 255) / 256);
    if (n == 16) hipLaunchKernelGGL((skinny_trsm_kernel<T, 16>), dim3(grid), dim3(256), 0, c->stream, m, R, ldr, X, ldx, skip);
    else if (n == 32) hipLaunchKernelGGL((skinny_trsm_kernel<T, 32>), dim3(grid), dim3(256), 0, c->stream, m, R, ldr, X, ldx, skip);
    else if (n == 64) hipLaunchKernelGGL((skinny_trsm_kernel<T, 64>), dim3(grid), dim3(256), 0, c->stream, m, R, ldr, X, ldx, skip);
    else return -3;
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template <typename T> int potrf_upper_enqueue(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int* info_dev);

template <typename T>
static int cholqr2_skinny(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* R1, T* R2, T* dev1, bool* good) {
    *good = false;
    int* flags = (int*)ws_alloc<int64_t>(c, 2);      // [0] info of the first factorization, [1] of the second
    if (!flags) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    RLHIP_CHECK(hipMemsetAsync(flags, 0, 2 * sizeof(int64_t), c->stream));
    RLHIP_CHECK(hipMemsetAsync(dev1, 0, sizeof(T), c->stream));
    int rc = laset<T>(c, 2, n, n, T(0), T(0), R1, n);
    if (!rc) rc = syrk<T>(c, Upper, 1, n, m, T(1), A, lda, T(0), R1, n);
    if (!rc) rc = potrf_upper_enqueue<T>(c, n, R1, n, flags);
    if (rc) return rc < 0 ? rc : RLHIP_ERR_HIP(hipErrorUnknown);
    rc = skinny_trsm<T>(c, m, n, R1, n, A, lda, flags);
    if (rc) return rc;
    rc = laset<T>(c, 2, n, n, T(0), T(0), R2, n);
    if (!rc) rc = syrk<T>(c, Upper, 1, n, m, T(1), A, lda, T(0), R2, n);
    if (!rc) rc = potrf_upper_enqueue<T>(c, n, R2, n, flags + 2);
    if (rc) return rc < 0 ? rc : RLHIP_ERR_HIP(hipErrorUnknown);
    hipLaunchKernelGGL(r2_identity_dev_kernel<T>, dim3(1), dim3(256), 0, c->stream, (int)n, R2, (int64_t)n, dev1);
    RLHIP_LAUNCH_CHECK();
    RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 44, flags, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 46, dev1, sizeof(T), hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(rlhip_stream_sync(c));
    const int info1 = *(const int*)(c->h_mail + 44), info2 = *(const int*)(c->h_mail + 45);
    const T dev_h = *(const T*)(c->h_mail + 46);
    if (info1) return 0;                                                             // A untouched (the solve saw the flag)
    if (info2 || !(dev_h <= T(1e-2))) return trmm_right_upper<T>(c, NonUnit, m, n, T(1), R1, n, A, lda);   // restore A = Q1 R1
    rc = skinny_trsm<T>(c, m, n, R2, n, A, lda, nullptr);                            // A = Q
    if (rc) return rc;
    rc = trmm_right_upper<T>(c, NonUnit, n, n, T(1), R1, n, R2, n);                   // R2 <- R2 R1
    if (!rc) *good = true;
    return rc;
}

template <typename T>
static int cholqr2_inplace(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* R1, T* R2, T* dev1, bool* good) {
    *good = false;
    if ((n == 16 || n == 32 || n == SKN) && m >= 4096) return cholqr2_skinny<T>(c, m, n, A, lda, R1, R2, dev1, good);
    int info = 0;
    int rc = laset<T>(c, 2, n, n, T(0), T(0), R1, n);
    if (!rc) rc = syrk<T>(c, Upper, 1, n, m, T(1), A, lda, T(0), R1, n);
    if (!rc) rc = potrf_upper<T>(c, n, R1, n, &info);
    if (rc || info) return rc;                                                       // A untouched so far
    rc = trsm_right_upper<T>(c, NonUnit, m, n, T(1), R1, n, A, lda);
    if (!rc) rc = laset<T>(c, 2, n, n, T(0), T(0), R2, n);
    if (!rc) rc = syrk<T>(c, Upper, 1, n, m, T(1), A, lda, T(0), R2, n);
    if (!rc) rc = potrf_upper<T>(c, n, R2, n, &info);
    bool ok = !rc && !info;
    if (ok) {
        hipLaunchKernelGGL(r2_identity_dev_kernel<T>, dim3(1), dim3(256), 0, c->stream, (int)n, R2, (int64_t)n, dev1);
        T dev_h = 0;
        RLHIP_CHECK(hipMemcpyAsync(&dev_h, dev1, sizeof(T), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        ok = (dev_h <= T(1e-2));                                                     // Q1 was orthonormal to ~1e-2: pass 2 is accurate
    }
    if (!ok) {                                                                       // restore A = Q1 R1
        int rc2 = trmm_right_upper<T>(c, NonUnit, m, n, T(1), R1, n, A, lda);
        return rc ? rc : rc2;
    }
    rc = trsm_right_upper<T>(c, NonUnit, m, n, T(1), R2, n, A, lda);                 // A = Q
    if (!rc) rc = trmm_right_upper<T>(c, NonUnit, n, n, T(1), R1, n, R2, n);         // R2 <- R2 R1 (upper; lower part is zero)
    if (!rc) *good = true;
    return rc;
}

struct SasoOp;
int saso_build(rlhip_ctx* c, int64_t d, int64_t m, int nnz, int mode, const uint32_t ctr[4], const uint32_t key[2], uint32_t next_ctr[4], SasoOp** out);
int saso_destroy(rlhip_ctx* c, SasoOp* op);
template <typename T> int saso_apply(rlhip_ctx* c, const SasoOp* op, int64_t n, T alpha, const T* A, int64_t lda, T beta, T* B, int64_t ldb);
template <typename T> int geqrf(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau_dev);

// A (m x n, tall) <- its orthonormal factor by Cholesky-QR twice, R2 (n x n, ld n) <- the upper-triangular R with A_in = Q R2; for an
// ill-conditioned panel once more behind a sparse-sketch preconditioner (below).  *good = false: A holds its input (up to rounding, or bit
// for bit after the preconditioned attempt) and the caller takes the Householder route.  The caller owns the arena mark.
template <typename T>
static int cholqr_orthonormal(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* R2, bool* good_out) {
    *good_out = false;
    T* R1 = ws_alloc<T>(c, (size_t)n * n);
    T* dev1 = ws_alloc<T>(c, 4);
    if (!R1 || !dev1) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    bool good = false;
    int rc = cholqr2_inplace<T>(c, m, n, A, lda, R1, R2, dev1, &good);
    if (rc) return rc;
    // An ill-conditioned tall panel (ABRIK's Krylov blocks on a quickly decaying operator, rl_abrik.hh:333: cond 1e10 and beyond) is
    // where Cholesky-QR gives up and the column-by-column Householder kernel took over -- 10 ms for a 200000 x 32 panel.  Before that,
    // one more BLAS-3 attempt in the manner of CQRRT (rl_cqrrt.hh:124-200): a sparse sketch S A (2n rows), its small QR, and
    // A R_sk^-1 has condition O(1) whatever A's was (short of numerical rank deficiency), so Cholesky-QR twice goes through and
    // R = R_chol R_sk.  The sketch uses a fixed counter: the factorization stays a deterministic function of A.
    if (!good && m >= 8 * n && n >= 2) {
        const int64_t d = 2 * n > n + 32 ? 2 * n : n + 32;
        T* Acopy = ws_alloc<T>(c, (size_t)m * n);
        T* Ask = ws_alloc<T>(c, (size_t)d * n);
        T* R0 = ws_alloc<T>(c, (size_t)n * n);
        T* tsk = ws_alloc<T>(c, (size_t)n);
        if (Acopy && Ask && R0 && tsk) {
            rc = lacpy<T>(c, 2, m, n, A, lda, Acopy, m);
            SasoOp* S = nullptr;
            const uint32_t ctr[4] = {0x9e3779b9u, 0, 0, 0}, key[2] = {0x51ab17u, 0x2f};
            uint32_t nxt[4];
            if (!rc) rc = saso_build(c, d, m, (int)(d < 8 ? d : 8), 1, ctr, key, nxt, &S);
            if (!rc) rc = saso_apply<T>(c, S, n, T(1), A, lda, T(0), Ask, d);
            if (S) saso_destroy(c, S);
            if (!rc) rc = geqrf<T>(c, d, n, Ask, d, tsk);
            std::vector<T> dg((size_t)n);
            if (!rc) {
                RLHIP_CHECK(hipMemcpy2DAsync(dg.data(), sizeof(T), Ask, (size_t)(d + 1) * sizeof(T), sizeof(T), (size_t)n, hipMemcpyDeviceToHost, c->stream));
                RLHIP_CHECK(rlhip_stream_sync(c));
            }
            bool usable = !rc;
            if (usable) {
                T dmax = 0, dmin = std::numeric_limits<T>::max();
                for (int64_t i = 0; i < n; ++i) { const T a = std::abs(dg[i]); dmax = a > dmax ? a : dmax; dmin = a < dmin ? a : dmin; }
                usable = dmax > T(0) && dmin > T(64) * std::numeric_limits<T>::epsilon() * dmax;      // numerically rank deficient: leave it to Householder
            }
            if (usable) {
                rc = laset<T>(c, 2, n, n, T(0), T(0), R0, n);
                if (!rc) rc = lacpy<T>(c, 0, n, n, Ask, d, R0, n);
                if (!rc) rc = trsm_right_upper<T>(c, NonUnit, m, n, T(1), R0, n, A, lda);        // A <- A R_sk^-1
                if (!rc) rc = cholqr2_inplace<T>(c, m, n, A, lda, R1, R2, dev1, &good);
                if (!rc && good) rc = trmm_right_upper<T>(c, NonUnit, n, n, T(1), R0, n, R2, n); // R <- R_chol R_sk
                if (rc || !good) { int rc2 = lacpy<T>(c, 2, m, n, Acopy, m, A, lda); if (!rc) rc = rc2; good = false; }   // the input, bit for bit
                else c->path_count[5]++;
            }
        }
        if (rc) return rc;
    }
    *good_out = good;
    return 0;
}

template <typename T>
int geqrf_cholqr(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau, int* done) {
    *done = 0;
    size_t mark = rlhip_ws_mark(c);
    T* R2 = ws_alloc<T>(c, (size_t)n * n);
    T* Tm = ws_alloc<T>(c, (size_t)n * n);
    T* D = ws_alloc<T>(c, (size_t)n);
    if (!R2 || !Tm || !D) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    bool good = false;
    int rc = cholqr_orthonormal<T>(c, m, n, A, lda, R2, &good);
    if (rc) { rlhip_ws_release(c, mark); return rc; }
    if (!good) { rlhip_ws_release(c, mark); return 0; }                              // let Householder do it
    rc = orhr_col<T>(c, m, n, n, A, lda, Tm, n, D);                                  // V below the diagonal, T, sign vector D
    if (!rc) rc = row_sign<T>(c, n, R2, n, D);                                      // R <- D R
    if (!rc) rc = tau_from_t<T>(c, n, n, Tm, n, tau);
    if (!rc) rc = lacpy<T>(c, 0, n, n, R2, n, A, lda);                              // upper triangle incl. diagonal
    rlhip_ws_release(c, mark);
    if (!rc) *done = 1;
    return rc;
}
template int geqrf_cholqr<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*, int*);
template int geqrf_cholqr<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*, int*);

// geqrf followed by ungqr(m, n, n) in ONE pass for a tall panel (ABRIK's Krylov blocks, rl_abrik.hh:333-342 / :420-444 / :552-570; HQRQ,
// rl_orth.hh:157-162): A <- the first n columns of the Householder Q, R (n x n, ld ldr) <- the triangle geqrf would have left (zero below).
// The BLAS-3 geqrf above goes orthonormal factor -> reflectors (orhr_col over all m rows, T factor) and ungqr then goes reflectors ->
// orthonormal factor again: two passes over the panel and ~25 launches that cancel.  LAPACK's reconstruction contract is
// Q_cholesky = Q_householder diag(D), R_householder = diag(D) R_cholesky with D = +-1 decided by the sign-modified LU of the TOP n x n block alone
// (orhr_col's D depends on no other row), so: Cholesky-QR twice, D from a copy of the top block, one column scaling.  Same Q and R as the two
// calls to rounding.  *done = 0: not taken (A as geqrf_cholqr leaves it), the caller runs geqrf + ungqr.
template <typename T>
__global__ void scale_cols_sign_kernel(int64_t m, int64_t n, T* __restrict__ A, int64_t lda, const T* __restrict__ D) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * n) return;
    const int64_t i = idx % m, j = idx / m;
    if (D[j] < T(0)) A[i + j * lda] = -A[i + j * lda];
}
template <typename T>
int geqrf_q(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr, int* done) {
    *done = 0;
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (lda < (m > 1 ? m : 1)) return -5;
    if (ldr < (n > 1 ? n : 1)) return -7;
    if (n == 0 || m == 0) { *done = 1; return 0; }
    if (!(m >= 2 * n && n >= 8 && (size_t)m * n >= 16384)) return 0;                 // (the shapes the BLAS-3 geqrf serves)
    size_t mark = rlhip_ws_mark(c);
    T* R2 = ws_alloc<T>(c, (size_t)n * n);
    T* Qt = ws_alloc<T>(c, (size_t)n * n);
    T* D = ws_alloc<T>(c, (size_t)n);
    if (!R2 || !Qt || !D) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    bool good = false;
    int rc = cholqr_orthonormal<T>(c, m, n, A, lda, R2, &good);
    if (rc || !good) { rlhip_ws_release(c, mark); return rc; }
    rc = lacpy<T>(c, 2, n, n, A, lda, Qt, n);
    if (!rc) rc = lunp_top<T>(c, n, Qt, n, D);                                      // only D is wanted: the sign pattern of the top block's LU
    if (!rc) rc = row_sign<T>(c, n, R2, n, D);                                      // R <- D R
    if (!rc) {
        hipLaunchKernelGGL(scale_cols_sign_kernel<T>, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, c->stream, m, n, A, lda, (const T*)D);   // Q <- Q D
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) rc = RLHIP_ERR_HIP(le);
    }
    if (!rc) rc = laset<T>(c, 2, n, n, T(0), T(0), R, ldr);
    if (!rc) rc = lacpy<T>(c, 0, n, n, R2, n, R, ldr);
    rlhip_ws_release(c, mark);
    if (!rc) *done = 1;
    return rc;
}
template int geqrf_q<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*, int64_t, int*);
template int geqrf_q<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*, int64_t, int*);


// Rows [toff, toff + tcnt) of the unit-lower-triangular factor stored implicitly in Vtop (br x br: strictly lower part significant)
// written out explicitly (zeros above the diagonal, ones on it): the local rows of the reflector block a rank needs when the rows
// of a BQRRP panel are sharded across ranks.
template <typename T>
__global__ void vrows_explicit_kernel(int64_t br, int64_t toff, int64_t tcnt, const T* __restrict__ Vtop, int64_t ldv, T* __restrict__ out,
                                      int64_t ldo) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= tcnt * br) return;
    const int64_t i = idx % tcnt, j = idx / tcnt, t = toff + i;
    T v = 0;
    if (j < t) v = Vtop[t + j * ldv]; else if (j == t) v = 1;
    out[i + j * ldo] = v;
}
template <typename T>
int vrows_explicit(rlhip_ctx* c, int64_t br, int64_t toff, int64_t tcnt, const T* Vtop, int64_t ldv, T* out, int64_t ldo) {
    if (tcnt <= 0 || br <= 0) return 0;
    hipLaunchKernelGGL(vrows_explicit_kernel<T>, dim3((unsigned)((tcnt * br + 255) / 256)), dim3(256), 0, c->stream, br, toff, tcnt, Vtop, ldv, out, ldo);
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template int vrows_explicit<double>(rlhip_ctx*, int64_t, int64_t, int64_t, const double*, int64_t, double*, int64_t);
template int vrows_explicit<float>(rlhip_ctx*, int64_t, int64_t, int64_t, const float*, int64_t, float*, int64_t);

}  // namespace rlhip
