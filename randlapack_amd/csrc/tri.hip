// Triangular solve / multiply from the right with an upper-triangular factor:
//   trsm:  B <- alpha * B * inv(U)     (blas::trsm Side::Right, Uplo::Upper, NoTrans:
//          RandLAPACK/comps/rl_orth.hh:95, drivers/rl_cqrrpt.hh:302,338, drivers/rl_bqrrp.hh:457,464)
//   trmm:  B <- alpha * B * U          (blas::trmm, drivers/rl_cqrrpt.hh:345, drivers/rl_bqrrp.hh:497)
//
// trsm design.  Rows of B are independent in a right-side solve, so the matrix is cut into 256-row
// slabs, one row per lane, and solved by TRUE substitution (no explicit inverse of U: on CQRRPT's
// preconditioning step cond(U) can exceed 1e10 and an inverse-based solve would lose the
// eps*||A|| residual the reference's tests demand).  Column blocks of width DB=256 are chained
// left-to-right; the off-diagonal contribution of earlier blocks is one MFMA GEMM per block
// (B_J = alpha*B_J - X_{<J} U_{<J,J}), the diagonal block is a fused kernel:
//   * U_JJ is first packed row-major into scratch so that the values a lane needs next are wave-uniform
//     and contiguous -> the compiler turns them into s_load_dwordx16 + SGPR operands of v_fma_f64,
//   * 32-column sub-blocks live in registers (64 VGPRs); earlier x values of the same row are re-read
//     from L1/L2 (coalesced: lanes = consecutive rows).
// fp64 vector FMA and fp64 MFMA have the same peak on MI355X (78.6 TF), so the VALU diagonal kernel is
// not a bottleneck: at k=256 it is 2*m*k^2/2 flops against 2*m*n*k for the sketch GEMM.
#include "rlhip_internal.h"

namespace rlhip {
template <typename T>
int gemm_impl(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
              int64_t lda, const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri, double* ssqA_dev = nullptr,
              int* ssq_done = nullptr);
}

namespace {

constexpr int SB = 32;    // sub-block solved by the row-per-lane kernel (SB = 64 halves the passes but its LDS tile allows only
                          // two waves per CU: measured 714 us per pass against 123 us for SB = 32)
constexpr int RW = 256;   // rows (= threads) per workgroup of that kernel: LDS tile SB x RW + packed triangle SB x SB
constexpr int DB = 256;   // diagonal block handled by one fused kernel

// Ut[l * ldp + c] = U[l, c] for l <= c < nb (row-major, zero below the diagonal), padded with the identity
// up to ldp (a multiple of SB).  Diagonal entries hold the RECIPROCAL (LAPACK's dtrsm multiplies by
// 1/A(j,j) as well), or 1 for Diag::Unit.
template <typename T>
__global__ void pack_upper_rows_kernel(int nb, int ldp, int unit, const T* __restrict__ U, int64_t ldu,
                                       T* __restrict__ Ut) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ldp * ldp) return;
    int c = idx % ldp, l = idx / ldp;
    T v = 0;
    if (l < nb && c < nb) {
        if (l < c) v = U[l + (int64_t)c * ldu];
        else if (l == c) v = unit ? T(1) : T(1) / U[l + (int64_t)c * ldu];
    } else if (l == c) {
        v = 1;
    }
    Ut[(int64_t)l * ldp + c] = v;
}

// One row per lane.  The 32 values of a row are fetched with 32 back-to-back loads (clamped column index, no branches:
// a branch per column serialised the loads behind s_waitcnt vmcnt(0), and feeding the triangle through SGPR operands
// spilled thousands of SGPRs to VGPR lanes -- that version ran at 2 TB/s).  The solve then works on 8-column register
// groups: columns already solved live in an LDS tile (row r of the workgroup = lane r, conflict free), their
// contribution is a ROLLED loop (tiny code, broadcast 16-byte reads of the packed triangle), the 8 x 8 triangle of the
// group itself is unrolled.
template <typename T>
__global__ __launch_bounds__(RW) void trsm_diag_kernel(int64_t m, int nb, int ldp, T alpha,
                                                        const T* __restrict__ Ut, T* __restrict__ B,
                                                        int64_t ldb) {
    constexpr int GW = 8;                                  // register group width
    __shared__ __attribute__((aligned(16))) T sU[SB * SB]; // packed rows of the triangle, reciprocal diagonal
    __shared__ T sX[SB * RW];                              // sX[c * RW + lane]: solved columns of this workgroup's rows
    const int tid = threadIdx.x;
    for (int e = tid; e < SB * SB; e += RW) sU[e] = Ut[e];
    const int64_t r = (int64_t)blockIdx.x * RW + tid;
    const bool live = r < m;
    T* __restrict__ row = B + (live ? r : m - 1);          // dead lanes shadow the last row without storing
    {
        T v[SB];
#pragma unroll
        for (int c = 0; c < SB; ++c) {
            const int cc = (c < nb) ? c : (nb - 1);
            v[c] = row[(int64_t)cc * ldb];
        }
#pragma unroll
        for (int c = 0; c < SB; ++c) sX[c * RW + tid] = alpha * v[c];
    }
    __syncthreads();                                       // sU complete (sX is only touched by its own lane)
    for (int g0 = 0; g0 < SB; g0 += GW) {
        T x[GW];
#pragma unroll
        for (int j = 0; j < GW; ++j) x[j] = sX[(g0 + j) * RW + tid];
        for (int l = 0; l < g0; ++l) {                     // earlier columns of the block (rolled)
            const T xl = sX[l * RW + tid];
            const T* u = sU + l * SB + g0;
#pragma unroll
            for (int j = 0; j < GW; ++j) x[j] -= xl * u[j];
        }
#pragma unroll
        for (int j = 0; j < GW; ++j) {                     // the group's own 8 x 8 triangle
            const T* u = sU + (g0 + j) * SB + g0;
            x[j] *= u[j];
#pragma unroll
            for (int j2 = j + 1; j2 < GW; ++j2) x[j2] -= x[j] * u[j2];
        }
#pragma unroll
        for (int j = 0; j < GW; ++j) sX[(g0 + j) * RW + tid] = x[j];
    }
    if (live) {
        for (int c = 0; c < nb; ++c) row[(int64_t)c * ldb] = sX[c * RW + tid];
    }
    (void)ldp;
}

template <typename T>
__global__ void copy_triu_kernel(int64_t n, int unit, const T* __restrict__ U, int64_t ldu, T* __restrict__ W) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int64_t i = idx % n, j = idx / n;
    T v = 0;
    if (i < j) v = U[i + j * ldu];
    else if (i == j) v = unit ? T(1) : U[i + j * ldu];
    W[idx] = v;
}

}  // namespace

namespace rlhip {

template <typename T>
int lacpy(rlhip_ctx* c, int uplo, int64_t m, int64_t n, const T* A, int64_t lda, T* B, int64_t ldb);

template <typename T>
int trsm_right_upper(rlhip_ctx* c, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, T* B,
                     int64_t ldb) {
    if (m < 0) return -6;
    if (n < 0) return -7;
    if (lda < (n > 1 ? n : 1)) return -10;
    if (ldb < (m > 1 ? m : 1)) return -12;
    if (m == 0 || n == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* Ut = ws_alloc<T>(c, (size_t)SB * SB);
    if (!Ut) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    // Two-level blocking.  Outer 256-column blocks: the contribution of everything to the left is ONE wide MFMA
    // GEMM (N = 256 -> stream-K path at scale).  Inside a block, 32-column sub-blocks: a narrow MFMA GEMM (N = 32,
    // K <= 224) brings in the already solved columns of the block, then the row-per-lane substitution kernel
    // finishes the 32 x 32 triangle entirely in registers.  (A single VALU kernel for the whole 256-block ran at
    // ~8 % of the fp64 vector peak -- its inner loop re-reads x from L2 and U through the scalar cache.  Two fused
    // MFMA kernels were also tried and REJECTED at m = 1e6, k = 1024: 64-row slab in LDS + U from L2: 60 ms;
    // wave-owned rows with X fragments from L2 + staged U chunks: 38 ms; this two-level scheme: 36 ms (31 ms after the
    // substitution kernel was fixed).  A third fused design -- 32-row slab of a whole 256-block resident in LDS, two
    // workgroups per CU, right-looking with the eight tile accumulators in registers and block row s of U in flight during
    // the substitution of sub-block s -- was correct but ran 7.0 ms per 256-block against 3.85 ms here: with only one wave
    // of a workgroup substituting (32 rows), the 8 x ~1000 dependent LDS/FMA instructions per slab (2.2 ms per block)
    // and the U fragment loads (1.7 ms) cannot hide behind the 0.7 ms of MFMA work; the unfused kernels spread the same
    // substitution over every wave of every CU.)
    for (int64_t j0 = 0; j0 < n; j0 += DB) {
        const int nb = (int)((n - j0 < DB) ? (n - j0) : DB);
        T a = alpha;
        if (j0 > 0) {
            int rc = gemm_impl<T>(c, 0, 0, m, nb, j0, T(-1), B, ldb, A + j0 * lda, lda, alpha, B + j0 * ldb, ldb, 0);
            if (rc) { rlhip_ws_release(c, mark); return rc; }
            a = T(1);
        }
        for (int s0 = 0; s0 < nb; s0 += SB) {
            const int sbw = (nb - s0 < SB) ? (nb - s0) : SB;
            const int64_t jc = j0 + s0;
            T a2 = a;
            if (s0 > 0) {
                int rc = gemm_impl<T>(c, 0, 0, m, sbw, s0, T(-1), B + j0 * ldb, ldb, A + j0 + jc * lda, lda, a, B + jc * ldb,
                                      ldb, 0);
                if (rc) { rlhip_ws_release(c, mark); return rc; }
                a2 = T(1);
            }
            hipLaunchKernelGGL(pack_upper_rows_kernel<T>, dim3((SB * SB + 255) / 256), dim3(256), 0, c->stream, sbw, SB, diag,
                               A + jc + jc * lda, lda, Ut);
            hipLaunchKernelGGL(trsm_diag_kernel<T>, dim3((unsigned)((m + RW - 1) / RW)), dim3(RW), 0, c->stream, m, sbw, SB, a2,
                               Ut, B + jc * ldb, ldb);
            RLHIP_LAUNCH_CHECK();
        }
    }
    rlhip_ws_release(c, mark);
    return 0;
}

template <typename T>
int trmm_right_upper(rlhip_ctx* c, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, T* B,
                     int64_t ldb) {
    if (m < 0) return -6;
    if (n < 0) return -7;
    if (lda < (n > 1 ? n : 1)) return -10;
    if (ldb < (m > 1 ? m : 1)) return -12;
    if (m == 0 || n == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* W = ws_alloc<T>(c, (size_t)n * n);
    T* Bc = ws_alloc<T>(c, (size_t)m * n);
    if (!W || !Bc) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    hipLaunchKernelGGL(copy_triu_kernel<T>, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, c->stream, n, diag, A,
                       lda, W);
    RLHIP_LAUNCH_CHECK();
    int rc = lacpy<T>(c, 2, m, n, B, ldb, Bc, m);
    if (!rc) rc = gemm_impl<T>(c, 0, 0, m, n, n, alpha, Bc, m, W, n, T(0), B, ldb, 0);
    rlhip_ws_release(c, mark);
    return rc;
}

// B <- alpha * op(A) * B, A m x m upper triangular (Side::Left, Uplo::Upper), B m x n: the small n x n products of the
// linop QR drivers (rl_cqrrt_linops.hh:322, rl_scholqr3_linops.hh:352).  Same masked-copy + MFMA GEMM as the right-side form.
template <typename T>
int trmm_left_upper(rlhip_ctx* c, int trans, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, T* B, int64_t ldb) {
    if (m < 0) return -6;
    if (n < 0) return -7;
    if (lda < (m > 1 ? m : 1)) return -10;
    if (ldb < (m > 1 ? m : 1)) return -12;
    if (m == 0 || n == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* W = ws_alloc<T>(c, (size_t)m * m);
    T* Bc = ws_alloc<T>(c, (size_t)m * n);
    if (!W || !Bc) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    hipLaunchKernelGGL(copy_triu_kernel<T>, dim3((unsigned)((m * m + 255) / 256)), dim3(256), 0, c->stream, m, diag, A, lda, W);
    RLHIP_LAUNCH_CHECK();
    int rc = lacpy<T>(c, 2, m, n, B, ldb, Bc, m);
    if (!rc) rc = gemm_impl<T>(c, trans ? 1 : 0, 0, m, n, m, alpha, W, m, Bc, m, T(0), B, ldb, 0);
    rlhip_ws_release(c, mark);
    return rc;
}
template int trmm_left_upper<double>(rlhip_ctx*, int, int, int64_t, int64_t, double, const double*, int64_t, double*, int64_t);
template int trmm_left_upper<float>(rlhip_ctx*, int, int, int64_t, int64_t, float, const float*, int64_t, float*, int64_t);

template int trsm_right_upper<double>(rlhip_ctx*, int, int64_t, int64_t, double, const double*, int64_t, double*, int64_t);
template int trsm_right_upper<float>(rlhip_ctx*, int, int64_t, int64_t, float, const float*, int64_t, float*, int64_t);
template int trmm_right_upper<double>(rlhip_ctx*, int, int64_t, int64_t, double, const double*, int64_t, double*, int64_t);
template int trmm_right_upper<float>(rlhip_ctx*, int, int64_t, int64_t, float, const float*, int64_t, float*, int64_t);

}  // namespace rlhip
