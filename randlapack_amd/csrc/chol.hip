// Device Cholesky factorisation A = U^T U (Uplo::Upper), replacing lapack::potrf on the CholQR steps
// (RandLAPACK/comps/rl_orth.hh:81, drivers/rl_cqrrpt.hh:311, drivers/rl_bqrrp.hh:462).
//
// n <= 448: ONE workgroup (potrf_small_kernel: 32-wide panels that never leave the CU, MFMA trailing update from LDS).
// Larger n: two-level blocking -- 256-wide steps of {one-workgroup diagonal block, block row through the blocked trsm of the transposed
// slab, K = 256 tri-tile GEMM update} (potrf_upper below).  (Round 1's 32-wide right-looking loop -- one panel launch + one K = 32
// update per 32 columns, 3.9 ms at n = 1024 against 1.31 -- was removed in round 3.)
// A non-positive (or NaN) pivot stores its 1-based global index in a device flag; later kernels see
// the flag and leave the matrix untouched, so the caller gets LAPACK's info and a partially factored
// matrix, as with dpotrf.
#include "rlhip_internal.h"
#include <cstdlib>

namespace {

constexpr int NB = 32;


__global__ void zero_int_kernel(int* p) { *p = 0; }

// ---- whole factorization in ONE workgroup for n <= 512 (the k x k Gram matrices of CholQR: k = 256 on the RSVD path).
// The blocked path above costs ~60 us per 32-column panel in launches and barriers (0.5 ms at n = 256, three times per
// RSVD step, replicated on every rank of a row-sharded run); here a panel is a few microseconds:
//   * the 32 x 32 diagonal block is factored by ONE wave without workgroup barriers: lane j owns column j in registers,
//     row k of U travels through LDS once per step;
//   * the block row U12 = U11^-T A12 is one column per thread (forward substitution from the LDS copy of U11);
//   * the trailing update A22 -= U12^T U12 reads U12 from LDS (<= 32 x 480 doubles) and touches each upper-triangle
//     entry of A22 once, 2 x 2 entries per thread per pass.
constexpr int PS_MAXN = 448;
constexpr int PS_LD = 34;          // column stride (doubles) of the LDS copy of U12: conflict-free MFMA fragment reads

// value of lane `src` (0..3) of the caller's quad, in all four lanes (DPP quad_perm broadcast)
template <typename T>
__device__ __forceinline__ T quad_bcast(T v, const int src) {
    if constexpr (sizeof(T) == 8) {
        const double d = (double)v;
        int lo = __double2loint(d), hi = __double2hiint(d), lo2, hi2;
        switch (src) {
            case 0: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x00, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x00, 0xF, 0xF, false); break;
            case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x55, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x55, 0xF, 0xF, false); break;
            case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0xAA, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0xAA, 0xF, 0xF, false); break;
            default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0xFF, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0xFF, 0xF, 0xF, false); break;
        }
        return (T)__hiloint2double(hi2, lo2);
    } else {
        int w = __float_as_int((float)v), w2;
        switch (src) {
            case 0: w2 = __builtin_amdgcn_update_dpp(0, w, 0x00, 0xF, 0xF, false); break;
            case 1: w2 = __builtin_amdgcn_update_dpp(0, w, 0x55, 0xF, 0xF, false); break;
            case 2: w2 = __builtin_amdgcn_update_dpp(0, w, 0xAA, 0xF, 0xF, false); break;
            default: w2 = __builtin_amdgcn_update_dpp(0, w, 0xFF, 0xF, 0xF, false); break;
        }
        return (T)__int_as_float(w2);
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void potrf_small_kernel(int n, T* __restrict__ A, int64_t lda, int* __restrict__ info, int info_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];
    T* sU11 = reinterpret_cast<T*>(ps_smem);          // [32][33]
    T* sRow = sU11 + 32 * 33;                          // [32] : row k of the diagonal block during its factorization
    T* sInv = sRow + 32;                               // [32] : 1 / u_kk
    T* sU12 = sInv + 32;                               // [rest16][PS_LD] : sU12[c*PS_LD + l] = U12[l, c], zero padded to 16 columns
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (info_base > 0 && *info != 0) return;           // a diagonal block of an earlier step of the two-level driver already failed
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += NB) {
        const int jb = (n - j0 < NB) ? (n - j0) : NB;
        const int rest = n - j0 - jb;
        // ---- 1. diagonal block: wave 0, lane j < 32 owns column j (rows 0..j matter).  1/sqrt from the hardware seed +
        //      Newton, sqrt = d * rsqrt(d) with one correction step; no IEEE divide/sqrt sequences on the serial path.
        if (wid == 0) {
            T col[NB];
            const int j = lane & 31;
#pragma unroll
            for (int i = 0; i < NB; ++i) {    // clamped addresses + select: 32 loads in flight instead of 32 branch-separated round trips
                const T t = A[(j0 + ((i < jb) ? i : (jb - 1))) + (int64_t)(j0 + ((j < jb) ? j : (jb - 1))) * lda];
                col[i] = (lane < 32 && i < jb && j < jb && i <= j) ? t : T(0);
            }
            int bad = 0;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (k < jb && !bad) {
                    const double ck = (double)col[k];
                    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ck), k),
                                                      __builtin_amdgcn_readlane(__double2loint(ck), k));   // pivot a_kk (lane k)
                    if (!(d > 0.0)) { bad = k + 1; }
                    else {
                        double y = __builtin_amdgcn_rsq(d);
                        y = y * fma(-0.5 * d * y, y, 1.5);
                        y = y * fma(-0.5 * d * y, y, 1.5);
                        double r = d * y;
                        r = fma(0.5 * y, fma(-r, r, d), r);                 // Heron correction: r = sqrt(d) to rounding
                        const T ukj = (j == k) ? (T)r : (T)((double)col[k] * y);   // row k of U (valid for j >= k)
                        col[k] = ukj;
                        if (lane < 32) sRow[j] = (j >= k) ? ukj : T(0);
                        if (lane == 0) sInv[k] = (T)y;
                        // (no wait on the write: the LDS executes one wavefront's requests in order, the reads below follow it)
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i = k + 1; i < NB; ++i) col[i] -= sRow[i] * ukj;   // a_ij -= u_ki u_kj  (i > k, column j)
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (bad && lane == 0) s_bad = j0 + bad;
            if (lane < 32) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    sU11[i * 33 + j] = (i <= j) ? col[i] : T(0);
                    // on a non-positive pivot (step bad - 1) the rows above it are final and go back, and the pivot entry holds its updated,
                    // non-positive value -- what dpotf2 leaves behind (Chol_check.cc reads the leading block of such a factor)
                    const bool keep = !bad || i < bad - 1 || (i == bad - 1 && j == bad - 1);
                    if (keep && i < jb && j < jb && i <= j) A[(j0 + i) + (int64_t)(j0 + j) * lda] = col[i];
                }
            }
        }
        __syncthreads();
        if (s_bad) break;
        if (rest <= 0) break;
        // ---- 2. block row: solve U11^T x = a for every column of A12.  FOUR lanes per column (lane g of the quad owns rows g, g + 4, ...):
        //      one thread per column left 4 of the 16 waves with ~1000 dependent LDS-read / FMA instructions each (100 of the kernel's
        //      250 us at n = 256); the quads put 14 waves to work with ~400 instructions each and x[l] travels by a DPP quad broadcast.
        //      (rest > 0 implies a full 32-column panel)
        for (int c = tid >> 2; c < rest; c += 256) {
            const int g = tid & 3;
            T x[NB / 4];
            T* colp = A + j0 + (int64_t)(j0 + jb + c) * lda;
#pragma unroll
            for (int r = 0; r < NB / 4; ++r) x[r] = colp[g + 4 * r];
#pragma unroll
            for (int l = 0; l < NB; ++l) {
                // the owner of row l (lane l % 4 of the quad) finishes x[l]; everybody in the quad gets it
                const T mine = x[l >> 2] * sInv[l];
                const T xl = quad_bcast(mine, l & 3);
                if (g == (l & 3)) x[l >> 2] = xl;
#pragma unroll
                for (int r = l >> 2; r < NB / 4; ++r) {
                    const int i = g + 4 * r;
                    if (i > l) x[r] -= sU11[l * 33 + i] * xl;
                }
            }
#pragma unroll
            for (int r = 0; r < NB / 4; ++r) {
                colp[g + 4 * r] = x[r];
                sU12[c * PS_LD + g + 4 * r] = x[r];
            }
        }
        for (int e = tid; e < (((rest + 15) / 16) * 16 - rest) * NB; e += 1024)      // zero the padding columns
            sU12[(rest + e / NB) * PS_LD + (e % NB)] = T(0);
        __syncthreads();
        // ---- 3. trailing update on the upper triangle of A22 on the matrix core: 16 x 16 tiles (I <= J), K = 32.
        //      The operands are swapped (columns of tile J as the MFMA A operand) so that a lane's results run along
        //      rows of A22 -> 128-byte segments.  sU12's column stride of 34 doubles makes the fragment reads conflict-free.
        {
            typedef double d4_t __attribute__((ext_vector_type(4)));
            const int nt = (rest + 15) / 16;
            const int ntile = nt * (nt + 1) / 2;
            const int fr = lane & 15, fk = lane >> 4;
            T* A22 = A + (j0 + jb) + (int64_t)(j0 + jb) * lda;
            for (int t = wid; t < ntile; t += 16) {
                int tj = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while ((tj + 1) * (tj + 2) / 2 <= t) ++tj;
                while (tj * (tj + 1) / 2 > t) --tj;
                const int ti = t - tj * (tj + 1) / 2;
                const int i0 = 16 * ti, c0 = 16 * tj;
                d4_t acc = {0, 0, 0, 0};
                const T* pa = sU12 + (c0 + fr) * PS_LD + fk;      // A operand: A[m = fr][k = fk] = U12[l][c0 + m]
                const T* pb = sU12 + (i0 + fr) * PS_LD + fk;      // B operand: B[k = fk][n = fr] = U12[l][i0 + n]
                // acc[r] = E[row = fk + 4r][col = fr] = D[i0 + fr][c0 + fk + 4r].  The four old values are requested first with clamped
                // addresses (no branch between the loads; their latency overlaps the eight MFMAs).
                const int gi = i0 + fr;
                const int gic = gi < rest ? gi : rest - 1;
                T oldv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gj = c0 + fk + 4 * r;
                    oldv[r] = A22[gic + (int64_t)(gj < rest ? gj : rest - 1) * lda];
                }
#pragma unroll
                for (int st = 0; st < NB / 4; ++st)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)pa[4 * st], (double)pb[4 * st], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gj = c0 + fk + 4 * r;
                    if (gi < rest && gj < rest && gi <= gj) A22[gi + (int64_t)gj * lda] = oldv[r] - (T)acc[r];
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) *info = s_bad ? info_base + s_bad : 0;
}


// Block row of the two-level factorization: U12 = U11^-T A12, in place.  Every column of A12 (jb x rest, jb <= 256) is an independent forward
// substitution with the lower triangular U11^T, so ONE launch with a wavefront per column does the whole block row: the column lives in
// the wave's registers (lane l holds rows l, l + 64, ...), step k divides by the pivot and subtracts x_k times column k of U11^T (= row k of
// U11, read from the transposed copy UT so that the lanes' loads are contiguous; the next column is requested before the division of the
// current step).  It replaces transpose -> blocked right-side solve (pack kernels, conditioning guard with its host read, one MFMA solve
// kernel) -> transpose: ~190 us of launches and a host round trip per 256-block for 2 * 256^2 * rest flops (C3's 1024 x 1024 Gram matrix:
// 1.38 ms for the whole factorization, 0.74 of it in the four diagonal blocks).  Plain substitution: no explicit inverse, no guard needed.
__device__ __forceinline__ double cr_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float cr_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <typename T>
__global__ __launch_bounds__(256) void chol_rowsolve_kernel(int jb, int64_t rest, const T* __restrict__ UT, T* __restrict__ A12, int64_t lda,
                                                            const int* __restrict__ info) {
    if (*info != 0) return;                                 // a diagonal block failed: nothing right of it is read by anybody
    constexpr int RPL = 4, PD = 4;                          // rows per lane; columns of U11^T in flight (a step is ~200 cycles of dependent
                                                            // arithmetic, a load from L2 three to four times that: one column ahead the chain waited for it, 68 us per launch)
    const int lane = threadIdx.x & 63;
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= rest) return;                                // (wave-uniform)
    T* b = A12 + col * lda;
    T x[RPL], u[PD][RPL];
    auto fetch = [&](int k, T (&dst)[RPL]) {                // column k of U11^T (clamped: past the end the last one again, unused)
        const int kc = (k < jb) ? k : jb - 1;
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const int i = lane + 64 * q;
            dst[q] = (i < jb) ? UT[i + (int64_t)kc * jb] : T(1);
        }
    };
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int i = lane + 64 * q;
        x[q] = (i < jb) ? b[i] : T(0);
    }
#pragma unroll
    for (int t = 0; t < PD; ++t) fetch(t, u[t]);
#pragma unroll
    for (int q0 = 0; q0 < RPL; ++q0) {
        for (int l4 = 0; l4 < 64; l4 += PD) {
            if (64 * q0 + l4 >= jb) break;
#pragma unroll
            for (int t = 0; t < PD; ++t) {
                const int l0 = l4 + t, k = 64 * q0 + l0;
                if (k < jb) {                               // (wave-uniform)
                    const T xk = cr_lane(x[q0], l0) / cr_lane(u[t][q0], l0);
                    if (lane == l0) x[q0] = xk;
#pragma unroll
                    for (int q = q0; q < RPL; ++q) {
                        const int i = lane + 64 * q;
                        if (i > k && i < jb) x[q] -= u[t][q] * xk;
                    }
                }
                fetch(k + PD, u[t]);                        // the slot is free: its column is requested PD steps ahead of its use
            }
        }
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int i = lane + 64 * q;
        if (i < jb) b[i] = x[q];
    }
}

}  // namespace

namespace rlhip {

template <typename T>
int gemm_impl(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
              int64_t lda, const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri, double* ssqA_dev = nullptr,
              int* ssq_done = nullptr);

template <typename T>
int trsm_right_upper(rlhip_ctx* c, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, T* B, int64_t ldb);

template <typename T>
int potrf_upper(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int* info_host) {
    *info_host = 0;
    if (n < 0) return -2;
    if (lda < (n > 1 ? n : 1)) return -4;
    if (n == 0) return 0;
    int* d_info = (int*)(c->d_mail + 8);
    if (n <= PS_MAXN) {
        const size_t smem = (size_t)(32 * 33 + 64 + (size_t)(n + 16) * PS_LD) * sizeof(T);
        RLHIP_FUNC_LDS(c, potrf_small_kernel<T>, 150 * 1024);
        hipLaunchKernelGGL(potrf_small_kernel<T>, dim3(1), dim3(1024), smem, c->stream, (int)n, A, lda, d_info, 0);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 8, d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        *info_host = *(int*)(c->h_mail + 8);
        return 0;
    }
    {
        // Two-level blocking for the n x n Gram matrices of CQRRPT / BQRRP's Cholesky-QR panels (n = 1024 .. 4096): 256-wide block steps,
        //   diagonal block      : the one-workgroup kernel above (its 32-wide panels never leave the CU),
        //   block row           : U12 = U11^-T A12, a wavefront per column of A12 in one launch (chol_rowsolve_kernel above; rounds 1-5a: the
        //                         RIGHT-side solve of the transposed slab on the blocked trsm of tri.hip, between two transposes),
        //   trailing update     : A22 -= U12^T U12 with K = 256 on the MFMA tri-tile GEMM.
        constexpr int64_t BS = 256;
        const size_t mark = rlhip_ws_mark(c);
        T* UT = ws_alloc<T>(c, (size_t)BS * BS);                 // U11^T of the current step (chol_rowsolve_kernel)
        if (!UT) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        RLHIP_FUNC_LDS(c, potrf_small_kernel<T>, 150 * 1024);
        hipLaunchKernelGGL(zero_int_kernel, dim3(1), dim3(1), 0, c->stream, d_info);
        int rc = 0;
        for (int64_t j0 = 0; j0 < n && !rc; j0 += BS) {
            const int64_t jb = (n - j0 < BS) ? (n - j0) : BS;
            const int64_t rest = n - j0 - jb;
            T* A11 = A + j0 + j0 * lda;
            const size_t smem = (size_t)(32 * 33 + 64 + (size_t)(jb + 16) * PS_LD) * sizeof(T);
            hipLaunchKernelGGL(potrf_small_kernel<T>, dim3(1), dim3(1024), smem, c->stream, (int)jb, A11, lda, d_info, (int)j0 + 1);
            RLHIP_LAUNCH_CHECK();
            if (rest <= 0) break;
            T* A12 = A + j0 + (j0 + jb) * lda;
            T* A22 = A + (j0 + jb) + (j0 + jb) * lda;
            rc = transpose<T>(c, jb, jb, A11, lda, UT, jb, 0);
            if (!rc) {
                hipLaunchKernelGGL(chol_rowsolve_kernel<T>, dim3((unsigned)((rest + 3) / 4)), dim3(256), 0, c->stream, (int)jb, rest, (const T*)UT, A12, lda,
                                   (const int*)d_info);
                const hipError_t le = hipGetLastError();
                if (le != hipSuccess) rc = RLHIP_ERR_HIP(le);        // (the arena mark is released below on every path)
            }
            if (!rc) rc = gemm_impl<T>(c, 1, 0, rest, rest, jb, T(-1), A12, lda, A12, lda, T(1), A22, lda, 1);
        }
        rlhip_ws_release(c, mark);
        if (rc) return rc;
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 8, d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        const int v = *(int*)(c->h_mail + 8);
        *info_host = v ? v - 1 : 0;
        return 0;
    }
}

// The one-workgroup factorization ENQUEUED only: LAPACK's info (0, or the 1-based index of the first non-positive pivot) goes to the DEVICE
// word `info_dev`, which later kernels of the stream may test; nothing is read back.  n <= 448 only (returns 1 otherwise: not available).
template <typename T>
int potrf_upper_enqueue(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int* info_dev) {
    if (n <= 0 || n > PS_MAXN || lda < n) return 1;
    const size_t smem = (size_t)(32 * 33 + 64 + (size_t)(n + 16) * PS_LD) * sizeof(T);
    RLHIP_FUNC_LDS(c, potrf_small_kernel<T>, 150 * 1024);
    hipLaunchKernelGGL(potrf_small_kernel<T>, dim3(1), dim3(1024), smem, c->stream, (int)n, A, lda, info_dev, 0);
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template int potrf_upper_enqueue<double>(rlhip_ctx*, int64_t, double*, int64_t, int*);
template int potrf_upper_enqueue<float>(rlhip_ctx*, int64_t, float*, int64_t, int*);

template int potrf_upper<double>(rlhip_ctx*, int64_t, double*, int64_t, int*);
template int potrf_upper<float>(rlhip_ctx*, int64_t, float*, int64_t, int*);

}  // namespace rlhip
