// Triangular solve / multiply from the right with an upper-triangular factor:
//   trsm:  B <- alpha * B * inv(U)     (blas::trsm Side::Right, Uplo::Upper, NoTrans:
//          RandLAPACK/comps/rl_orth.hh:95, drivers/rl_cqrrpt.hh:302,338, drivers/rl_bqrrp.hh:457,464)
//   trmm:  B <- alpha * B * U          (blas::trmm, drivers/rl_cqrrpt.hh:345, drivers/rl_bqrrp.hh:497)
//
// trsm design.  Column blocks of width 256 are chained left to right; the contribution of earlier blocks is one MFMA GEMM per
// block (B_J = alpha * B_J - X_{<J} U_{<J,J}).  A diagonal block U_JJ is solved by one of two kernels, chosen PER BLOCK:
//   * blk path (trsm_blk_kernel): one launch, everything on the matrix cores, with the explicit inverses of the 32 x 32 diagonal
//     sub-blocks.  Taken only where those are well conditioned (kappa_F <= 1e3, measured by the pack kernel and read back once per
//     call), so that the inverse costs at most eps * 1e3.
//   * substitution path (trsm_diag_kernel + narrow GEMMs): TRUE substitution, one row per lane.  On CQRRPT's preconditioning step
//     the triangle is the R factor of an ill-conditioned sketch (graded rows, cond > 1e10) and an inverse-based solve would lose the
//     eps * ||A|| residual the reference's tests demand (tests/test_gpu_kernels.py::test_trsm_is_substitution_not_inverse).
//     U_JJ is packed row-major so that the values a lane needs next are wave-uniform and contiguous; 32-column sub-blocks live in
//     registers, earlier x values of the same row in an LDS tile.
#include <cstdlib>
#include "rlhip_internal.h"
#include "rlhip.h"
#include <cstdio>

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


// ---------------------------------------------------------------------------------------------------------------------
// One-launch solve of a whole 256-column diagonal block on the matrix cores ("blk" path).
//   pack:   Upk (256 x 256, ROW-major) = strictly upper part of the block, zero elsewhere / in the padding;
//           Dinv[s] (32 x 32, row-major) = inverse of the s-th 32 x 32 diagonal block (one thread per column, back substitution
//           in registers), identity in the padding.
//   solve:  a wave owns 16 rows of B.  Left-looking over the eight 32-column sub-blocks:
//               T   = a * B_s - sum_{c < 32 s} X[:, c] U[c, s-block]      (MFMA, X fragments straight from B's solved columns)
//               X_s = T * inv(U_ss)                                        (MFMA; for fp64 the accumulator layout of
//                     v_mfma_f64_16x16x4 (row = lane/16 + 4 r) already IS the operand layout of the next product, fp32 needs
//                     one cross-lane move per fragment)
//           and X_s goes back to B.  576 MFMAs per wave, every byte of B read and written once, U served from L2.
// Replaces, per 256-block, 8 substitution launches + 8 packs + 7 narrow GEMMs (570 us at m = 32768; 3.85 ms at m = 1e6).
// The diagonal blocks are applied through their explicit 32 x 32 inverses (as MAGMA / rocBLAS do): the error of a sub-block solve
// is eps * cond(U_ss) instead of the componentwise bound of substitution.
template <typename T> struct BlkMma;
template <> struct BlkMma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double x, double y, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }
    static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
    static constexpr int CS = 4, CL = 1;      // drow(lane, r) = CL * (lane >> 4) + CS * r
    // value T[m = lane & 15][col = c0 + (lane >> 4)] of a 16 x 16 accumulator tile (c0 multiple of 4): already in this lane
    static __device__ __forceinline__ double operand(const acc_t& t, int c0, int) { return t[c0 >> 2]; }
};
template <> struct BlkMma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float x, float y, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0); }
    static __device__ __forceinline__ int drow(int lane, int r) { return 4 * (lane >> 4) + r; }
    static constexpr int CS = 1, CL = 4;
    // fp32 accumulators hold columns 4 (lane/16) + r: column c0 + (lane >> 4) sits in lane group c0 / 4, register (lane >> 4)
    static __device__ __forceinline__ float operand(const acc_t& t, int c0, int lane) {
        const int src = (lane & 15) + 16 * (c0 >> 2);
        const float v0 = __shfl(t[0], src), v1 = __shfl(t[1], src), v2 = __shfl(t[2], src), v3 = __shfl(t[3], src);
        const int r = lane >> 4;
        return r == 0 ? v0 : r == 1 ? v1 : r == 2 ? v2 : v3;
    }
};

constexpr int BW = 256;   // block width of the blk path

// blockIdx.y = 256-column block of the n x n triangle A.  bad[blk] is raised when a diagonal 32 x 32 block is too ill conditioned
// (||U_ss||_F ||U_ss^-1||_F > limit, or not finite) for its explicit inverse to be used: that block then takes the substitution path.
template <typename T>
__global__ __launch_bounds__(256) void trsm_blk_pack_kernel(int64_t n, int unit, const T* __restrict__ A, int64_t ldu, T* __restrict__ Upk_all,
                                                            T* __restrict__ Dinv_all, int* __restrict__ bad, double limit2) {
    const int tid = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.y * BW;
    const int nb = (int)((n - j0 < BW) ? (n - j0) : BW);
    const T* U = A + j0 + j0 * ldu;
    T* Upk = Upk_all + (int64_t)blockIdx.y * BW * BW;
    T* Dinv = Dinv_all + (int64_t)blockIdx.y * (BW / 32) * 1024;
    if (blockIdx.x < BW / 32) {                         // inverse of diagonal block s
        __shared__ T sU[32][33];
        __shared__ double s_ni[32];
        const int s = blockIdx.x, o = 32 * s;
        for (int e = tid; e < 32 * 32; e += 256) {
            const int i = e & 31, j = e >> 5;
            T v = (i == j) ? T(1) : T(0);
            if (o + i < nb && o + j < nb && i <= j) v = (i == j && unit) ? T(1) : U[(o + i) + (int64_t)(o + j) * ldu];
            sU[i][j] = v;
        }
        __syncthreads();
        if (tid < 32) {
            const int j = tid;
            T x[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = T(0);
#pragma unroll
            for (int i = 31; i >= 0; --i) {             // back substitution for column j of the inverse (rows i <= j)
                if (i <= j) {
                    T acc = (i == j) ? T(1) : T(0);
#pragma unroll
                    for (int l = i + 1; l < 32; ++l)
                        if (l <= j) acc -= sU[i][l] * x[l];
                    x[i] = acc / sU[i][i];
                }
            }
            T* out = Dinv + (int64_t)s * 1024;
            double ni = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                out[i * 32 + j] = x[i];                                   // row-major [k = i][n = j]; zero below the diagonal
                ni += (double)x[i] * (double)x[i];                        // (entries below the diagonal are exact zeros)
            }
            s_ni[j] = ni;
        }
        __syncthreads();
        if (tid == 0) {                                   // kappa_F^2 of the block, summed serially (32 + 528 terms)
            double nu = 0, ni = 0;
            for (int jj = 0; jj < 32; ++jj) {
                ni += s_ni[jj];
                for (int i = 0; i <= jj; ++i) nu += (double)sU[i][jj] * (double)sU[i][jj];
            }
            if (!(nu * ni <= limit2)) atomicOr(&bad[blockIdx.y], 1);   // NaN / inf compare false -> flagged
        }
        return;
    }
    // row-major copy of the strictly-upper, off-diagonal-block part
    const int64_t e0 = (int64_t)(blockIdx.x - BW / 32) * 256 + tid;
    const int64_t stride = (int64_t)(gridDim.x - BW / 32) * 256;
    for (int64_t e = e0; e < (int64_t)BW * BW; e += stride) {
        const int k = (int)(e / BW), nn = (int)(e % BW);
        T v = T(0);
        if (k < nb && nn < nb && (k >> 5) < (nn >> 5)) v = U[k + (int64_t)nn * ldu];   // blocks strictly above the diagonal blocks
        Upk[e] = v;
    }
}

template <typename T, int RT>
__global__ __launch_bounds__(256) void trsm_blk_kernel(int64_t m, int nb, T alpha, const T* __restrict__ Upk, const T* __restrict__ Dinv,
                                                       T* __restrict__ B, int64_t ldb, int s_lo, int c_lo) {
    // solves the sub-blocks s_lo .. of the block's first nb columns; the columns < c_lo have been applied by the caller (one GEMM)
    using M = BlkMma<T>;
    using acc_t = typename M::acc_t;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * (16 * RT);      // a wave owns RT row tiles of 16 rows: every U fragment feeds RT MFMAs
    if (row0 >= m) return;
    T* Brow[RT];
    bool live[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const int64_t r = row0 + 16 * q + fr;
        live[q] = r < m;
        Brow[q] = B + (live[q] ? r : m - 1);                               // clamped: loads unconditional, stores masked
    }
    const int nsub = (nb + 31) >> 5;
    for (int s = s_lo; s < nsub; ++s) {
        acc_t acc[RT][2];
#pragma unroll
        for (int q = 0; q < RT; ++q) { acc[q][0] = acc_t{0, 0, 0, 0}; acc[q][1] = acc_t{0, 0, 0, 0}; }
        // ---- contribution of the columns already solved
        const T* up = Upk + 32 * s + fr;
        // one earlier sub-block (32 columns) at a time with a constant trip count: the eight steps' loads are issued together
        for (int c0 = c_lo; c0 < 32 * s; c0 += 32) {
            T x0[8], x1[8], y[8][RT];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + 4 * u;
                x0[u] = up[(int64_t)(c + fk) * BW]; x1[u] = up[(int64_t)(c + fk) * BW + 16];
#pragma unroll
                for (int q = 0; q < RT; ++q) y[u][q] = Brow[q][(int64_t)(c + fk) * ldb];     // X[row][c + fk]
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < RT; ++q) {
                    acc[q][0] = M::mma(x0[u], y[u][q], acc[q][0]);
                    acc[q][1] = M::mma(x1[u], y[u][q], acc[q][1]);
                }
        }
        const T* dv = Dinv + (int64_t)s * 1024 + fr;
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            // ---- T = alpha * B_s - acc   (lane holds row fr, columns 32 s + 16 u + drow(lane, r))
            acc_t t[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 32 * s + 16 * u + M::drow(lane, r);
                    const T b = Brow[q][(int64_t)(col < nb ? col : nb - 1) * ldb];
                    t[u][r] = (col < nb) ? alpha * b - acc[q][u][r] : T(0);
                }
            // ---- X_s = T * inv(U_ss)
            acc_t xs[2] = {acc_t{0, 0, 0, 0}, acc_t{0, 0, 0, 0}};
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                const T y = M::operand(t[c >> 4], c & 15, lane);                         // T[row][c + fk]
                const T d0 = dv[(c + fk) * 32], d1 = dv[(c + fk) * 32 + 16];
                if (c < 16) xs[0] = M::mma(d0, y, xs[0]);                                // inverse is upper triangular: rows >= 16 do not reach columns < 16
                xs[1] = M::mma(d1, y, xs[1]);
            }
            if (live[q]) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = 32 * s + 16 * u + M::drow(lane, r);
                        if (col < nb) Brow[q][(int64_t)col * ldb] = xs[u][r];
                    }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused left-looking solve of a RANGE of 256-column blocks in ONE launch ("fused" path; replaces, per call, the chain
// GEMM update -> trsm_blk_kernel -> GEMM update -> ... of the blk path).
//   * a wave owns 16 rows of B for the whole launch and keeps the 16 x 256 tile of the block being solved in its accumulator
//     registers (16 tiles of 16 x 16; for fp64 the v_mfma_f64_16x16x4 accumulator layout IS the B-operand layout of the next
//     product, so solved columns feed later products without leaving the register file);
//   * U is packed once per call as Uneg = -(strictly block-upper part of U), ROW-major, zero inside the 32 x 32 diagonal blocks
//     (those are applied through their explicit inverses Dinv, as in the blk path and under the same kappa_F <= 1e3 guard);
//   * the eight waves of a workgroup share U through LDS: stages of 32 rows x 256 columns (row stride: fused_stride<T>() below, chosen so that the lane groups
//     of an operand read land on disjoint banks) are DMA'd (global_load_lds) into two stages, ONE rendezvous per stage
//     (32 rows of U = 128 MFMAs per wave), the next stage in flight meanwhile (round 5; rounds 1-4: a ring of three 16-row panels,
//     a rendezvous per 16 rows, and 116 bytes of register spills per lane that the coarser loop no longer needs);
//   * block J: T = alpha B_J - sum_{I < J} X_I U_IJ with X_I read back from B (this wave wrote those rows itself), then
//     right-looking over the eight 32-column sub-blocks: X_s = T_s inv(U_ss); T_t -= X_s U_st for t > s -- all 14 - 2 s tile
//     updates of a step are independent accumulators, so the matrix pipe is never waiting on a dependent result;
//   * two workgroups per CU (68 KiB + 8 KiB of LDS, <= 256 VGPRs): one's HBM phases (tile load / store) overlap the other's MFMAs.
// Per 16 rows and n = 1024: 8320 MFMAs, 32 KiB read + 8 KiB x (1 + 2 + 3 + 4) re-read from L2/MALL + 32 KiB written.
// LDS row stride of a stage (elements).  The operand read of a lane is ONE 16-byte (fp64) / 8-byte (fp32) access at row 4 q + fk, column
// pair 2 fr; the LDS serves a wave's ds_read_b128 in the four lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md,
// LDS): the eight lanes of row fk + 1 in a group sit BETWEEN the two runs of four lanes of row fk, so the group is conflict free exactly
// when the row stride is a multiple of 256 bytes -- no padding.  (Rounds 1-4 read 8 bytes per lane, lanes 0-31 together, and wanted the
// rows 128 bytes apart modulo 256: stride 272.  Round 5 paired the tiles -- 16-byte reads -- and kept the padding: every operand read of
// the fp64 kernel was a 2-way conflict, SQ_LDS_BANK_CONFLICT = 3.8 cycles per LDS instruction cycle in profiles/round5_pmc_trsm_fused.json.)
// fp32 reads 8 bytes per lane (lanes 0-31 = rows fk, fk + 1 together: the two rows have to sit 128 bytes apart modulo 256 -- stride 288;
// 272 left them 64 bytes apart, half of the lanes in a 2-way conflict: 2.365 -> 2.34 ms at BQRRP's 49152 x 2048 panel).
#ifndef RLHIP_TF_FSTR64
#define RLHIP_TF_FSTR64 256
#endif
#ifndef RLHIP_TF_FSTR32
#define RLHIP_TF_FSTR32 288
#endif
template <typename T> constexpr int fused_stride() { return sizeof(T) == 8 ? RLHIP_TF_FSTR64 : RLHIP_TF_FSTR32; }
// two stages of HPR = 32 rows of U (one rendezvous per 32 rows) + two inverses
template <typename T, int HPR> constexpr int fused_lds_bytes() { return 2 * HPR * fused_stride<T>() * (int)sizeof(T) + 2 * 32 * 32 * (int)sizeof(T); }

template <typename T>
__global__ __launch_bounds__(256) void trsm_neg_pack_kernel(int64_t n, int64_t n_pad, const T* __restrict__ U, int64_t ldu, T* __restrict__ Uneg) {
    // Uneg[k * n_pad + p(c)] = -U[k, c] where 32-block(k) < 32-block(c), zero elsewhere; 32 x 32 tiles through LDS (coalesced both ways).
    // p interleaves the two 16-column MFMA tiles of every 32 columns -- column 32 b + 16 h + f sits at 32 b + 2 f + h -- so that ONE 16-byte
    // (fp32: 8-byte) LDS read of the solve kernel fetches a lane's operand for both tiles.
    __shared__ T tile[32][33];
    const int64_t k0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool upper = blockIdx.y < blockIdx.x;
    for (int j = ty; j < 32; j += 8) {
        const int64_t k = k0 + tx, c = c0 + j;
        tile[j][tx] = (upper && k < n && c < n) ? -U[k + c * ldu] : T(0);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) Uneg[(k0 + i) * n_pad + c0 + 2 * (tx & 15) + (tx >> 4)] = tile[tx][i];
}

#ifndef RLHIP_TF_DRAIN
#define RLHIP_TF_DRAIN 0
#endif
template <int N> struct IntC { static constexpr int value = N; };

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NW = wavefronts per workgroup (16 rows each), HPR = rows of U per LDS stage (32 = two 16-row half panels), two stages
// OOP = 1, 2 (0: in place) = out of place: the right-hand side is READ from Bsrc (column c of the solve = column perm[c] - pbase of Bsrc when perm is given:
// CQRRPT's column pivoting folded into the solve, rl_cqrrpt.hh:288-300) and the solution is WRITTEN to B; the solved tiles needed by
// later blocks are re-read from B.  The pivot entries of a tile are wave-uniform (scalar loads), only the choice among the lane
// group's four columns is per lane, so the number of vector-memory requests per step -- which the counted waits rely on -- is unchanged.
// XASM: the X-operand loads of a panel step are issued from inline asm behind the kernel's own counted waits (false: plain C++ loads, the
// compiler's waits).  The asm form is only correct as long as the register allocator neither spills nor copies a destination register
// between its issue and the counted wait that covers it -- scripts/check_trsm_asm.py proves that on the disassembly of every build, and
// the build falls back to XASM = false when the proof fails.  gate / ngate: the launch does nothing when any of the ngate device words is
// non-zero (cholqrq below: the Cholesky factorization failed, or a diagonal block failed the conditioning guard).
template <typename T, int NW, int HPR, int OOP, bool XASM>
__global__ __launch_bounds__(NW * 64, 2) void trsm_fused_kernel(int64_t m, int64_t n, int64_t n_pad, T alpha, const T* __restrict__ Uneg,
                                                                const T* __restrict__ Dinv, T* __restrict__ B, int64_t ldb, int J0, int J1,
                                                                int K0blk, T* __restrict__ dump, const T* Bsrc, int64_t ldsrc,
                                                                const int64_t* __restrict__ perm, int64_t pbase, const int* __restrict__ gate, int ngate) {
    if (gate != nullptr) {
        int closed = 0;
        for (int i = 0; i < ngate; ++i) closed |= gate[i];
        if (closed) return;
    }
    static_assert(HPR == 32, "an LDS stage holds two 16-row half panels");
    using M = BlkMma<T>;
    using acc_t = typename M::acc_t;
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void glb_void_t;
    constexpr int RING = 2;
    constexpr int FSTR = fused_stride<T>();
    constexpr int DW = RLHIP_TF_DRAIN ? 0 : 8;                  // requests that may fly at a rendezvous of the diagonal block (see dpair)
    constexpr int HPB = HPR * FSTR * (int)sizeof(T);            // bytes per stage (16 rows: 34 KiB fp64, 17 KiB fp32)
    constexpr int NCH = HPB / 1024;                             // 1 KiB DMA pieces per panel
    constexpr int EPC = 1024 / (int)sizeof(T);                  // elements per piece
    constexpr int EPL = 16 / (int)sizeof(T);                    // elements per lane of a piece
    constexpr int P = (NCH + NW - 1) / NW;                      // pieces per wave and panel -- the SAME for every wave (a wave whose last
                                                                // slot falls beyond the panel re-fetches an earlier piece), so that the
                                                                // counted s_waitcnt vmcnt(...) below are compile-time constants
    constexpr int DCH = 32 * 32 * (int)sizeof(T) / 1024;        // pieces of one 32 x 32 inverse (one per wave, duplicates allowed)
    constexpr int NQ = 4;                                       // k-steps (MFMA depth 4) per 16-row half panel
    constexpr int PPB = 16;                                     // half panels per 256-block
    constexpr int DBY = 32 * 32 * (int)sizeof(T);               // bytes of one inverse
    extern __shared__ __attribute__((aligned(1024))) unsigned char tf_smem[];
    unsigned char* sD = tf_smem + RING * HPB;                   // two stages: the inverse of diagonal sub-block s lives in stage s & 1
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * NW + wid) * 16 + fr;
    const bool live = row < m;
    // tile element (j, r) of this lane sits in column 16 j + CS * r + CL * fk: a wave-uniform column (scalar base address) plus the lane
    // offset `loff` (32 bits: the host takes this path only while 4 ldb + m < 2^28)
    constexpr int CS = M::CS, CL = M::CL;
    const unsigned loff = (unsigned)((live ? row : m - 1) + (int64_t)CL * fk * ldb);
    // X operand of a k-step: column 4 q + fk of the panel.  A BYTE offset (< 2^31, see the host) so that uniform pointer + zero-extended lane
    // offset is an address the scalar-base form of global_load takes: no per-lane 64-bit pointers to keep (hipcc spilled and reloaded them)
    const unsigned xoffb = (unsigned)(((live ? row : m - 1) + (int64_t)fk * ldb) * (int64_t)sizeof(T));
    // The loads are issued from inline asm in that form and tracked by the kernel's own counted waits (they are the oldest requests of a
    // step, the P panel pieces the youngest, so `s_waitcnt vmcnt(P)` at the top of the next step covers them).  Left to hipcc the four
    // lane pointers lived in scratch and every panel step reloaded each one behind an s_waitcnt vmcnt(0) -- which also drained the panel
    // pieces in flight -- and the register copy xc = xn at the end of a step forced one more full drain: ~0.9 us of a 4.4 us step.
    auto xissue = [&](const T* ubase, T (&dst)[NQ]) {
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
            const T* bq = ubase + 4 * qq * ldb;                 // uniform
            if constexpr (!XASM) dst[qq] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(bq) + xoffb);
            else if constexpr (sizeof(T) == 8) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=&v"(dst[qq]) : "v"(xoffb), "s"(bq) : "memory");   // (s_nop: the base may have been written by the instruction before -- hipcc does not look into asm for that hazard)
            else asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=&v"(dst[qq]) : "v"(xoffb), "s"(bq) : "memory");
        }
    };
    // after a counted wait: the registers the asm loads filled are defined from here on (nothing hipcc scheduled earlier may stand in for them)
    auto xlanded = [&](T (&v)[NQ]) {
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) asm volatile("" : "+v"(v[qq]));
    };
    // raw right-hand-side element of this lane in tile (jj, r) of the 256-block starting at column cb0
    // OOP: the source column of every column of a 256-block is staged in LDS once per block (identity without a pivot vector), so a tile
    // element is ONE lane-indexed LDS read + ONE global load.  (With four scalar loads of the pivot entries and a per-lane choice per element
    // hipcc built ~390 branches with an s_waitcnt vmcnt(0) in most of them: the eight loads of a retired tile went out one HBM round trip at a time.)
    // OOP = 2: no pivot vector -- column c of the source is column c: the loads are the in-place kernel's with another base (no LDS look-up, no
    // 64-bit multiply per element; CQRRPT's second solve, W -> A: 18.35 -> in-place speed)
    __shared__ int s_perm[OOP == 1 ? 2 : 1][OOP == 1 ? 256 : 1];          // (source column indices: 32 bits -- as 64-bit words the tile loads' temporaries spilled)
    auto fill_perm = [&](int Jb) {
        if constexpr (OOP == 1) {
            if (threadIdx.x < 256) {
                int64_t cidx = (int64_t)Jb * 256 + threadIdx.x;
                if (cidx > n - 1) cidx = n - 1;
                s_perm[Jb & 1][threadIdx.x] = perm ? (int)(perm[cidx] - pbase) : (int)cidx;
            }
        }
    };
    const int64_t rowc = live ? row : m - 1;
    const unsigned loffsrc = (unsigned)(rowc + (int64_t)CL * fk * ldsrc);      // (OOP = 2; the host takes it only while 4 ldsrc + m < 2^28, as for loff)
    // raw right-hand-side element of this lane in tile (jj, r) of the 256-block starting at column cb0
    auto load_raw = [&](int64_t cb0, int jj, int r) -> T {
        if constexpr (OOP == 0) return (B + (cb0 + 16 * jj + CS * r) * ldb)[loff];
        else if constexpr (OOP == 2) return (Bsrc + (cb0 + 16 * jj + CS * r) * ldsrc)[loffsrc];
        else {
            const long long mc = (long long)s_perm[(int)(cb0 >> 8) & 1][16 * jj + CS * r + CL * fk];
            return Bsrc[mc * ldsrc + rowc];
        }
    };
    // per-lane source offsets (elements, relative to the panel's first element) and LDS byte offsets of this wave's DMA pieces
    // UNI (unpadded rows that are whole pieces): piece wid + NW i of a stage is piece wid of rows NW / PPR further down -- ONE lane offset
    // for all P requests of a wave, the row step goes into the uniform base (scalar adds): P - 1 fewer live VGPRs in a kernel that sits at
    // the 256-register limit (the padded stride needs a lane offset per piece: a row boundary falls inside the pieces)
    constexpr int PPR = (FSTR % EPC == 0) ? FSTR / EPC : 0;     // pieces per row
    constexpr bool UNI = PPR > 0 && NCH % NW == 0 && NW % (PPR > 0 ? PPR : 1) == 0;
    constexpr int PV = UNI ? 1 : P;
    unsigned poff[PV];                                          // BYTE offsets, unsigned 32-bit: with the uniform panel pointer they make a scalar-base + lane-offset address
    int pdst[P];                                                // (kept as 64-bit element offsets hipcc spilled them and reloaded each one, behind an s_waitcnt vmcnt(0), in every panel step)
    int64_t pstep = 0;                                          // UNI: bytes between the rows of consecutive pieces of a wave (uniform)
    if constexpr (UNI) {
        poff[0] = (unsigned)(((int64_t)(wid / PPR) * n_pad + (wid % PPR) * EPC + lane * EPL) * (int64_t)sizeof(T));
        pstep = (int64_t)(NW / PPR) * n_pad * (int64_t)sizeof(T);
#pragma unroll
        for (int i = 0; i < P; ++i) pdst[i] = (wid + NW * i) * 1024;
    } else {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            int c = wid + NW * i;
            if (c >= NCH) c -= NCH;                             // duplicate of an earlier piece (same bytes to the same place)
            const int e = c * EPC + lane * EPL;
            const int pr = e / FSTR;
            int pc = e - pr * FSTR;
            if (pc >= 256) pc = 0;                              // padding columns: any valid address
            poff[UNI ? 0 : i] = (unsigned)((pr * n_pad + pc) * (int64_t)sizeof(T));
            pdst[i] = c * 1024;
        }
    }
    const int dsrc = (wid % DCH) * EPC + lane * EPL, ddst = (wid % DCH) * 1024;
    auto issue_panel = [&](const T* base, int buf) {            // P pieces of the panel whose first element is `base` into ring stage `buf`
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const char* bi = UNI ? reinterpret_cast<const char*>(base) + (int64_t)i * pstep : reinterpret_cast<const char*>(base);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(bi + poff[UNI ? 0 : i]), (lds_void_t*)(tf_smem + buf * HPB + pdst[i]), 16, 0, 0);
        }
    };
    auto issue_dinv = [&](int64_t sblk) {                       // ONE piece per wave: inverse of global diagonal sub-block sblk -> stage sblk & 1
        __builtin_amdgcn_global_load_lds((glb_void_t*)(Dinv + sblk * 1024 + dsrc), (lds_void_t*)(sD + (int)(sblk & 1) * DBY + ddst), 16, 0, 0);
    };
    // One half panel: acc[j] += Uneg-panel(rows 4q + fk, tile j) * y[q] for the tiles j >= JLO (even), in groups of four products whose two
    // LDS fragments -- one 2-element read per tile PAIR, the packed operand interleaves the pair's columns -- are fetched before the MFMAs
    // of the group in front are queued; `side()` runs behind the first group.
    auto panel_mma = [&](auto jlo_c, acc_t (&acc)[16], const T* sU, const T (&y)[NQ], auto&& side) {
        typedef T v2_t __attribute__((ext_vector_type(2)));
        constexpr int JLO = decltype(jlo_c)::value;
        static_assert(JLO % 2 == 0, "tiles are consumed in pairs");
        constexpr int NP = (16 - JLO) / 2, TOT = NQ * NP, G = 2, NG = (TOT + G - 1) / G;
        const T* su = sU + fk * FSTR + 2 * fr;
        v2_t fa[G], fb[G];
#pragma unroll
        for (int e = 0; e < G; ++e)
            if (e < TOT) fa[e] = *reinterpret_cast<const v2_t*>(su + 4 * (e / NP) * FSTR + 32 * (JLO / 2 + e % NP));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int f = (g + 1) * G + e;
                if (f < TOT) fb[e] = *reinterpret_cast<const v2_t*>(su + 4 * (f / NP) * FSTR + 32 * (JLO / 2 + f % NP));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < G; ++e) {
                const int f = g * G + e;
                if (f < TOT) {
                    constexpr int dummy = 0; (void)dummy;
                    const int j = JLO + 2 * (f % NP);
                    acc[j] = M::mma(fa[e][0], y[f / NP], acc[j]);
                    acc[j + 1] = M::mma(fa[e][1], y[f / NP], acc[j + 1]);
                }
            }
            if (g == 0) side();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < G; ++e) fa[e] = fb[e];
        }
    };
    // first element of panel t of block J (t counts from the first fused block row K0blk; the diagonal block follows seamlessly)
    auto panel_ptr = [&](int J, int t) -> const T* { return Uneg + ((int64_t)K0blk * 256 + (int64_t)t * 16) * n_pad + (int64_t)J * 256; };   // t counts 16-row halves

#ifdef RLHIP_TF_PROF
    long long tfp[20]; for (int i = 0; i < 20; ++i) tfp[i] = 0;
    long long tft = wall_clock64();
#define TF_MARK(i) { const long long now_ = wall_clock64(); tfp[i] += now_ - tft; tft = now_; }
#else
#define TF_MARK(i)
#endif
    acc_t acc[16];
    T xc[NQ], xn[NQ];
    int ring = 0;                                               // ring stage of the panel the next step consumes
    // ---- prologue: the first two panels, the first inverse, the first tile, the first X operands
    if constexpr (OOP == 1) {
        fill_perm(J0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    issue_panel(panel_ptr(J0, 0), 0);
    issue_dinv((int64_t)J0 * 8);
    {
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = load_raw((int64_t)J0 * 256, j, r);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) { xc[q] = T(0); xn[q] = T(0); }
    if (J0 > K0blk) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) xc[q] = T(0);
        xissue(B + (int64_t)K0blk * 256 * ldb, xc);
    }
    for (int J = J0; J < J1; ++J) {
        const int64_t col0 = (int64_t)J * 256;
        const int ntoff = PPB * (J - K0blk);                    // panels of the blocks left of the diagonal block
        const bool has_next = (J + 1 < J1);
        if (has_next) fill_perm(J + 1);                          // read from the third diagonal step on: many rendezvous later
        // the raw tile (loaded in the prologue or behind the previous block's diagonal steps) -> alpha * B_J
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = alpha * acc[j][r];
        const T* xp = B + (int64_t)K0blk * 256 * ldb;                    // first column of panel t (uniform); the lane adds xoff = row + fk * ldb
        const int64_t xstep = (int64_t)16 * ldb;
        // ---- blocks left of the diagonal block: X from memory, all 16 tiles
        // two steps per trip: the operand registers alternate (xc: even panels, xn: odd panels; ntoff is a multiple of 16), no copies.
        // (A rendezvous in the MIDDLE of a panel with the next panel's first fragments prefetched across the boundary, as in the stream-K
        // GEMM, was built and measured: 4.18 us per panel step either way -- the boundary bubble is not what the step loses.)
        {
            // ---- 32-row stages (two of them): ONE rendezvous per 32 rows of U.  Pair p = half panels 2p, 2p + 1 in stage `ring`; the stage of
            //      pair p + 1 is requested right behind the rendezvous of pair p (every wave has left the other stage by then) and has a
            //      whole pair (~7 us) to land; the X operands keep their 16-row cadence (xc: even halves, xn: odd halves) and their counted
            //      waits: behind the first half only the P stage pieces are younger than xn's loads.
            for (int t = 0; t < ntoff; t += 2) {
                wait_vm<0>();                                   // stage of this pair (requested one pair ago) and xc (requested half a pair ago)
                xlanded(xc);
                __builtin_amdgcn_s_barrier();
                const T* sU = reinterpret_cast<const T*>(tf_smem + ring * HPB);
                const T* nbase = panel_ptr(J, t + 2);           // (always inside this block: the diagonal halves follow)
                xp += xstep;
                panel_mma(IntC<0>{}, acc, sU, xc, [&]() {
                    xissue(xp, xn);
                    issue_panel(nbase, ring ^ 1);
                });
                wait_vm<P>();
                xlanded(xn);
                xp += xstep;
                const bool more_x = (t + 2 < ntoff);
                panel_mma(IntC<0>{}, acc, sU + 16 * FSTR, xn, [&]() {
                    if (more_x) xissue(xp, xc);
                });
                ring ^= 1;
            }
            TF_MARK(0)
        }
        // ---- diagonal block: right-looking over the 32-column sub-blocks, X in registers
        auto solve_sub = [&](int s) {                           // X_s = T_s * inv(U_ss)   (compile-time s after unrolling)
            const T* dv = reinterpret_cast<const T*>(sD + (s & 1) * DBY) + fr;
            acc_t xs0 = acc_t{0, 0, 0, 0}, xs1 = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                const T y = M::operand(acc[2 * s + (c >> 4)], c & 15, lane);
                const T d0 = dv[(c + fk) * 32], d1 = dv[(c + fk) * 32 + 16];
                if (c < 16) xs0 = M::mma(d0, y, xs0);           // the inverse is upper triangular: rows >= 16 do not reach columns < 16
                xs1 = M::mma(d1, y, xs1);
            }
            acc[2 * s] = xs0; acc[2 * s + 1] = xs1;
        };
        // sub-block s is final: its two tiles go back to B and the registers take the same tiles of the NEXT block (raw, scaled at
        // that block's start).  ALWAYS 8 stores + 8 loads per wave, so that the counted waits stay exact: rows / columns outside the
        // matrix store to a scratch line, the last block re-loads its own tile (values unused).
        auto retire_store = [&](int s) {
            T* bj = B + col0 * ldb;
            T* dl = dump + threadIdx.x;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T* dst = bj + (int64_t)(16 * (2 * s + u) + CS * r) * ldb + loff;
                    *(live ? dst : dl) = acc[2 * s + u][r];
                }
        };
        auto retire_load = [&](int s) {
            const int64_t cbn = has_next ? col0 + 256 : col0;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[2 * s + u][r] = load_raw(cbn, 2 * s + u, r);
        };
        auto retire_tiles = [&](int s) { retire_store(s); retire_load(s); };
        {
            // diagonal block with 32-row stages: one rendezvous per 32-column sub-block.  Requests of a pair in issue order: the 8 STORES of the
            // tiles retired there, the next stage's P pieces, one piece of the next inverse, then the 8 LOADS that refill the retired
            // registers -- the only requests that may still fly at the next rendezvous (DW = 8).  That count is only right while the 8
            // loads really SIT behind the LDS-DMA pieces in the instruction stream, and they are hipcc's to place: with Bsrc declared
            // `const __restrict__` it treated them as invariant and sank all 64 of a block to the block's end, across the waits, so that
            // vmcnt(8) let 8 of the 10 pieces fly -- seen as a wrong second tile of a sub-block in some wavefronts of the out-of-place solve.
            // Bsrc is a plain pointer now (a load may not cross a wait's memory clobber), and scripts/check_trsm_asm.py proves on every
            // build that no LDS-DMA piece is outstanding at any s_barrier; if that proof fails the Makefile rebuilds with
            // -DRLHIP_TF_DRAIN=1 (every rendezvous of the diagonal block drains the counter) and proves that build.
            auto dpair = [&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                wait_vm<(s >= 2) ? DW : 0>();
                __builtin_amdgcn_s_barrier();
                const T* sU = reinterpret_cast<const T*>(tf_smem + ring * HPB);
                solve_sub(s);
                T y[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) y[q] = M::operand(acc[2 * s], 4 * q, lane);
                panel_mma(IntC<2 * s + 2>{}, acc, sU, y, [&]() {
                    if (s >= 1) retire_store(s - 1);
                    issue_panel((s + 1 < 7) ? panel_ptr(J, ntoff + 2 * (s + 1)) : (has_next ? panel_ptr(J + 1, 0) : panel_ptr(J, 0)), ring ^ 1);
                    issue_dinv((int64_t)J * 8 + s + 1);                                      // other stage: read last by solve_sub(s - 1)
                    if (s >= 1) retire_load(s - 1);
                });
#pragma unroll
                for (int q = 0; q < NQ; ++q) y[q] = M::operand(acc[2 * s + 1], 4 * q, lane);
                panel_mma(IntC<2 * s + 2>{}, acc, sU + 16 * FSTR, y, [&]() {});
                ring ^= 1;
                TF_MARK(1 + 2 * s)
            };
            dpair(IntC<0>{}); dpair(IntC<1>{}); dpair(IntC<2>{}); dpair(IntC<3>{}); dpair(IntC<4>{}); dpair(IntC<5>{}); dpair(IntC<6>{});
            wait_vm<DW>();                                      // the last inverse and the next block's first stage have landed; the 8 loads of retire(5) may fly
            __builtin_amdgcn_s_barrier();
            solve_sub(7);
            retire_tiles(6);
            retire_tiles(7);
            TF_MARK(15)
            if (has_next) {
                issue_dinv((int64_t)(J + 1) * 8);
                xissue(B + (int64_t)K0blk * 256 * ldb, xc);
            }
        }
    }
#ifdef RLHIP_TF_PROF
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) for (int i = 0; i < 16; ++i) ((long long*)(dump + 512))[i] = tfp[i];
#endif
}

// *bad = 1 unless perm[0..n) - pbase is a permutation of 0..n-1 (seen: n bits, zeroed by the caller)
// (nsrc: number of source columns the entries may name; the whole-solve callers pass n -- a permutation of 1 .. n)
__global__ void perm_check_kernel(int64_t n, const int64_t* __restrict__ perm, int64_t pbase, unsigned* __restrict__ seen, int* __restrict__ bad, int64_t nsrc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t sidx = perm[i] - pbase;
    if (sidx < 0 || sidx >= nsrc) { atomicExch(bad, 1); return; }
    const unsigned bit = 1u << (sidx & 31);
    if (atomicOr(seen + (sidx >> 5), bit) & bit) atomicExch(bad, 1);
}

// dst[:, c] = src[:, perm[c] - pbase] (perm == nullptr: plain copy); threads along rows
template <typename T>
__global__ __launch_bounds__(256) void gather_cols_kernel(int64_t m, int64_t n, const T* __restrict__ src, int64_t lds_, const int64_t* __restrict__ perm,
                                                          int64_t pbase, T* __restrict__ dst, int64_t ldd) {
    const int64_t c = blockIdx.y;
    const int64_t sc = perm ? perm[c] - pbase : c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) dst[i + c * ldd] = src[i + sc * lds_];
}


// RLHIP_TF_XASM_DEFAULT comes from the build (Makefile: 1 when scripts/check_trsm_asm.py has proven the asm-issued loads of THIS build safe,
// else 0); RLHIP_TRSM_XASM = 0 / 1 overrides it per call (tests compare the two builds of the kernel bit for bit).
#ifndef RLHIP_TF_XASM_DEFAULT
#define RLHIP_TF_XASM_DEFAULT 1
#endif
// rows from which the one-launch fused solve is taken (>= one 64-row workgroup per CU); shorter inputs run the per-block kernels
constexpr int64_t TF_MIN_ROWS = 16384;
inline bool tf_xasm(const rlhip_ctx* c) {      // RLHIP_OPT_TRSM_XASM: -1 = what the build proved (Makefile: scripts/check_trsm_asm.py)
    const int64_t o = c->opt[RLHIP_OPT_TRSM_XASM];
    return o < 0 ? (RLHIP_TF_XASM_DEFAULT != 0) : (o != 0);
}

template <typename T, int NW, int OOP, int HPR>
int tf_launch_nw(rlhip_ctx* c, int64_t m, int64_t n, int64_t n_pad, T alpha, const T* Uneg, const T* Dinv, T* B, int64_t ldb, int J0, int J1, int K0blk, T* dump,
                 const T* Bsrc, int64_t ldsrc, const int64_t* perm, int64_t pbase, const int* gate, int ngate) {
    const dim3 grid((unsigned)((m + 16 * NW - 1) / (16 * NW)));
    constexpr int lds = fused_lds_bytes<T, HPR>();
    if (tf_xasm(c)) {
        RLHIP_FUNC_LDS(c, (trsm_fused_kernel<T, NW, HPR, OOP, true>), lds);
        hipLaunchKernelGGL((trsm_fused_kernel<T, NW, HPR, OOP, true>), grid, dim3(64 * NW), lds, c->stream, m, n, n_pad, alpha, Uneg, Dinv, B, ldb, J0, J1, K0blk,
                           dump, Bsrc, ldsrc, perm, pbase, gate, ngate);
    } else {
        RLHIP_FUNC_LDS(c, (trsm_fused_kernel<T, NW, HPR, OOP, false>), lds);
        hipLaunchKernelGGL((trsm_fused_kernel<T, NW, HPR, OOP, false>), grid, dim3(64 * NW), lds, c->stream, m, n, n_pad, alpha, Uneg, Dinv, B, ldb, J0, J1, K0blk,
                           dump, Bsrc, ldsrc, perm, pbase, gate, ngate);
    }
    RLHIP_LAUNCH_CHECK();
    return 0;
}

// Wavefronts per workgroup.  A launch costs ceil(workgroups / CUs) rounds of ~NW time units (one workgroup per CU, the CU throughput-bound
// inside a round: profiles/round5_late_experiments.txt), so a row count that leaves the last round of 128-row workgroups half empty pays for
// rows it does not have: 49152 rows = 384 workgroups = two rounds = the price of 65536 rows.  The fp32 in-place solve (BQRRP's panels, whose
// row count shrinks by b per iteration) therefore has a 192-row instantiation (twelve wavefronts, three per SIMD: its 148 VGPRs fit) taken
// when its rounds cost less: 49152 rows = 256 workgroups = ONE round of 12 units instead of two of 8.
template <typename T, int OOP, int HPR>
int tf_launch_hpr(rlhip_ctx* c, int64_t m, int64_t n, int64_t n_pad, T alpha, const T* Uneg, const T* Dinv, T* B, int64_t ldb, int J0, int J1, int K0blk, T* dump,
                  const T* Bsrc, int64_t ldsrc, const int64_t* perm, int64_t pbase, const int* gate, int ngate) {
    if constexpr (sizeof(T) == 4 && OOP == 0) {
        const int64_t ncu = c->num_cu > 0 ? c->num_cu : 256;
        auto cost = [&](int64_t nw) { const int64_t wg = (m + 16 * nw - 1) / (16 * nw); return ((wg + ncu - 1) / ncu) * nw; };
        if (cost(12) < cost(8))
            return tf_launch_nw<T, 12, OOP, HPR>(c, m, n, n_pad, alpha, Uneg, Dinv, B, ldb, J0, J1, K0blk, dump, Bsrc, ldsrc, perm, pbase, gate, ngate);
    }
    return tf_launch_nw<T, 8, OOP, HPR>(c, m, n, n_pad, alpha, Uneg, Dinv, B, ldb, J0, J1, K0blk, dump, Bsrc, ldsrc, perm, pbase, gate, ngate);
}

template <typename T, int OOP>
int tf_launch(rlhip_ctx* c, int64_t m, int64_t n, int64_t n_pad, T alpha, const T* Uneg, const T* Dinv, T* B, int64_t ldb, int J0, int J1, int K0blk, T* dump,
              const T* Bsrc, int64_t ldsrc, const int64_t* perm, int64_t pbase, const int* gate, int ngate) {
    return tf_launch_hpr<T, OOP, 32>(c, m, n, n_pad, alpha, Uneg, Dinv, B, ldb, J0, J1, K0blk, dump, Bsrc, ldsrc, perm, pbase, gate, ngate);
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
    // blk path (one MFMA launch per 256-block): needs the explicit inverses of the 32 x 32 diagonal blocks, so it is taken per block
    // only where those are well conditioned (kappa_F <= 1e3: error eps * kappa stays at 1e-13); graded triangles -- the R factor of an
    // ill-conditioned sketch in CQRRPT's preconditioning step -- keep the componentwise-stable substitution kernels.
    const int64_t nblk = (n + BW - 1) / BW;
    const bool try_blk = m >= 16 && n >= 128 && nblk <= 32;   // narrow solves (orhr_col, potrf panels: n = 32) are one pack + one substitution launch already
    T* Upk_all = try_blk ? ws_alloc<T>(c, (size_t)nblk * BW * BW) : nullptr;
    T* Dinv_all = try_blk ? ws_alloc<T>(c, (size_t)nblk * (BW / 32) * 1024) : nullptr;
    int* bad_dev = try_blk ? ws_alloc<int>(c, 32) : nullptr;
    int bad_host[32];
    for (int i = 0; i < 32; ++i) bad_host[i] = 1;
    if (try_blk) {
        if (!Upk_all || !Dinv_all || !bad_dev) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        RLHIP_CHECK(hipMemsetAsync(bad_dev, 0, 32 * sizeof(int), c->stream));
        hipLaunchKernelGGL(trsm_blk_pack_kernel<T>, dim3(BW / 32 + 24, (unsigned)nblk), dim3(256), 0, c->stream, n, diag, A, lda, Upk_all, Dinv_all, bad_dev,
                           1.0e6);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 16, bad_dev, 32 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        for (int i = 0; i < 32; ++i) bad_host[i] = ((int*)(c->h_mail + 16))[i];
    }
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
    // fused path: ONE launch solves a whole run of well-conditioned 256-blocks (trsm_fused_kernel); blocks whose diagonal sub-blocks are
    // ill conditioned still take the substitution kernels below, with a GEMM bringing in everything to their left.
    const bool use_fused = try_blk && m >= TF_MIN_ROWS && n >= BW && (4 * ldb + m) < ((int64_t)1 << 28);   // (32-bit lane offsets in the kernel)
    const int64_t n_pad = nblk * BW;
    T* Uneg = nullptr;
    T* fdump = nullptr;
    if (use_fused) {
        bool any_good = false;
        for (int64_t b = 0; (b + 1) * BW <= n; ++b) any_good |= !bad_host[b];   // (a ragged last block stays on the blk path)
        if (any_good) {
            Uneg = ws_alloc<T>(c, (size_t)n_pad * n_pad);
            if (!Uneg) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
            hipLaunchKernelGGL(trsm_neg_pack_kernel<T>, dim3((unsigned)(n_pad / 32), (unsigned)(n_pad / 32)), dim3(256), 0, c->stream, n, n_pad, A, lda, Uneg);
            RLHIP_LAUNCH_CHECK();
            fdump = ws_alloc<T>(c, 512 + 64);
            if (!fdump) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        }
    }
    for (int64_t j0 = 0; j0 < n; j0 += DB) {
        const int nb = (int)((n - j0 < DB) ? (n - j0) : DB);
        T a = alpha;
        if (Uneg && !bad_host[j0 / BW] && j0 + BW <= n) {
            // run of good blocks [Jb, Je): everything left of Jb comes in through one GEMM, the rest is fused
            const int Jb = (int)(j0 / BW);
            int Je = Jb;
            while (Je < nblk && !bad_host[Je] && (int64_t)(Je + 1) * BW <= n) ++Je;
            const int64_t jend = (Je * (int64_t)BW < n) ? Je * (int64_t)BW : n;
            if (j0 > 0) {
                int rc = gemm_impl<T>(c, 0, 0, m, jend - j0, j0, T(-1), B, ldb, A + j0 * lda, lda, alpha, B + j0 * ldb, ldb, 0);
                if (rc) { rlhip_ws_release(c, mark); return rc; }
                a = T(1);
            }
            {
                const int lrc = tf_launch<T, 0>(c, m, n, n_pad, a, Uneg, Dinv_all, B, ldb, Jb, Je, Jb, fdump, (const T*)nullptr, (int64_t)0, (const int64_t*)nullptr, (int64_t)0,
                                                    (const int*)nullptr, 0);
                if (lrc) { rlhip_ws_release(c, mark); return lrc; }
            }
#ifdef RLHIP_TF_PROF
            {
                long long pf[16];
                rlhip_stream_sync(c);
                hipMemcpy(pf, fdump + 512, sizeof(pf), hipMemcpyDeviceToHost);
                fprintf(stderr, "[trsm prof, one workgroup, us over %d blocks] off-diagonal %.1f | diag steps", Je - Jb, pf[0] / 100.0);
                for (int i = 1; i <= 14; ++i) fprintf(stderr, " %.1f", pf[i] / 100.0);
                fprintf(stderr, " | tail %.1f\n", pf[15] / 100.0);
            }
#endif
            c->path_count[2]++;
            j0 = (int64_t)(Je - 1) * DB;     // the loop increment moves on to block Je
            continue;
        }
        if (j0 > 0) {
            int rc = gemm_impl<T>(c, 0, 0, m, nb, j0, T(-1), B, ldb, A + j0 * lda, lda, alpha, B + j0 * ldb, ldb, 0);
            if (rc) { rlhip_ws_release(c, mark); return rc; }
            a = T(1);
        }
        if (try_blk && !bad_host[j0 / BW]) {
            // (narrower pieces, 128 / 64 columns, let the fused kernel solve pieces of the block with a GEMM in between -- more of the
            // flops at GEMM speed).  Measured at C3 (m = 2^20, n = 1024): CQRRPT 105.5 ms whole blocks, 107.8 ms halves, 117.2 ms quarters:
            // the extra passes over B cost more than the fused kernel's lower rate, so whole blocks stay the default.
            constexpr int hb_env = BW;
            const T* Upk_b = Upk_all + (j0 / BW) * (int64_t)BW * BW;
            const T* Dinv_b = Dinv_all + (j0 / BW) * (int64_t)(BW / 32) * 1024;
            for (int h0 = 0; h0 < nb; h0 += hb_env) {
                const int hb = (nb - h0 < hb_env) ? (nb - h0) : hb_env;
                T ah = a;
                if (h0 > 0) {
                    const int64_t jc = j0 + h0;
                    // columns [j0, jc) of this block (the columns left of j0 went in above, with alpha)
                    int rc = gemm_impl<T>(c, 0, 0, m, hb, h0, T(-1), B + j0 * ldb, ldb, A + j0 + jc * lda, lda, a, B + jc * ldb, ldb, 0);
                    if (rc) { rlhip_ws_release(c, mark); return rc; }
                    ah = T(1);
                }
                if (m >= 65536)       // enough rows to fill the chip with 128-row workgroups: two row tiles per wave halve the U-fragment traffic (four: slower, 108.6 vs 105.5 ms at C3)
                    hipLaunchKernelGGL((trsm_blk_kernel<T, 2>), dim3((unsigned)((m + 127) / 128)), dim3(256), 0, c->stream, m, h0 + hb, ah, Upk_b, Dinv_b,
                                       B + j0 * ldb, ldb, h0 / 32, h0);
                else
                    hipLaunchKernelGGL((trsm_blk_kernel<T, 1>), dim3((unsigned)((m + 63) / 64)), dim3(256), 0, c->stream, m, h0 + hb, ah, Upk_b, Dinv_b,
                                       B + j0 * ldb, ldb, h0 / 32, h0);
                RLHIP_LAUNCH_CHECK();
            }
            continue;
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
            c->path_count[3]++;
        }
    }
    rlhip_ws_release(c, mark);
    return 0;
}

// Out-of-place solve with the column pivoting folded in:  B = alpha * (Bsrc * P) * inv(A),  column c of Bsrc * P = column perm[c] - 1
// of Bsrc (perm: LAPACK-style 1-based pivot vector on the device, or nullptr for P = I).  CQRRPT's "permute A, then precondition it"
// (rl_cqrrpt.hh:288-300) is one pass over A this way instead of two, and its second solve writes Q straight back into A.
// Taken in ONE fused launch when every 256-block passes the conditioning guard and n is a multiple of 256; otherwise the columns are
// gathered into B by a copy kernel and the in-place solver above runs on B.
template <typename T>
int trsm_right_upper_oop(rlhip_ctx* c, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, const T* Bsrc, int64_t ldsrc,
                         const int64_t* perm_dev, T* B, int64_t ldb) {
    if (m < 0) return -6;
    if (n < 0) return -7;
    if (lda < (n > 1 ? n : 1)) return -10;
    if (ldsrc < (m > 1 ? m : 1)) return -12;
    if (ldb < (m > 1 ? m : 1)) return -15;
    if (m == 0 || n == 0) return 0;
    if ((const T*)B == Bsrc) { if (perm_dev) return -14; return trsm_right_upper<T>(c, diag, m, n, alpha, A, lda, B, ldb); }
    const int64_t nblk = (n + BW - 1) / BW;
    bool fused = m >= TF_MIN_ROWS && n >= BW && n % BW == 0 && nblk <= 32 && (4 * ldb + m) < ((int64_t)1 << 28);
    size_t mark = rlhip_ws_mark(c);
    bool perm_checked = false;
    if (fused) {
        T* Upk_all = ws_alloc<T>(c, (size_t)nblk * BW * BW);
        T* Dinv_all = ws_alloc<T>(c, (size_t)nblk * (BW / 32) * 1024);
        int* bad_dev = ws_alloc<int>(c, 40);
        unsigned* seen = ws_alloc<unsigned>(c, (size_t)n / 32 + 2);
        T* Uneg = ws_alloc<T>(c, (size_t)n * n);
        T* fdump = ws_alloc<T>(c, 512);
        if (!Upk_all || !Dinv_all || !bad_dev || !seen || !Uneg || !fdump) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        auto chk = [&](hipError_t e) { if (e != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(e); } return 0; };   // every error exit gives the arena mark back
        if (int rc = chk(hipMemsetAsync(bad_dev, 0, 33 * sizeof(int), c->stream))) return rc;
        hipLaunchKernelGGL(trsm_blk_pack_kernel<T>, dim3(BW / 32 + 24, (unsigned)nblk), dim3(256), 0, c->stream, n, diag, A, lda, Upk_all, Dinv_all, bad_dev,
                           1.0e6);
        if (int rc = chk(hipGetLastError())) return rc;
        if (perm_dev) {     // the pivot vector is validated on the device; its verdict rides on the guard's read-back (slot 32)
            if (int rc = chk(hipMemsetAsync(seen, 0, ((size_t)n + 31) / 32 * sizeof(unsigned), c->stream))) return rc;
            hipLaunchKernelGGL(perm_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, perm_dev, (int64_t)1, seen, bad_dev + 32, n);
            if (int rc = chk(hipGetLastError())) return rc;
            perm_checked = true;
        }
        // The verdict (33 words: the conditioning guard of every block, the pivot check) travels to the host while the device runs on: the
        // packed triangle and the fused solve are enqueued right behind the read-back, GATED on the same device words (a launch that finds
        // one of them set does nothing), and the host waits for the read-back only -- no idle device during the round trip.
        if (int rc = chk(hipMemcpyAsync(c->h_mail + 16, bad_dev, 33 * sizeof(int), hipMemcpyDeviceToHost, c->stream))) return rc;
        if (int rc = chk(hipEventRecord(c->ev_flag, c->stream))) return rc;
        hipLaunchKernelGGL(trsm_neg_pack_kernel<T>, dim3((unsigned)(n / 32), (unsigned)(n / 32)), dim3(256), 0, c->stream, n, n, A, lda, Uneg);
        if (int rc = chk(hipGetLastError())) return rc;
        {
            // (no pivot vector: the identity-column twin of the kernel, when the source's lane offsets fit 32 bits too)
            const bool ident = !perm_dev && (4 * ldsrc + m) < ((int64_t)1 << 28);
            const int lrc = ident ? tf_launch<T, 2>(c, m, n, n, alpha, Uneg, Dinv_all, B, ldb, 0, (int)nblk, 0, fdump, Bsrc, ldsrc, perm_dev, (int64_t)1, (const int*)bad_dev, 33)
                                  : tf_launch<T, 1>(c, m, n, n, alpha, Uneg, Dinv_all, B, ldb, 0, (int)nblk, 0, fdump, Bsrc, ldsrc, perm_dev, (int64_t)1, (const int*)bad_dev, 33);
            if (lrc) { rlhip_ws_release(c, mark); return lrc; }
        }
        if (int rc = chk(hipEventSynchronize(c->ev_flag))) return rc;
        if (perm_checked && ((int*)(c->h_mail + 16))[32] != 0) { rlhip_ws_release(c, mark); return -7; }     // jpvt is not a permutation of 1..n (as col_swap reports it)
        for (int64_t b = 0; b < nblk; ++b) fused = fused && ((int*)(c->h_mail + 16))[b] == 0;
        if (fused) {
            c->path_count[4]++;
            rlhip_ws_release(c, mark);
            return 0;
        }
    }
    rlhip_ws_release(c, mark);
    if (perm_dev && !perm_checked) {     // gather-copy route: validate before anything is written
        size_t mk2 = rlhip_ws_mark(c);
        unsigned* seen = ws_alloc<unsigned>(c, (size_t)n / 32 + 2);
        int* bad = ws_alloc<int>(c, 8);
        if (!seen || !bad) { rlhip_ws_release(c, mk2); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        hipError_t pe = hipMemsetAsync(seen, 0, ((size_t)n + 31) / 32 * sizeof(unsigned), c->stream);
        if (pe == hipSuccess) pe = hipMemsetAsync(bad, 0, sizeof(int), c->stream);
        if (pe == hipSuccess) {
            hipLaunchKernelGGL(perm_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, perm_dev, (int64_t)1, seen, bad, n);
            pe = hipGetLastError();
        }
        if (pe == hipSuccess) pe = hipMemcpyAsync(c->h_mail + 16, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (pe == hipSuccess) pe = rlhip_stream_sync(c);
        rlhip_ws_release(c, mk2);
        if (pe != hipSuccess) return RLHIP_ERR_HIP(pe);
        if (*(int*)(c->h_mail + 16) != 0) return -7;
    }
    {
        unsigned gx = (unsigned)((m + 256 * 8 - 1) / (256 * 8));
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(gather_cols_kernel<T>, dim3(gx, (unsigned)n), dim3(256), 0, c->stream, m, n, Bsrc, ldsrc, perm_dev, (int64_t)1, B, ldb);
        RLHIP_LAUNCH_CHECK();
    }
    return trsm_right_upper<T>(c, diag, m, n, alpha, A, lda, B, ldb);
}

// A RANGE of 256-blocks of the same out-of-place solve: columns [col0, col1) of  B = alpha * (Bsrc * P) * inv(A)  given that B[:, 0 : col0)
// already holds the solution's leading columns (col0 = 0: none).  Only the leading col1 x col1 part of A and perm[0 : col1) are read, and perm
// maps into `nsrc` source columns (a PREFIX of a pivot vector is not a permutation of 1 .. col1).  This is what lets CQRRPT solve for the
// first half of the preconditioned matrix while the second half of the sketch is still being factored (rl_cqrrpt.hh: the leading block of R
// and the leading pivots are final after half the steps of the pivoted QR).  The fused kernel walks block columns left to right and re-reads
// the solved blocks from B, so a range is the same launch with other bounds: the pieces of a split solve are bitwise the whole solve.
// Returns 0 (done), 1 (not taken: sizes or conditioning outside the fused kernel's domain -- nothing written, the caller takes the whole
// solve), or an error (< 0; -7: perm is not injective into 1 .. nsrc).
template <typename T>
int trsm_right_upper_oop_range(rlhip_ctx* c, int diag, int64_t m, int64_t nsrc, T alpha, const T* A, int64_t lda, const T* Bsrc, int64_t ldsrc,
                               const int64_t* perm_dev, T* B, int64_t ldb, int64_t col0, int64_t col1) {
    if (m < 0) return -6;
    if (col0 < 0 || col1 <= col0 || col0 % BW || col1 % BW) return -7;
    if (nsrc < col1) return -7;
    if (lda < col1) return -10;
    if (ldsrc < (m > 1 ? m : 1)) return -12;
    if (ldb < (m > 1 ? m : 1)) return -15;
    if (m == 0) return 0;
    if ((const T*)B == Bsrc) return -14;
    const int64_t n = col1, nblk = n / BW;
    if (m < TF_MIN_ROWS || nblk > 32 || (4 * ldb + m) >= ((int64_t)1 << 28)) return 1;
    size_t mark = rlhip_ws_mark(c);
    T* Upk_all = ws_alloc<T>(c, (size_t)nblk * BW * BW);
    T* Dinv_all = ws_alloc<T>(c, (size_t)nblk * (BW / 32) * 1024);
    int* bad_dev = ws_alloc<int>(c, 40);
    unsigned* seen = ws_alloc<unsigned>(c, (size_t)nsrc / 32 + 2);
    T* Uneg = ws_alloc<T>(c, (size_t)n * n);
    T* fdump = ws_alloc<T>(c, 512);
    if (!Upk_all || !Dinv_all || !bad_dev || !seen || !Uneg || !fdump) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    hipError_t he = hipMemsetAsync(bad_dev, 0, 33 * sizeof(int), c->stream);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(trsm_blk_pack_kernel<T>, dim3(BW / 32 + 24, (unsigned)nblk), dim3(256), 0, c->stream, n, diag, A, lda, Upk_all, Dinv_all, bad_dev, 1.0e6);
        he = hipGetLastError();
    }
    if (he == hipSuccess && perm_dev) {
        he = hipMemsetAsync(seen, 0, ((size_t)nsrc + 31) / 32 * sizeof(unsigned), c->stream);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(perm_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, perm_dev, (int64_t)1, seen, bad_dev + 32, nsrc);
            he = hipGetLastError();
        }
    }
    // (the verdict is read back while the device runs on: the solve is enqueued behind the read-back and gated on the same words -- see the
    //  whole-matrix form above)
    if (he == hipSuccess) he = hipMemcpyAsync(c->h_mail + 16, bad_dev, 33 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipEventRecord(c->ev_flag, c->stream);
    if (he != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(he); }
    hipLaunchKernelGGL(trsm_neg_pack_kernel<T>, dim3((unsigned)(n / 32), (unsigned)(n / 32)), dim3(256), 0, c->stream, n, n, A, lda, Uneg);
    he = hipGetLastError();
    int lrc = (he == hipSuccess) ? 0 : RLHIP_ERR_HIP(he);
    if (!lrc) lrc = tf_launch<T, 1>(c, m, n, n, alpha, Uneg, Dinv_all, B, ldb, (int)(col0 / BW), (int)nblk, 0, fdump, Bsrc, ldsrc, perm_dev, (int64_t)1, (const int*)bad_dev, 33);
    he = hipEventSynchronize(c->ev_flag);
    if (!lrc && he != hipSuccess) lrc = RLHIP_ERR_HIP(he);
    if (!lrc && perm_dev && ((int*)(c->h_mail + 16))[32] != 0) lrc = -7;
    if (!lrc) {
        bool good = true;
        for (int64_t bb = 0; bb < nblk; ++bb) good = good && ((int*)(c->h_mail + 16))[bb] == 0;
        if (!good) lrc = 1;                       // the gated launch did nothing
    }
    if (!lrc) c->path_count[4]++;
    rlhip_ws_release(c, mark);
    return lrc;
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

template <typename T>
int potrf_upper_enqueue(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int* info_dev);

// Cholesky-QR, Q factor only (RandLAPACK/comps/rl_orth.hh:69-98: syrk -> potrf -> trsm) as ONE stream of kernels with ONE host read:
//   R (k x k, ld k) = chol(A^T A) (upper; the strictly lower part is zero), A <- A R^-1, *info_host = LAPACK's potrf info.
// The factorization leaves its info in a device word; the conditioning guard of the fused solve leaves its verdicts next to it; the
// fused solve is launched unconditionally and does nothing when any of those words is set.  The host reads them all at the end: info != 0
// -> A is untouched (the reference returns before its trsm as well); a guard verdict -> the solve is repeated by the ordinary route
// (substitution where the explicit 32 x 32 inverses are not trustworthy).  `reduce_gram`: row-sharded input, the Gram matrix is summed over
// the ranks before it is factored.  Returns 1 when the shape is not served here (short or ragged inputs): the caller runs the three calls.
template <typename T>
int cholqrq(rlhip_ctx* c, int64_t m, int64_t k, T* A, int64_t lda, T* R, int reduce_gram, int* info_host) {
    const int64_t nblk = (k + BW - 1) / BW;
    // Served or not is decided from RANK-UNIFORM quantities only when the Gram matrix is summed over the ranks: every rank must issue the same
    // collectives with the same counts (shards that straddle the row threshold used to split into an all-reduce of k*k + 1 words here
    // and one of k*k words in the caller's fallback).  What is rank-local -- too few rows for the fused solve, the 32-bit lane offsets, no room
    // for its scratch -- only selects the SOLVE kernel behind the common Gram / all-reduce / Cholesky sequence.
    if (c->opt[RLHIP_OPT_CHOLQRQ_ONE_STREAM] == 0 || k < BW || k % BW != 0 || k > 448 || nblk > 31 || lda < (m > 1 ? m : 1)) return 1;
    bool local_ok = m >= TF_MIN_ROWS && (4 * lda + m) < ((int64_t)1 << 28);
    if (!reduce_gram && !local_ok) return 1;
    *info_host = 0;
    size_t mark = rlhip_ws_mark(c);
    auto fail = [&](int rc) { rlhip_ws_release(c, mark); return rc; };
    int* words = ws_alloc<int>(c, 40);                            // [0] potrf info, [1 .. nblk] guard verdicts
    if (!words) { rlhip_ws_release(c, mark); return reduce_gram ? RLHIP_ERR_HIP(hipErrorOutOfMemory) : 1; }
    T *Upk_all = nullptr, *Dinv_all = nullptr, *Uneg = nullptr, *fdump = nullptr;
    if (local_ok) {
        Upk_all = ws_alloc<T>(c, (size_t)nblk * BW * BW);
        Dinv_all = ws_alloc<T>(c, (size_t)nblk * (BW / 32) * 1024);
        Uneg = ws_alloc<T>(c, (size_t)k * k);
        fdump = ws_alloc<T>(c, 512 + 64);
        if (!Upk_all || !Dinv_all || !Uneg || !fdump) {
            if (!reduce_gram) return fail(1);
            local_ok = false;                                     // (rank-local: the substitution route below needs none of them)
        }
    }
    hipError_t e0 = hipMemsetAsync(words, 0, 40 * sizeof(int), c->stream);
    if (e0 != hipSuccess) return fail(RLHIP_ERR_HIP(e0));
    int rc = laset<T>(c, 2, k, k, T(0), T(0), R, k);
    if (!rc && m > 0) rc = syrk<T>(c, Upper, 1, k, m, T(1), A, lda, T(0), R, k);
    if (!rc && reduce_gram) {
        if (sizeof(T) == 8 && c->norma_state == 1 && !c->norma_reduced) {
            // a deferred ||A||_F^2 is waiting (QB: rl_qb.hh:168): its sum over the ranks rides on THIS all-reduce as word k*k of the buffer instead
            // of taking a scalar collective (and a host round trip) of its own.  (norma_state is set by the product that precedes this call on
            // EVERY rank -- rank-uniform.)
            double* G2 = ws_alloc<double>(c, (size_t)k * k + 1);
            if (!G2) return fail(RLHIP_ERR_HIP(hipErrorOutOfMemory));
            hipError_t e = hipMemcpyAsync(G2, R, (size_t)k * k * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(G2 + (size_t)k * k, (double*)(c->d_mail + 40), sizeof(double), hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) return fail(RLHIP_ERR_HIP(e));
            rc = rlhip_allreduce_sum_f64(c, G2, k * k + 1);
            if (!rc) {
                e = hipMemcpyAsync(R, G2, (size_t)k * k * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(c->h_mail + 41, G2 + (size_t)k * k, sizeof(double), hipMemcpyDeviceToHost, c->stream);
                if (e != hipSuccess) return fail(RLHIP_ERR_HIP(e));
                c->norma_reduced = 1;
            }
        } else {
            rc = (sizeof(T) == 8) ? rlhip_allreduce_sum_f64(c, (double*)R, k * k) : rlhip_allreduce_sum_f32(c, (float*)R, k * k);
        }
    }
    if (!rc) rc = potrf_upper_enqueue<T>(c, k, R, k, words);
    if (rc) return fail(rc < 0 ? rc : RLHIP_ERR_HIP(hipErrorUnknown));
    if (local_ok) {
        hipLaunchKernelGGL(trsm_blk_pack_kernel<T>, dim3(BW / 32 + 24, (unsigned)nblk), dim3(256), 0, c->stream, k, (int)NonUnit, R, k, Upk_all, Dinv_all, words + 1, 1.0e6);
        hipLaunchKernelGGL(trsm_neg_pack_kernel<T>, dim3((unsigned)(k / 32), (unsigned)(k / 32)), dim3(256), 0, c->stream, k, k, R, k, Uneg);
        {
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return fail(RLHIP_ERR_HIP(le));
        }
        rc = tf_launch<T, 0>(c, m, k, k, T(1), Uneg, Dinv_all, A, lda, 0, (int)nblk, 0, fdump, (const T*)nullptr, (int64_t)0, (const int64_t*)nullptr, (int64_t)0, words, 1 + (int)nblk);
        if (rc) return fail(rc);
    }
    hipError_t e1 = hipMemcpyAsync(c->h_mail + 44, words, 40 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e1 == hipSuccess) e1 = rlhip_stream_sync(c);
    rlhip_ws_release(c, mark);
    if (e1 != hipSuccess) return RLHIP_ERR_HIP(e1);
    const int* w = (const int*)(c->h_mail + 44);
    *info_host = w[0];
    if (w[0] != 0) return 0;                                      // not positive definite: A untouched, R partially factored (as dpotrf leaves it)
    bool guard = !local_ok;
    for (int64_t b = 0; b < nblk; ++b) guard = guard || (w[1 + b] != 0);
    if (guard) return trsm_right_upper<T>(c, NonUnit, m, k, T(1), R, k, A, lda);
    c->path_count[2]++;
    c->path_count[11]++;
    return 0;
}
template int cholqrq<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*, int, int*);
template int cholqrq<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*, int, int*);

template int trsm_right_upper<double>(rlhip_ctx*, int, int64_t, int64_t, double, const double*, int64_t, double*, int64_t);
template int trsm_right_upper_oop<double>(rlhip_ctx*, int, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t, const int64_t*, double*, int64_t);
template int trsm_right_upper_oop<float>(rlhip_ctx*, int, int64_t, int64_t, float, const float*, int64_t, const float*, int64_t, const int64_t*, float*, int64_t);
template int trsm_right_upper_oop_range<double>(rlhip_ctx*, int, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t, const int64_t*, double*, int64_t, int64_t, int64_t);
template int trsm_right_upper_oop_range<float>(rlhip_ctx*, int, int64_t, int64_t, float, const float*, int64_t, const float*, int64_t, const int64_t*, float*, int64_t, int64_t, int64_t);
template int trsm_right_upper<float>(rlhip_ctx*, int, int64_t, int64_t, float, const float*, int64_t, float*, int64_t);
template int trmm_right_upper<double>(rlhip_ctx*, int, int64_t, int64_t, double, const double*, int64_t, double*, int64_t);
template int trmm_right_upper<float>(rlhip_ctx*, int, int64_t, int64_t, float, const float*, int64_t, float*, int64_t);

}  // namespace rlhip
