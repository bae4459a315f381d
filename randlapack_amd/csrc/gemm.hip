// MFMA GEMM family for gfx950 (MI355X):  C = alpha * op(A) * op(B) + beta * C,  column-major.
//
// Replaces, on the sketch-and-factor hot path, every blas::gemm the reference issues
// (RandLAPACK/comps/rl_rs.hh:142,153,165; rl_rf.hh:123; rl_qb.hh:210-218,260; drivers/rl_rsvd.hh:148)
// and serves as the engine underneath syrk / trsm / trmm / larfb in this library.
//
// Design (CDNA4-first, not a cuBLAS call pattern):
//  * v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, one 64x64 (or 64x32 / 64x16) output tile per
//    64-lane wavefront, accumulators resident in the unified VGPR/AGPR file.
//  * The MFMA "A" operand is fed with op(B) and the MFMA "B" operand with op(A), so each lane's D
//    elements lie along the contiguous (row) direction of column-major C -> 128-byte store segments.
//  * op(A) / op(B) tiles are staged through LDS with paddings chosen for conflict-free ds_read_b64
//    fragment reads (see lds strides below); global loads are 16-byte, coalesced along whichever index
//    is contiguous in memory for that operand ("MC" = tile-row index contiguous, "KC" = reduction
//    index contiguous).
//  * register prefetch of tile t+1 while tile t is multiplied, double-buffered LDS, ONE barrier per
//    K-tile.
//  * tall reductions (A^T*Q, Gram matrices) use a deterministic split-K: each K-slice writes a slab
//    to scratch and a second kernel sums the slabs in fixed order (bitwise reproducible run to run,
//    unlike atomics).
#include "rlhip_internal.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct Mma;
template <>
struct Mma<double> {
    using acc_t = d4_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 D layout: col = lane & 15, row = (lane >> 4) + 4 * r
    static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mma<float> {
    using acc_t = f4_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // f32 D layout: col = lane & 15, row = 4 * (lane >> 4) + r
    static __device__ __forceinline__ int drow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

template <typename T>
struct GemmArgs {
    int64_t M, N, K;
    const T* A;
    int64_t lda;
    const T* B;
    int64_t ldb;
    T* C;
    int64_t ldc;
    T alpha, beta;
    int64_t kchunk;  // K extent handled by one z-slice (multiple of BK)
    T* slab;         // split-K scratch (M*N per z-slice) or nullptr
    int tri;         // 0: all tiles; 1: skip tiles strictly below the diagonal (syrk upper, M==N, BM==BN)
};

constexpr int PADM = 16;  // MC tile: row stride (R + 16) elements -> (R+16) % 32 == 16 for R % 32 == 0
template <typename T>
struct PadK {
    static constexpr int v = 2;  // KC tile, f64: stride BK+2 == 18 -> i*18 + kq distinct mod 32
};
template <>
struct PadK<float> {
    // stride 18 floats: the fragment read of a half wave (rows fr = 0..15, k = 4s + {0, 1}) touches banks (18 fr + k) mod 32 = all 32
    // banks once.  Stride 20 (16-byte aligned chunks) maps the 16 rows onto 8 bank groups: 2-way conflicts on every fragment read
    // (PMC: SQ_LDS_BANK_CONFLICT = 8 % of the CU cycles per k-contiguous operand, profiles/round1_pmc_gemm_f32.txt).  The staging
    // stores of such a tile are therefore two 8-byte stores per 4-float chunk.
    static constexpr int v = 2;
};

template <typename T, int ROWS, int BK, bool KC>
struct TileGeom {
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int stride = KC ? (BK + PadK<T>::v) : (ROWS + PADM);
    static constexpr int elems = KC ? ROWS * stride : BK * stride;
};

// Loads one operand tile (ROWS x BK logical, element (r, kk)) into registers.
// KC: element (r,kk) at g[kk + r*ld]   MC: element (r,kk) at g[r + kk*ld]
template <typename T, int ROWS, int BK, int NT>
struct RegCnt {
    static constexpr int V = 16 / (int)sizeof(T);
    static constexpr int TCH = ROWS * BK / V;            // 16-byte chunks in the tile
    static constexpr int NCH = (TCH + NT - 1) / NT;      // chunks per thread (last may be idle)
    static constexpr int n = NCH * V;
};

template <typename T, int ROWS, int BK, bool KC, int NT, bool VEC, bool CHECK>
__device__ __forceinline__ void tile_gload(T* reg, const T* __restrict__ g,
                                           int64_t ld, int64_t r0, int64_t k0, int64_t rmax, int64_t kmax,
                                           int tid) {
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int NCH = RegCnt<T, ROWS, BK, NT>::NCH;
    constexpr int TCH = RegCnt<T, ROWS, BK, NT>::TCH;
    constexpr int CPL = KC ? (BK / V) : (ROWS / V);  // chunks per contiguous line
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        int c = tid + u * NT;
        if (TCH % NT != 0 && c >= TCH) break;
        int line = c / CPL;
        int off = (c % CPL) * V;
        int64_t r = KC ? (int64_t)line : (int64_t)off;
        int64_t kk = KC ? (int64_t)off : (int64_t)line;
        const T* p = KC ? (g + (k0 + kk) + (r0 + r) * ld) : (g + (r0 + r) + (k0 + kk) * ld);
        if (VEC && !CHECK) {
            typedef T vec_t __attribute__((ext_vector_type(V)));
            vec_t v = *reinterpret_cast<const vec_t*>(p);
#pragma unroll
            for (int e = 0; e < V; ++e) reg[u * V + e] = v[e];
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                int64_t rr = KC ? (r0 + r) : (r0 + r + e);
                int64_t kq = KC ? (k0 + kk + e) : (k0 + kk);
                bool ok = !CHECK || (rr < rmax && kq < kmax);
                reg[u * V + e] = ok ? p[e] : T(0);
            }
        }
    }
}

template <typename T, int ROWS, int BK, bool KC, int NT>
__device__ __forceinline__ void tile_sstore(const T* reg, T* __restrict__ s,
                                            int tid) {
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int NCH = RegCnt<T, ROWS, BK, NT>::NCH;
    constexpr int TCH = RegCnt<T, ROWS, BK, NT>::TCH;
    constexpr int CPL = KC ? (BK / V) : (ROWS / V);
    constexpr int stride = TileGeom<T, ROWS, BK, KC>::stride;
    typedef T vec_t __attribute__((ext_vector_type(V)));
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        int c = tid + u * NT;
        if (TCH % NT != 0 && c >= TCH) break;
        int line = c / CPL;
        int off = (c % CPL) * V;
        if constexpr (KC && sizeof(T) == 4) {
            typedef T half_t __attribute__((ext_vector_type(2)));
            half_t lo, hi;
            lo[0] = reg[u * V + 0]; lo[1] = reg[u * V + 1]; hi[0] = reg[u * V + 2]; hi[1] = reg[u * V + 3];
            *reinterpret_cast<half_t*>(s + line * stride + off) = lo;
            *reinterpret_cast<half_t*>(s + line * stride + off + 2) = hi;
        } else {
            vec_t v;
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = reg[u * V + e];
            *reinterpret_cast<vec_t*>(s + line * stride + off) = v;
        }
    }
}

// fragment read: element (r, kk) of the staged tile
template <typename T, int ROWS, int BK, bool KC>
__device__ __forceinline__ T tile_frag(const T* __restrict__ s, int r, int kk) {
    constexpr int stride = TileGeom<T, ROWS, BK, KC>::stride;
    return KC ? s[r * stride + kk] : s[kk * stride + r];
}

// blockIdx.x enumerates output tiles.  The hardware deals consecutive workgroup ids round-robin over the 8
// XCDs (id % 8); the remap below hands every XCD a CONTIGUOUS run of logical tile ids, and logical ids run
// N-fastest, so the N-tiles that share one A row-panel (and neighbouring M-tiles that share B) sit on
// the same XCD's L2 at the same time.  Bijective for any tile count (guide section 5, T1).
__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nwg) {
    const int64_t q = nwg / 8, r = nwg % 8;
    const int64_t xcd = bid % 8, slot = bid / 8;
    const int64_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <typename T, bool A_KC, bool B_KC, int BM, int BN, int BK, int WM, int WN, bool VEC, int MINW, bool EARLY, int DBG = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, MINW) void gemm_kernel(GemmArgs<T> g) {
    constexpr int NWM = BM / WM, NWN = BN / WN;
    constexpr int NT = NWM * NWN * 64;
    constexpr int TM = WM / 16, TN = WN / 16;
    using GA = TileGeom<T, BM, BK, A_KC>;
    using GB = TileGeom<T, BN, BK, B_KC>;
    using acc_t = typename Mma<T>::acc_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sA0 = reinterpret_cast<T*>(smem_raw);
    T* sB0 = sA0 + 2 * GA::elems;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid % NWM) * WM;
    const int wn0 = (wid / NWM) * WN;

    const int64_t tiles_n = (g.N + BN - 1) / BN;
    const int64_t lid = xcd_remap((int64_t)blockIdx.x, (int64_t)gridDim.x);
    const int64_t tile_m = lid / tiles_n, tile_n = lid % tiles_n;
    if (g.tri && tile_m > tile_n) return;
    const int64_t m0 = tile_m * BM, n0 = tile_n * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * g.kchunk;
    const int64_t kend = (kbeg + g.kchunk < g.K) ? (kbeg + g.kchunk) : g.K;
    const int64_t nk = (kend - kbeg + BK - 1) / BK;

    const bool full_mn = (m0 + BM <= g.M) && (n0 + BN <= g.N);

    acc_t acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = acc_t{0, 0, 0, 0};

    T ra[RegCnt<T, BM, BK, NT>::n];
    T rb[RegCnt<T, BN, BK, NT>::n];

    auto gload = [&](int64_t kt) {
        if (DBG == 1 && kt > 0) return;  // diagnostic: LDS+MFMA loop without global traffic
        const int64_t k0 = kbeg + kt * BK;
        const bool full = full_mn && (k0 + BK <= kend);
        if (full) {
            tile_gload<T, BM, BK, A_KC, NT, VEC, false>(ra, g.A, g.lda, m0, k0, g.M, kend, tid);
            tile_gload<T, BN, BK, B_KC, NT, VEC, false>(rb, g.B, g.ldb, n0, k0, g.N, kend, tid);
        } else {
            tile_gload<T, BM, BK, A_KC, NT, VEC, true>(ra, g.A, g.lda, m0, k0, g.M, kend, tid);
            tile_gload<T, BN, BK, B_KC, NT, VEC, true>(rb, g.B, g.ldb, n0, k0, g.N, kend, tid);
        }
    };

    if (nk > 0) {
        gload(0);
        tile_sstore<T, BM, BK, A_KC, NT>(ra, sA0, tid);
        tile_sstore<T, BN, BK, B_KC, NT>(rb, sB0, tid);
    }
    __syncthreads();

    const int fr = lane & 15;  // index inside a 16-tile
    const int fk = lane >> 4;  // k offset inside a 4-step

    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = (int)(kt & 1);
        const T* sA = sA0 + cur * GA::elems;
        const T* sB = sB0 + cur * GB::elems;
        const bool more = (kt + 1 < nk);
        if (more) gload(kt + 1);

#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            if (EARLY && s == BK / 8 && more) {
                // stage tile t+1 into the idle LDS buffer while half of this tile's MFMAs are still queued:
                // only the barrier itself (not the LDS write latency) then separates two tiles' MFMA streams
                T* nA = sA0 + (cur ^ 1) * GA::elems;
                T* nB = sB0 + (cur ^ 1) * GB::elems;
                tile_sstore<T, BM, BK, A_KC, NT>(ra, nA, tid);
                tile_sstore<T, BN, BK, B_KC, NT>(rb, nB, tid);
            }
            T af[TM], bf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) af[t] = tile_frag<T, BM, BK, A_KC>(sA, wm0 + 16 * t + fr, 4 * s + fk);
#pragma unroll
            for (int u = 0; u < TN; ++u) bf[u] = tile_frag<T, BN, BK, B_KC>(sB, wn0 + 16 * u + fr, 4 * s + fk);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) acc[t][u] = Mma<T>::mma(bf[u], af[t], acc[t][u]);
        }

        if (!EARLY && more) {
            T* nA = sA0 + (cur ^ 1) * GA::elems;
            T* nB = sB0 + (cur ^ 1) * GB::elems;
            tile_sstore<T, BM, BK, A_KC, NT>(ra, nA, tid);
            tile_sstore<T, BN, BK, B_KC, NT>(rb, nB, tid);
        }
        __syncthreads();
    }

    // epilogue. lane owns C[i = fr][j = drow(lane, r)] of each 16x16 tile.
    if (g.slab) {
        T* out = g.slab + (int64_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int64_t i = m0 + wm0 + 16 * t + fr;
                    int64_t j = n0 + wn0 + 16 * u + Mma<T>::drow(lane, r);
                    if (i < g.M && j < g.N) out[i + j * g.M] = acc[t][u][r];
                }
    } else if (g.beta != T(0)) {
        // C is read for one row of 16 x 16 tiles at a time, ALL of those loads before the first use (clamped indices keep them unconditional):
        // written as `v += beta * C[..]` inside the store loop, every element was its own load -> s_waitcnt vmcnt(0) -> store round trip,
        // 64 in a row per thread (15-20 us of every accumulating launch, e.g. the rank-32 updates of the LU)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            T cv[TN][4];
            const int64_t i = m0 + wm0 + 16 * t + fr;
            const int64_t ic = i < g.M ? i : g.M - 1;
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t j = n0 + wn0 + 16 * u + Mma<T>::drow(lane, r);
                    cv[u][r] = g.C[ic + (j < g.N ? j : g.N - 1) * g.ldc];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t j = n0 + wn0 + 16 * u + Mma<T>::drow(lane, r);
                    if (i < g.M && j < g.N && !(g.tri && i > j))     // tri: LAPACK uplo contract, strictly lower part untouched
                        g.C[i + j * g.ldc] = g.alpha * acc[t][u][r] + g.beta * cv[u][r];
                }
        }
    } else {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int64_t i = m0 + wm0 + 16 * t + fr;
                    int64_t j = n0 + wn0 + 16 * u + Mma<T>::drow(lane, r);
                    if (i < g.M && j < g.N && !(g.tri && i > j)) g.C[i + j * g.ldc] = g.alpha * acc[t][u][r];
                }
    }
}

// sums split-K slabs in a fixed order: C = alpha * sum_z slab[z] + beta * C.  Four lanes share an output entry (lane q takes the slices
// z = q mod 4, eight loads in flight each, then a fixed two-step butterfly): with one thread per entry walking all slices the reduction of
// the 170 slabs of a tall-skinny Gram matrix took half as long as the product itself.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int64_t M, int64_t N, int nz, const T* __restrict__ slab, T alpha, T beta,
                                                            T* __restrict__ C, int64_t ldc, int tri, int tile) {
    const int64_t total = M * N;
    const int q = threadIdx.x & 3;
    for (int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; idx < ((total + 63) / 64) * 64; idx += ((int64_t)gridDim.x * blockDim.x) >> 2) {
        const int64_t e = idx < total ? idx : total - 1;             // (whole wavefronts stay in the loop for the shuffles)
        T s = 0;
        int z = q;
        for (; z + 28 < nz; z += 32) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(z + 4 * u) * total + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < nz; z += 4) s += slab[(int64_t)z * total + e];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        const int64_t i = e % M, j = e / M;
        if (q != 0 || idx >= total || (tri && i > j)) continue;      // LAPACK uplo contract: the strictly lower triangle is never written
        T v = alpha * s;
        if (beta != T(0)) v += beta * C[i + j * ldc];
        C[i + j * ldc] = v;
    }
}

// ---- C (m x n, both <= 64) = alpha * A^T B + beta * C for TALL operands (k rows from 8192 up): the Gram matrices and cross products of
// narrow panels -- ABRIK's 200000 x 32 Krylov blocks, CQRRT / Cholesky-QR on a few dozen columns, larft's V^T V.  The tiled kernel above
// spends a 128 x 128 (or 256 x 32) output tile on them, 6-16 x the MFMA work and a split-K slab per 128 x 128 tile: 138 us for the
// 32 x 32 Gram matrix of a 200000 x 32 block, whose 51 MB stream in 10 us (profiles/round5_c5_abrik_timeline.txt).  Here a workgroup
// streams 64 KiB slabs of rows -- every column of A (and of B, unless B IS A) with its rows contiguous: 512-byte wavefront loads -- into an
// LDS image [column][rows + 2] (the padding makes the fragment read of 16 columns x 2 rows hit 32 distinct bank pairs), each of its four
// wavefronts multiplies a quarter of the slab's rows on the matrix core (v_mfma_*_16x16x4: A-operand lane (i, kk) = column i of A at row kk,
// B-operand lane (kk, j) = column j of B) into 16 x 16 accumulator tiles that stay in registers across the workgroup's slabs, the four
// wavefronts' tiles are added in wave order, and the workgroup's m x n partial goes to a slab that splitk_reduce_kernel sums in fixed
// order: bitwise reproducible.  B == A shares the fragments; tri (a Gram matrix's upper triangle): only tiles touching it are multiplied.
template <typename T, int NTA, int NTB, bool SAME>
__global__ __launch_bounds__(256) void gemm_tn_skinny_kernel(int m, int n, int64_t k, const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                             int64_t ldb, T* __restrict__ slab, int tri) {
    using M = Mma<T>;
    using acc_t = typename M::acc_t;
    constexpr int WA = 16 * NTA, WB = SAME ? 0 : 16 * NTB, W = WA + WB;
    constexpr int NTJ = SAME ? NTA : NTB;
    constexpr int KS = (W <= 32) ? 256 : (W <= 64) ? 128 : 64;      // rows per slab: 64 KiB of fp64 per slab whatever the width
    constexpr int S = KS + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
    T* sm = reinterpret_cast<T*>(tn_smem);                           // [W][S]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    acc_t acc[NTA][NTJ];
#pragma unroll
    for (int ti = 0; ti < NTA; ++ti)
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) acc[ti][tj] = acc_t{0, 0, 0, 0};
    const int64_t nslab = (k + KS - 1) / KS;
    // a slab travels through registers: ALL of a thread's NE loads of slab s + 1 are in flight while slab s is multiplied out of LDS (eight
    // at a time, each batch behind the previous one's LDS stores, a slab cost four HBM round trips)
    constexpr int NE = W * KS / 256;
    static_assert(W * KS % 256 == 0, "whole rounds of the workgroup");
    T regs[NE];
    auto fetch = [&](int64_t sl) {
        const int64_t r0 = sl * KS;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + 256 * u;
            const int cc = e / KS, r = e - cc * KS;
            const int64_t row = r0 + r;
            // (clamped address + select: the loads stay unconditional and go out together)
            const bool inA = cc < WA;
            const int col = inA ? (cc < m ? cc : m - 1) : ((cc - WA) < n ? (cc - WA) : n - 1);
            const int64_t rw = row < k ? row : k - 1;
            const T v = inA ? A[rw + (int64_t)col * lda] : B[rw + (int64_t)col * ldb];
            regs[u] = (row < k && (inA ? cc < m : (cc - WA) < n)) ? v : T(0);
        }
    };
    if ((int64_t)blockIdx.x < nslab) fetch(blockIdx.x);
    for (int64_t sl = blockIdx.x; sl < nslab; sl += gridDim.x) {
        __syncthreads();                                             // everybody has left the previous slab
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + 256 * u;
            const int cc = e / KS, r = e - cc * KS;
            sm[cc * S + r] = regs[u];
        }
        __syncthreads();
        if (sl + gridDim.x < nslab) fetch(sl + gridDim.x);
        const T* sa = sm + fr * S + wid * (KS / 4) + fk;
#pragma unroll 4
        for (int kk = 0; kk < KS / 4; kk += 4) {
            T fa[NTA], fb[NTJ];
#pragma unroll
            for (int ti = 0; ti < NTA; ++ti) fa[ti] = sa[16 * ti * S + kk];
#pragma unroll
            for (int tj = 0; tj < NTJ; ++tj) fb[tj] = SAME ? fa[tj] : sa[(WA + 16 * tj) * S + kk];
#pragma unroll
            for (int ti = 0; ti < NTA; ++ti)
#pragma unroll
                for (int tj = 0; tj < NTJ; ++tj)
                    if (!(SAME && tri) || ti <= tj) acc[ti][tj] = M::mma(fa[ti], fb[tj], acc[ti][tj]);      // (tri: wave-uniform)
        }
    }
    // the four wavefronts' tiles, added in wave order through LDS; then the workgroup's partial (m x n, column-major) to its slab
    __syncthreads();
    T* red = sm;                                                     // [NTA * 16][NTJ * 16], row index fastest
    constexpr int RM = 16 * NTA;
    for (int w = 0; w < 4; ++w) {
        if (wid == w) {
#pragma unroll
            for (int ti = 0; ti < NTA; ++ti)
#pragma unroll
                for (int tj = 0; tj < NTJ; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ti + M::drow(lane, r), j = 16 * tj + fr;
                        const T v = acc[ti][tj][r];
                        red[i + j * RM] = (w == 0) ? v : red[i + j * RM] + v;
                    }
        }
        __syncthreads();
    }
    T* out = slab + (int64_t)blockIdx.x * m * n;
    for (int e = tid; e < m * n; e += 256) {
        const int i = e % m, j = e / m;
        out[e] = red[i + j * RM];
    }
}

template <typename T>
__global__ void scale_kernel(int64_t M, int64_t N, T beta, T* __restrict__ C, int64_t ldc) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = M * N;
    for (; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = idx % M, j = idx / M;
        C[i + j * ldc] = (beta == T(0)) ? T(0) : beta * C[i + j * ldc];
    }
}


// ---- small products (the k x k matrices of the SVD / Cholesky-QR tails: 256^3 and the like).  The tiled kernel above gives such a product
// 4 workgroups (128 x 128 tiles, no split: the K loop is 16 tiles long) = 40 us at 256^3 on 4 of 256 CUs.  Here a workgroup of four waves
// owns a 32 x 32 block of C (one 16 x 16 MFMA tile per wave), operands come straight from L2 with element strides (any transposition), eight
// k-steps of loads in flight per lane: 64 workgroups, ~6 us at 256^3.  Operands are swapped (B as the MFMA A operand) so that a lane's
// results run along the rows of column-major C.
template <typename T>
__global__ __launch_bounds__(256) void gemm_small_kernel(int M, int N, int K, T alpha, const T* __restrict__ A, int64_t sai, int64_t sak, const T* __restrict__ B,
                                                         int64_t sbk, int64_t sbj, T beta, T* __restrict__ C, int64_t ldc) {
    typedef double d4s_t __attribute__((ext_vector_type(4)));
    typedef float f4s_t __attribute__((ext_vector_type(4)));
    using acc_t = typename std::conditional<sizeof(T) == 8, d4s_t, f4s_t>::type;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int i0 = blockIdx.x * 32 + (wid & 1) * 16, j0 = blockIdx.y * 32 + (wid >> 1) * 16;
    if (i0 >= M || j0 >= N) return;
    const int ia = (i0 + fr < M) ? i0 + fr : M - 1, jb = (j0 + fr < N) ? j0 + fr : N - 1;      // clamped: every load is unconditional
    const T* pa = A + (int64_t)ia * sai;
    const T* pb = B + (int64_t)jb * sbj;
    acc_t acc = {0, 0, 0, 0};
    const int K8 = (K / 32) * 32;
    for (int k0 = 0; k0 < K8; k0 += 32) {
        T av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k0 + 4 * u + fk;
            av[u] = pa[(int64_t)kk * sak];
            bv[u] = pb[(int64_t)kk * sbk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (sizeof(T) == 8) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[u], av[u], acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[u], av[u], acc, 0, 0, 0);
        }
    }
    for (int k0 = K8; k0 < K; k0 += 4) {
        const int kk = k0 + fk;
        const int kc = (kk < K) ? kk : K - 1;
        T a = pa[(int64_t)kc * sak], b = pb[(int64_t)kc * sbk];
        if (kk >= K) { a = T(0); b = T(0); }
        if constexpr (sizeof(T) == 8) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc, 0, 0, 0);
    }
    // swapped operands: D[m = column index j][n = row index i];  f64: register r <-> m = 4 r + fk;  f32: m = 4 fk + r
    const int i = i0 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + ((sizeof(T) == 8) ? (4 * r + fk) : (4 * fk + r));
        if (i < M && j < N) {
            T v = alpha * (T)acc[r];
            if (beta != T(0)) v += beta * C[i + (int64_t)j * ldc];
            C[i + (int64_t)j * ldc] = v;
        }
    }
}

template <typename T, bool A_KC, bool B_KC, int BM, int BN, int BK, int WM, int WN, int MINW, bool EARLY, bool VEC, int DBG = 0>
int launch_one(rlhip_ctx* c, GemmArgs<T>& g, int64_t splitk) {
    using GA = TileGeom<T, BM, BK, A_KC>;
    using GB = TileGeom<T, BN, BK, B_KC>;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr size_t smem = 2 * (size_t)(GA::elems + GB::elems) * sizeof(T);
    const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    dim3 grid((unsigned)tiles, 1, (unsigned)splitk);
    auto kern = gemm_kernel<T, A_KC, B_KC, BM, BN, BK, WM, WN, VEC, MINW, EARLY, DBG>;
    RLHIP_FUNC_LDS(c, kern, smem);
    hipLaunchKernelGGL(kern, grid, dim3(NT), smem, c->stream, g);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

template <typename T, bool A_KC, bool B_KC, int BM, int BN, int BK, int WM, int WN, int MINW, bool EARLY>
int launch_cfg(rlhip_ctx* c, GemmArgs<T>& g, bool vec, int64_t splitk) {
    if (vec) return launch_one<T, A_KC, B_KC, BM, BN, BK, WM, WN, MINW, EARLY, true>(c, g, splitk);
    return launch_one<T, A_KC, B_KC, BM, BN, BK, WM, WN, MINW, EARLY, false>(c, g, splitk);
}

constexpr int NUM_CU = 256;

template <typename T, bool A_KC, bool B_KC>
int gemm_dispatch(rlhip_ctx* c, GemmArgs<T> g, int tri) {
    constexpr int BK = 16;       // fp32 with BK = 32: no LDS conflicts at all, yet 31.4 / 33.2 ms against 30.95 ms per apply product (measured twice, two tile shapes)
    const int64_t M = g.M, N = g.N, K = g.K;
    constexpr int V = 16 / (int)sizeof(T);
    // 16-byte vector loads need aligned bases, leading dimensions multiple of V and (for MC tiles) no
    // ragged start; tile origins are multiples of 16 already.
    bool vec = ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0) && (g.lda % V == 0) && (g.ldb % V == 0);

    // tile shape by N (the narrow dimension on this path), then split-K to fill 256 CUs
    constexpr int variant = 0;         // (bit 1: one 128 x 256 workgroup per CU for N > 128; bit 0: early LDS staging -- both measured slower)
    int cfg;
    int64_t bm, bn;
    // 128 x 128 tiles with two workgroups per CU beat one 128 x 256 workgroup per CU on every driver measured (the second workgroup's
    // MFMAs cover the first one's barrier + fragment-read bubble at each K tile): BQRRP 32768^2 fp32 1397 -> 1366 ms, 65536^2 fp32
    // 5389 -> 5299 ms, 16384^2 fp64 588 -> 572 ms; RSVD and CQRRPT (stream-K kernel for their big products) unchanged.
    if (N > 128 && !tri && !(variant & 2)) { cfg = 5; bm = 128; bn = 128; }
    else if (N > 128 && !tri) { cfg = 0; bm = 128; bn = 256; }
    else if (N > 64 || tri) { cfg = 1; bm = 128; bn = 128; }
    else if (N > 32) { cfg = 2; bm = 256; bn = 64; }
    else if (N > 16) { cfg = 3; bm = 256; bn = 32; }
    else { cfg = 4; bm = 256; bn = 16; }

    int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if (tri) { int64_t tn = (N + bn - 1) / bn; tiles = tn * (tn + 1) / 2; }
    int64_t ktiles = (K + BK - 1) / BK;
    // Split-K choice by a tiny time model: workgroup rounds over 256 CUs (one resident workgroup per
    // CU at these LDS sizes) x per-slice reduction length, plus the slab write+read traffic.
    int64_t splitk = 1;
    {
        const int64_t slots = NUM_CU * ((cfg == 0) ? 1 : 2);  // co-resident workgroups on the chip
        const double flops_cu = 78.6e12 / slots * (sizeof(T) == 4 ? 2.0 : 1.0);
        const double t_tile_k = 2.0 * bm * bn * BK / flops_cu;  // seconds per K-tile per workgroup
        const double slab_bw = 4.0e12;
        // slices of >= 32 K-tiles, except for the Gram matrices of tall-skinny factors (k x k outputs: <= 8 tiles, 1500+ K-tiles): there 48
        // slices x 3 tiles left 112 of 256 CUs without a workgroup (91 us for the 25000 x 256 Gram matrix of a shard), so slices go down to 8
        int64_t maxs = ktiles / ((tiles <= 8) ? 8 : 32);
        if (maxs > 256) maxs = 256;
        double best = 1e300;
        for (int64_t s = 1; s <= (maxs < 1 ? 1 : maxs); ++s) {
            int64_t kc = (ktiles + s - 1) / s;
            int64_t se = (ktiles + kc - 1) / kc;  // effective slices
            int64_t rounds = (tiles * se + slots - 1) / slots;
            double t = rounds * (kc * t_tile_k + 2e-6);
            if (se > 1) t += 2.0 * se * (double)M * N * sizeof(T) / slab_bw + 3e-6;
            if ((double)se * M * N * sizeof(T) > 8e9) continue;
            if (t < best * 0.97) { best = t; splitk = se; }
        }
    }
    int64_t kchunk = ((ktiles + splitk - 1) / splitk) * BK;
    splitk = (K + kchunk - 1) / kchunk;
    if (splitk < 1) splitk = 1;
    g.kchunk = kchunk;
    g.tri = tri;

    size_t mark = rlhip_ws_mark(c);
    T alpha = g.alpha, beta = g.beta;
    if (splitk > 1) {
        g.slab = ws_alloc<T>(c, (size_t)splitk * M * N);
        if (!g.slab) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    } else {
        g.slab = nullptr;
    }

    int rc = 0;
    switch (cfg) {
        case 0:
            if (variant & 4) rc = launch_one<T, A_KC, B_KC, 128, 256, BK, 64, 64, 1, false, true, 1>(c, g, splitk);
            else if (variant & 1) rc = launch_cfg<T, A_KC, B_KC, 128, 256, BK, 64, 64, 1, true>(c, g, vec, splitk);
            else rc = launch_cfg<T, A_KC, B_KC, 128, 256, BK, 64, 64, 1, false>(c, g, vec, splitk);
            break;
        case 5:
            if (variant & 1) rc = launch_cfg<T, A_KC, B_KC, 128, 128, BK, 64, 64, 2, true>(c, g, vec, splitk);
            else rc = launch_cfg<T, A_KC, B_KC, 128, 128, BK, 64, 64, 2, false>(c, g, vec, splitk);
            break;
        case 1: rc = launch_cfg<T, A_KC, B_KC, 128, 128, BK, 64, 64, 2, false>(c, g, vec, splitk); break;
        case 2: rc = launch_cfg<T, A_KC, B_KC, 256, 64, BK, 64, 64, 2, false>(c, g, vec, splitk); break;
        case 3: rc = launch_cfg<T, A_KC, B_KC, 256, 32, BK, 64, 32, 2, false>(c, g, vec, splitk); break;
        default: rc = launch_cfg<T, A_KC, B_KC, 256, 16, BK, 64, 16, 2, false>(c, g, vec, splitk); break;
    }
    if (rc) { rlhip_ws_release(c, mark); return rc; }

    if (splitk > 1) {
        int64_t total = M * N;
        int blocks = (int)((4 * total + 255) / 256);                 // four lanes per entry
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, c->stream, M, N, (int)splitk,
                           g.slab, alpha, beta, g.C, g.ldc, tri, (int)bm);
        RLHIP_LAUNCH_CHECK();
    }
    rlhip_ws_release(c, mark);
    return 0;
}

}  // namespace

namespace rlhip {

template <typename T>
int gemm_streamk(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda, const T* B,
                 int64_t ldb, T beta, T* C, int64_t ldc, double* ssqA_dev, int tri);

template <typename T>
static int try_streamk(rlhip_ctx* c, int ta, int tb, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda, const T* B, int64_t ldb,
                       T beta, T* C, int64_t ldc, double* ssq, int tri) {
    return gemm_streamk<T>(c, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, ssq, tri);
}

// returns 1 if the product was done here (gemm_tn_skinny_kernel), 0 if the caller should carry on, < 0 on error
template <typename T>
static int gemm_tn_skinny(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda, const T* B, int64_t ldb, T beta, T* C,
                          int64_t ldc, int tri) {
    if (m > 64 || n > 64 || k < 8192) return 0;
    const bool same = (A == B) && lda == ldb && m == n;
    if (tri && !same) return 0;
    const int nta = m <= 32 ? 2 : 4, ntb = n <= 32 ? 2 : 4;
    const int W = same ? 16 * nta : 16 * (nta + ntb);
    const int KS = (W <= 32) ? 256 : (W <= 64) ? 128 : 64;
    const size_t lds = (size_t)W * (KS + 2) * sizeof(T) > (size_t)64 * 64 * sizeof(T) ? (size_t)W * (KS + 2) * sizeof(T) : (size_t)64 * 64 * sizeof(T);
    const int64_t nslab = (k + KS - 1) / KS;
    // two workgroups per CU in flight, and every workgroup the same number of slabs (782 slabs on 512 workgroups: half of them walk two,
    // half one -- 391 workgroups walk two each); the partials of G workgroups are what the reduction reads: no more of them than needed
    int64_t G = 2 * (int64_t)c->num_cu;
    if (G > nslab) G = nslab;
    { const int64_t per = (nslab + G - 1) / G; G = (nslab + per - 1) / per; }
    size_t mark = rlhip_ws_mark(c);
    T* slab = ws_alloc<T>(c, (size_t)G * m * n);
    if (!slab) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    hipError_t le = hipSuccess;
#define RLHIP_TN_LAUNCH(NA, NB_, SM)                                                                                                  \
    do {                                                                                                                              \
        RLHIP_FUNC_LDS(c, (gemm_tn_skinny_kernel<T, NA, NB_, SM>), 150 * 1024);                                                       \
        hipLaunchKernelGGL((gemm_tn_skinny_kernel<T, NA, NB_, SM>), dim3((unsigned)G), dim3(256), lds, c->stream, (int)m, (int)n, k, A, lda, B, ldb, slab, tri); \
        le = hipGetLastError();                                                                                                       \
    } while (0)
    if (same) { if (nta == 2) RLHIP_TN_LAUNCH(2, 2, true); else RLHIP_TN_LAUNCH(4, 4, true); }
    else if (nta == 2 && ntb == 2) RLHIP_TN_LAUNCH(2, 2, false);
    else if (nta == 4 && ntb == 2) RLHIP_TN_LAUNCH(4, 2, false);
    else if (nta == 2 && ntb == 4) RLHIP_TN_LAUNCH(2, 4, false);
    else RLHIP_TN_LAUNCH(4, 4, false);
#undef RLHIP_TN_LAUNCH
    if (le == hipSuccess) {
        const int64_t total = m * n;
        int blocks = (int)((4 * total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, c->stream, m, n, (int)G, (const T*)slab, alpha, beta, C, ldc, tri, 0);
        le = hipGetLastError();
    }
    rlhip_ws_release(c, mark);
    if (le != hipSuccess) return RLHIP_ERR_HIP(le);
    return 1;
}

template <typename T>
int gemm_impl(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
              int64_t lda, const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri, double* ssqA_dev, int* ssq_done) {
    if (ssq_done) *ssq_done = 0;
    if (m < 0) return -3;
    if (n < 0) return -4;
    if (k < 0) return -5;
    if (m == 0 || n == 0) return 0;
    if (k == 0 || alpha == T(0)) {
        if (beta == T(1)) return 0;
        int64_t total = m * n;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(scale_kernel<T>, dim3(blocks), dim3(256), 0, c->stream, m, n, beta, C, ldc);
        RLHIP_LAUNCH_CHECK();
        return 0;
    }
    int64_t arows = transA ? k : m, brows = transB ? n : k;
    if (lda < (arows > 1 ? arows : 1)) return -8;
    if (ldb < (brows > 1 ? brows : 1)) return -10;
    if (ldc < (m > 1 ? m : 1)) return -13;
    if (transA && !transB && !ssqA_dev) {        // narrow tall products: Gram matrices / cross products of panels of <= 64 columns
        const int rc = gemm_tn_skinny<T>(c, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    // contraction lengths that are not a multiple of the k-step (row shards of 200000/8 = 25000 rows): the multiple-of-16
    // part goes down the persistent path, the < 16 leftover is one accumulate pass of the generic kernel
    constexpr int64_t SKK = (sizeof(T) == 8) ? 16 : 32;      // K-tile of the persistent kernel (128 bytes per row)
    if (!transB && k % SKK != 0 && k >= 1024 && n % 256 == 0 && !ssqA_dev && (tri ? (m == n) : (m >= 128))) {
        const int64_t k_main = (k / SKK) * SKK;
        int rc = gemm_impl<T>(c, transA, transB, m, n, k_main, alpha, A, lda, B, ldb, beta, C, ldc, tri, nullptr, nullptr);
        if (rc) return rc;
        const T* A2 = transA ? (A + k_main) : (A + k_main * lda);
        return gemm_impl<T>(c, transA, transB, m, n, k - k_main, alpha, A2, lda, B + k_main, ldb, T(1), C, ldc, tri, nullptr, nullptr);
    }
    // fp32, long contractions (BQRRP's W = V^T C over up to 63488 rows): the persistent kernel keeps one fp32 fma chain per output through
    // its whole K, which is only accurate up to ~16k products (gemm_sk.hip) -- so the contraction is cut into chunks of 16384 that
    // accumulate into C (beta = 1 from the second chunk on): the rounding behaviour of a 4-way split-K at the persistent kernel's rate.
    if (sizeof(T) == 4 && !tri && !transB && k > 16384 && k % SKK == 0 && m >= 128 && n % 256 == 0 && !ssqA_dev) {
        {
            const int64_t KC = 16384;
            for (int64_t k0 = 0; k0 < k; k0 += KC) {
                const int64_t kc = (k - k0 < KC) ? (k - k0) : KC;
                const T* Ac = transA ? (A + k0) : (A + k0 * lda);
                int rc = gemm_impl<T>(c, transA, transB, m, n, kc, alpha, Ac, lda, B + k0, ldb, k0 == 0 ? beta : T(1), C, ldc, 0, nullptr, nullptr);
                if (rc) return rc;
            }
            return 0;
        }
    }
    // fp32 Gram matrices over more than 16384 rows (BQRRP's Cholesky-QR panels: 49152 x 2048 at block iteration 8): the same accumulating
    // chunks of 16384 as the rectangular products above, each on the persistent kernel's triangular tile map -- declined by that kernel as ONE
    // contraction (its single fp32 chain per entry), the whole Gram matrix went to the tiled kernel: 3.6 ms for 1.5 ms of MFMA work
    if (sizeof(T) == 4 && tri && !transB && m == n && n % 256 == 0 && k > 16384 && k % SKK == 0) {
        const int64_t KC = 16384;
        for (int64_t k0 = 0; k0 < k; k0 += KC) {
            const int64_t kc = (k - k0 < KC) ? (k - k0) : KC;
            const T* Ac = transA ? (A + k0) : (A + k0 * lda);
            const T* Bc = B + k0;                                  // (op(B) = B is k x n: a row offset)
            int rc = gemm_impl<T>(c, transA, transB, m, n, kc, alpha, Ac, lda, Bc, ldb, k0 == 0 ? beta : T(1), C, ldc, 1, nullptr, nullptr);
            if (rc) return rc;
        }
        return 0;
    }
    if (tri && !transB && m == n && n % 256 == 0 && k % SKK == 0) {
        int rc = try_streamk<T>(c, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, 1);
        if (rc < 0) return rc;
        if (rc == 1) return 0;
    }
    if (!tri && !transB && m >= 128 && n % 256 == 0 && k % SKK == 0) {
        // big data passes: persistent stream-K kernel.  A partial last tile row (m % 128 rows) rides along in the same launch when it is made
        // of whole 16-byte pieces (200000 = 1562 x 128 + 64, a shard of 25000 = 195 x 128 + 40); otherwise the multiple-of-128 row block
        // goes down the persistent path and the generic kernel takes the rest
        constexpr int64_t EPP_ = 16 / (int64_t)sizeof(T);
        const int64_t m_main = ((m % 128) % EPP_ == 0) ? m : (m / 128) * 128;
        int rc = try_streamk<T>(c, transA, transB, m_main, n, k, alpha, A, lda, B, ldb, beta, C, ldc, ssqA_dev, 0);
        if (rc < 0) return rc;
        if (rc == 1) {
            if (ssqA_dev && ssq_done) *ssq_done = (m_main == m) ? 2 : 1;   // 1: covers op(A)'s first (m / 128) * 128 rows, the caller adds the peeled block; 2: all of it
            if (m_main == m) return 0;
            const T* A2 = transA ? (A + m_main * lda) : (A + m_main);
            return gemm_impl<T>(c, transA, transB, m - m_main, n, k, alpha, A2, lda, B, ldb, beta, C + m_main, ldc, 0,
                                nullptr, nullptr);
        }
    }
    if (!tri && m <= 512 && n <= 512 && k <= 2048 && m * n >= 1024 && ((m + 127) / 128) * ((n + 127) / 128) <= 16) {
        {
            // op(A)(i, kk): NoTrans A[i + kk lda], Trans A[kk + i lda];  op(B)(kk, j): NoTrans B[kk + j ldb], Trans B[j + kk ldb]
            hipLaunchKernelGGL(gemm_small_kernel<T>, dim3((unsigned)((m + 31) / 32), (unsigned)((n + 31) / 32)), dim3(256), 0, c->stream, (int)m, (int)n, (int)k, alpha, A,
                               transA ? lda : (int64_t)1, transA ? (int64_t)1 : lda, B, transB ? ldb : (int64_t)1, transB ? (int64_t)1 : ldb, beta, C, ldc);
            RLHIP_LAUNCH_CHECK();
            return 0;
        }
    }
    GemmArgs<T> g;
    g.M = m; g.N = n; g.K = k; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta; g.kchunk = 0; g.slab = nullptr; g.tri = 0;
    // op(A) element (i,kk):  NoTrans -> A[i + kk*lda] (MC);  Trans -> A[kk + i*lda] (KC)
    // op(B) element (kk,j):  NoTrans -> B[kk + j*ldb] (KC);  Trans -> B[j + kk*ldb] (MC)
    if (!transA && !transB) return gemm_dispatch<T, false, true>(c, g, tri);
    if (transA && !transB) return gemm_dispatch<T, true, true>(c, g, tri);
    if (!transA && transB) return gemm_dispatch<T, false, false>(c, g, tri);
    return gemm_dispatch<T, true, false>(c, g, tri);
}

template <typename T>
int gemm(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
         int64_t lda, const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {
    return gemm_impl<T>(c, transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0, nullptr, nullptr);
}

// syrk: only Trans (C = alpha*A^T*A + beta*C, A is k x n) and NoTrans (C = alpha*A*A^T + beta*C, A is n x k).
// Only tiles touching the upper triangle are computed and only elements i <= j are written: the strictly lower
// triangle of C is left untouched, as LAPACK promises (CQRRPT relies on it: R's lower part must stay zero for
// the trmm at rl_cqrrpt.hh:345).
template <typename T>
int syrk(rlhip_ctx* c, int uplo, int trans, int64_t n, int64_t k, T alpha, const T* A, int64_t lda, T beta,
         T* C, int64_t ldc) {
    if (uplo != Upper) return -2;  // the path only ever asks for Upper (rl_orth.hh:78, rl_cqrrpt.hh:310)
    if (trans) return gemm_impl<T>(c, 1, 0, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, 1, nullptr, nullptr);
    return gemm_impl<T>(c, 0, 1, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, 1, nullptr, nullptr);
}

template int gemm<double>(rlhip_ctx*, int, int, int64_t, int64_t, int64_t, double, const double*, int64_t,
                          const double*, int64_t, double, double*, int64_t);
template int gemm<float>(rlhip_ctx*, int, int, int64_t, int64_t, int64_t, float, const float*, int64_t,
                         const float*, int64_t, float, float*, int64_t);
template int syrk<double>(rlhip_ctx*, int, int, int64_t, int64_t, double, const double*, int64_t, double,
                          double*, int64_t);
template int syrk<float>(rlhip_ctx*, int, int, int64_t, int64_t, float, const float*, int64_t, float, float*,
                         int64_t);

}  // namespace rlhip

namespace rlhip {
template int gemm_impl<double>(rlhip_ctx*, int, int, int64_t, int64_t, int64_t, double, const double*, int64_t,
                               const double*, int64_t, double, double*, int64_t, int, double*, int*);
template int gemm_impl<float>(rlhip_ctx*, int, int, int64_t, int64_t, int64_t, float, const float*, int64_t,
                              const float*, int64_t, float, float*, int64_t, int, double*, int*);
}  // namespace rlhip
