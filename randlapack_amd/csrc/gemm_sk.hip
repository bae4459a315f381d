// Persistent "stream-K" fp64 MFMA GEMM for the two data passes of the sketch-and-factor path:
//     Y   = A  * Omega   (m x k  <-  m x n . n x k,  NN)       RandLAPACK/comps/rl_rf.hh:123, rl_rs.hh:153
//     B^T = A^T * Q      (n x k  <-  n x m . m x k,  TN)       RandLAPACK/comps/rl_qb.hh:218, rl_rs.hh:142,165
// i.e. C(M x N) = alpha * op(A) * B + beta * C with N a multiple of 256, K a multiple of 16, op(B) = B.
//
// Why a second GEMM kernel (measured on MI355X):
//   * the register-staged kernel in gemm.hip loses ~7 % to wave quantisation at 1563 tiles / 256 CUs and ~5 %
//     to exposed HBM latency, and its split-K needs a shape-dependent heuristic;
//   * here ONE workgroup per CU owns an equal, contiguous share of the (tile, k-tile) iteration space
//     ("stream-K"), so every CU finishes at the same time for ANY shape; tiles cut by a share boundary are
//     written as partial slabs and summed by a fix-up kernel in fixed k order (bitwise reproducible);
//   * operands go HBM -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write pass) into a
//     3-deep ring of 48 KiB stages (144 of the CU's 160 KiB): two K-tiles (~7 us) of loads are always in
//     flight behind a counted s_waitcnt vmcnt(6) and a single raw s_barrier per K-tile;
//   * fragments are fetched with ds_read_b128 (two MFMA operands per read, 4 LDS cycles per 16 bytes instead of
//     the 8-16 of the ds_read2_b64 pairs hipcc builds from scalar reads -- PMC showed 1e9 bank-conflict cycles
//     and a 14 % matrix-pipe bubble with those).  To make 16 contiguous bytes useful to ONE lane the
//     reduction index and the row index are re-enumerated (any bijection is legal as long as both operands
//     agree):  MFMA step (sigma, h), lane group fk  <->  kk = 8*sigma + 2*fk + h   (KC images: one 16-byte
//     piece = kk, kk+1);  MC image: fragment x, lane fr  <->  row 32*(x>>1) + 2*fr + (x&1)  (rows i, i+1).
//   * LDS images: the DMA destination is lane-linear by construction, so bank-conflict freedom is obtained
//     by permuting the SOURCE addresses:  KC operand (Omega, Q, A of A^T*Q): row r keeps 16-byte piece c at
//     slot c ^ ((r >> 1) & 7);  MC operand (A of A*Omega): plain (the b128 lane groups already spread).
//   * v_mfma_f64_16x16x4_f64, 64 x 64 accumulator tile per wave (2 x 4 waves -> 128 x 256 block tile).
//   * fp32 twin (BQRRP's compact-WY products, rl_bqrrp.hh:535-547): the SAME byte geometry -- a K-tile is 32 floats = 128 bytes per
//     row, so stages, DMA pieces, swizzle and ring are identical -- with v_mfma_f32_16x16x4_f32 (32-cycle issue, 16 independent
//     accumulators per wave).  A 16-byte piece now holds four consecutive k, i.e. it feeds FOUR MFMA steps:
//     step (sigma, h), lane group fk  <->  kk = 16*sigma + 4*fk + h;  MC image: lane fr reads rows 4*fr .. 4*fr+3 of one k-row, i.e.
//     fragment x, lane fr  <->  row 4*fr + x.  Accumulator layout of the f32 MFMA: lane (fr, fk), register r = C[row(fr)][16u + 4fk + r].
#include "rlhip_internal.h"
#include <cstdlib>
#include <type_traits>
#include <cstdio>

namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BM = 128, BN = 256;
constexpr int STAGE_A = BM * 128;             // 16 KiB: a K-tile is 128 bytes per row (16 doubles / 32 floats)
constexpr int STAGE_B = BN * 128;             // 32 KiB
constexpr int STAGE = STAGE_A + STAGE_B;      // 48 KiB
constexpr int NSTAGE = 3;
constexpr int SLAB_ELEMS = BM * BN;

template <int N> struct HC { static constexpr int value = N; };

template <typename T> struct SkT;
template <> struct SkT<double> {
    static constexpr int BK = 16, EPP = 2, NH = 2;      // k per tile, elements per 16-byte piece, MFMA steps fed by one piece
    typedef d2_t frag_t;                                 // one 16-byte LDS read
    typedef d4_t acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int ccol(int fk, int r) { return fk + 4 * r; }       // column inside a 16-wide tile of register r
};
template <> struct SkT<float> {
    static constexpr int BK = 32, EPP = 4, NH = 4;
    typedef f4_t frag_t;
    typedef f4_t acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int ccol(int fk, int r) { return 4 * fk + r; }
};

template <typename T>
struct SkArgs {
    int64_t M, N, K;          // N multiple of 256, K multiple of BK (caller peels the rest); M a multiple of 128 for the tri map, else any
                              // M >= 128 whose last, partial tile row has a multiple of EPP rows (its DMA pieces re-read the last valid rows)
    const T* A; int64_t lda;
    const T* B; int64_t ldb;
    T* C; int64_t ldc;
    T alpha, beta;
    int64_t tiles_m, tiles_n, ktiles;
    T* slab;                  // 2 * gridDim.x slots of 128 x 256
    double* ssq_part;         // nullptr, or gridDim.x partial sums of squares of op(A) (fused ||A||_F^2)
    int tri;                  // 1: syrk-upper -- only tiles touching i <= j are computed, only i <= j is written
    int64_t ntiles;           // number of active tiles
    int gs;                   // workgroups per lockstep group (see sk_group below); 1 = every workgroup on its own
    int stag;                 // K-tile stagger between the members of a lockstep group (transposed product only; 0 = walk in phase)
    unsigned long long* clk;  // nullptr, or 4 words: workgroup 0's shader-clock and 100 MHz timestamps at entry and exit (RLHIP_SK_CLOCK=1: effective clock)
};

// Lockstep groups.  With one share per workgroup, the workgroups of an XCD sit at unrelated k offsets, so the operand every tile needs
// in full (Omega of Y = A Omega: 41 MB; Q of B^T = A^T Q: 410 MB) streams through the 4 MiB L2 once PER TILE and the L2's fabric side
// carries 2.3x the algorithmic bytes (PMC, profiles/round3_pmc_gemm_sk_nn.json).  Here `gs` consecutive tiles form one unit of the
// (unit, k-tile) iteration space, a GROUP of gs workgroups owns an equal contiguous share of it, and member j of the group walks tile
// gs * unit + j over the same k range at the same time: the members request the same k-rows of the shared operand within
// microseconds of each other and all but one of them hit in L2.  Members of a group must share an L2, i.e. an XCD: workgroups are dealt
// to the XCDs round-robin (blockIdx % 8), so group = XCD + 8 * (slot in the XCD / gs).  (The placement only decides who shares a
// cache: any other dispatch order costs hits, never correctness.)
struct SkWho { int64_t group, member; };
__device__ __host__ __forceinline__ SkWho sk_group(int64_t w, int gs) {
    if (gs <= 1) return SkWho{w, 0};
    const int64_t slot = w / 8;
    return SkWho{(w % 8) + 8 * (slot / gs), slot % gs};
}
__device__ __host__ __forceinline__ int64_t sk_workgroup(int64_t group, int64_t member, int gs) {
    if (gs <= 1) return group;
    return (group % 8) + 8 * (gs * (group / 8) + member);
}

// active tile index -> tile descriptor: first row m0 of the 128-row operand block, first column of the tile's left 128 columns (nA0)
// and of its right 128 columns MINUS 128 (nB0: column jl >= 128 of the tile is global column nB0 + jl), and whether the tile is
// written transposed.
//   Full map: row-major over N (nA0 == nB0 == 256 tn).
//   Tri map (syrk, upper triangle of the n x n Gram matrix, T = n / 128 block rows): block row i needs the 128-blocks i .. T-1 of
//   its row.  They are taken in PAIRS starting at the diagonal block (tiles (i; i + 2p, i + 2p + 1): floor((T - i) / 2) per row);
//   an odd row is left with the single block (i, T-1).  Those leftovers all share the column block T-1, so two of them are computed
//   as ONE tile of the TRANSPOSED product  A_{T-1}^T [A_r1 A_r2]  (= [G(r1, T-1)^T  G(r2, T-1)^T]) and written transposed.
//   n = 1024: 16 + 2 = 18 tiles instead of the 20 of a 256-aligned map (of which four were half below the diagonal).
struct SkTile { int64_t m0, nA0, nB0; int transposed, single; };
template <typename T>
__device__ __forceinline__ SkTile sk_tile(const SkArgs<T>& g, int64_t a) {
    SkTile t;
    t.transposed = 0; t.single = 0;
    if (!g.tri) {
        const int64_t tm = a / g.tiles_n, tn = a - tm * g.tiles_n;
        t.m0 = tm * BM; t.nA0 = t.nB0 = tn * BN;
        return t;
    }
    const int64_t Tb = g.tiles_m;                   // 128-blocks per side
    for (int64_t i = 0; i < Tb; ++i) {
        const int64_t cnt = (Tb - i) / 2;
        if (a < cnt) { t.m0 = i * BM; t.nA0 = t.nB0 = (i + 2 * a) * BM; return t; }
        a -= cnt;
    }
    // leftover halves: odd block rows 1, 3, ..., paired two by two
    const int64_t r1 = 4 * a + 1;
    const int64_t r2 = (r1 + 2 < Tb) ? r1 + 2 : r1;     // an unpaired last one: the right half recomputes the left one and is not written
    t.m0 = (Tb - 1) * BM; t.nA0 = r1 * BM; t.nB0 = r2 * BM - BM; t.transposed = 1; t.single = (r2 == r1);
    return t;
}

// element (il, jl) of a tile -> position in C; false when the tri map must not write it
__device__ __forceinline__ bool sk_cpos(const SkTile& t, int tri, int64_t il, int64_t jl, int64_t& i, int64_t& j) {
    if (t.single && jl >= BM) return false;
    const int64_t col = (jl < BM ? t.nA0 : t.nB0) + jl;
    if (t.transposed) { i = col; j = t.m0 + il; }
    else { i = t.m0 + il; j = col; }
    return !(tri && i > j);
}

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

// O32: the per-lane source offsets of the DMA pieces are unsigned 32-bit BYTE offsets from a wave-uniform pointer (the host takes this
// form whenever 256 rows of either operand span less than 4 GiB): the request is `global_load_lds voff, s[base]` -- one VGPR per piece instead
// of a 64-bit lane pointer built per K-tile.  With 64-bit offsets (round 4) the fp64 kernels ran out of registers IN the K loop: hipcc
// spilled the LDS destination offset and reloaded it between the first and the second DMA request of every K-tile, behind an
// s_waitcnt vmcnt(0) -- i.e. every wave waited one HBM round trip for the piece it had just requested, with the matrix pipe draining.
template <typename T, bool A_KC, bool O32>
__global__ __launch_bounds__(512, 1) void gemm_sk_kernel(SkArgs<T> g) {
    using S = SkT<T>;
    using frag_t = typename S::frag_t;
    using acc_t = typename S::acc_t;
    constexpr int BK = S::BK, EPP = S::EPP, NH = S::NH;
    constexpr bool F64 = (sizeof(T) == 8);
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    using off_t = typename std::conditional<O32, unsigned, int64_t>::type;      // byte offsets
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform, and (UW) known to be: everything derived from it -- LDS destinations, M0, the stage bases -- then lives in SGPRs and the
    // K loop of the fp64 kernels fits its registers.  Measured per instantiation (scripts/exp/ab_gemm_f32.py, ms at the C2 / C4 shapes,
    // round-4 build -> lane-derived wid -> uniform wid):  fp64 NN 29.7 -> 29.7 -> 28.7;  fp64 TN 30.9 -> 30.4 -> 29.1;  fp32 NN 32.8 -> 32.55
    // -> 32.2;  fp32 TN 8.20 -> 8.05 -> 8.66 (same MFMA stream, all-scalar DMA issue; not understood) -- so fp32 TN keeps the lane-derived form.
    constexpr bool UW = !(A_KC && !F64);
    const int wid = UW ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int wm0 = (wid & 1) * 64, wn0 = (wid >> 1) * 64;
    const int fr = lane & 15, fk = lane >> 4;

    if (g.clk && blockIdx.x == 0 && tid == 0) { g.clk[0] = __builtin_readcyclecounter(); g.clk[1] = wall_clock64(); }
    const int64_t KT = g.ktiles;
    const int64_t GS = g.gs, w = blockIdx.x;
    const int64_t units = (g.ntiles + GS - 1) / GS;                   // a unit = GS consecutive tiles walked in lockstep by a group
    const int64_t W = units * KT;
    const int64_t P = (int64_t)gridDim.x / GS;                         // groups
    const SkWho who = sk_group(w, (int)GS);
    const int64_t ws = (who.group * W) / P, we = ((who.group + 1) * W) / P;
    const int64_t first_tile = ws / KT;                                // (first UNIT of the share)

    // ---- per-lane constants of the fragment reads (see header for the index re-enumeration)
    // KC image: 16-byte piece c of row r lives at byte r*128 + ((c ^ ((r>>1)&7)) << 4); with r = 16*x + fr the
    // key is fr>>1; step group sigma uses piece c = 4*sigma + fk
    const int kc_key = (fr >> 1) & 7;
    const int kc_off0 = ((0 + fk) ^ kc_key) << 4, kc_off1 = ((4 + fk) ^ kc_key) << 4;
    // MC image, fp64: byte(i, kk) = kk*1024 + i*8 ; lane reads rows (32*xi + 2*fr, +1) of k-row kk = 8*sigma + 2*fk + h
    //           fp32: byte(i, kk) = kk*512  + i*4 ; lane reads rows 4*fr .. 4*fr+3       of k-row kk = 16*sigma + 4*fk + h
    const int mc_row = F64 ? (wm0 + 2 * fr) * 8 : (wm0 + 4 * fr) * 4;
    const int mc_k = F64 ? 2 * fk * 1024 : 4 * fk * 512;

    // ---- per-lane source offsets of this wave's 6 DMA pieces per K-tile (elements, relative to tile origin)
    // A: chunks {wid, wid+8}; B: chunks {wid, wid+8, wid+16, wid+24}
    off_t a_src[2], b_src[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = wid + 8 * h;
        if constexpr (A_KC) {
            const int r = 8 * c + (lane >> 3), q = lane & 7;
            a_src[h] = (off_t)(((int64_t)r * g.lda + EPP * (q ^ ((r >> 1) & 7))) * (int64_t)sizeof(T));
        } else if constexpr (F64) {
            a_src[h] = (off_t)(((int64_t)c * g.lda + 2 * lane) * (int64_t)sizeof(T));                                  // chunk = one k-row of 128 doubles
        } else {
            a_src[h] = (off_t)(((int64_t)(2 * c + (lane >> 5)) * g.lda + 4 * (lane & 31)) * (int64_t)sizeof(T));       // chunk = two k-rows of 128 floats
        }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int c = wid + 8 * h;
        const int r = 8 * c + (lane >> 3), q = lane & 7;
        b_src[h] = (off_t)(((int64_t)r * g.ldb + EPP * (q ^ ((r >> 1) & 7))) * (int64_t)sizeof(T));
    }
    const int64_t a_step = A_KC ? (int64_t)BK : (int64_t)BK * g.lda;
    // the last tile row may be partial (M % 128 rows): its DMA pieces take the rows past the end from the last valid piece instead -- a row
    // of C depends on the same row of op(A) only, so what those rows hold never reaches a stored entry
    const int64_t m_rag0 = (g.tri || g.M % BM == 0) ? g.M : (g.M / BM) * BM;     // first row of the partial tile row (M: there is none)
    const int m_rem = (int)(g.M - m_rag0);
    off_t a_src_rag[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = wid + 8 * h;
        if constexpr (A_KC) {
            const int r = 8 * c + (lane >> 3), q = lane & 7;
            const int re = (m_rem > 0) ? r % m_rem : 0;                               // (spread over the valid rows: 96 lanes on ONE line would serialise in the texture path)
            a_src_rag[h] = (off_t)(((int64_t)re * g.lda + EPP * (q ^ ((r >> 1) & 7))) * (int64_t)sizeof(T));
        } else if constexpr (F64) {
            const int ro = (m_rem > 1) ? (2 * lane) % m_rem : 0;                      // (m_rem is even here)
            a_src_rag[h] = (off_t)(((int64_t)c * g.lda + ro) * (int64_t)sizeof(T));
        } else {
            const int ro = (m_rem > 3) ? (4 * (lane & 31)) % m_rem : 0;               // (m_rem is a multiple of 4 here)
            a_src_rag[h] = (off_t)(((int64_t)(2 * c + (lane >> 5)) * g.lda + ro) * (int64_t)sizeof(T));
        }
    }
    // which of this thread's two 16-byte pieces of an A stage (pieces tid and tid + 512) hold rows of the partial tile that exist
    bool ssq_ok_rag[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pc = tid + 512 * h;
        if constexpr (A_KC) ssq_ok_rag[h] = (pc >> 3) < m_rem;
        else if constexpr (F64) ssq_ok_rag[h] = 2 * (pc & 63) < m_rem;
        else ssq_ok_rag[h] = 4 * (pc & 31) < m_rem;
    }

    double ssq_acc = 0.0;
    for (int64_t pos = ws; pos < we;) {
        const int64_t unit = pos / KT;
        const int64_t kt0 = pos - unit * KT;
        int64_t nk = KT - kt0;
        if (nk > we - pos) nk = we - pos;
        const int64_t tile = GS * unit + who.member;
        if (tile >= g.ntiles) { pos += nk; continue; }                 // the last unit may be short: this member has nothing there
        const SkTile td = sk_tile(g, tile);
        const int64_t m0 = td.m0, n0 = td.nA0, k0 = kt0 * BK;
        const bool rag = (m0 >= m_rag0);                                // (never for the tri map)
        const off_t as0 = rag ? a_src_rag[0] : a_src[0], as1 = rag ? a_src_rag[1] : a_src[1];
        const int64_t m_lim = g.tri ? ((int64_t)1 << 62) : g.M;

        const T* Ag = A_KC ? (g.A + k0 + m0 * g.lda) : (g.A + m0 + k0 * g.lda);
        const T* Bg = g.B + k0 + td.nA0 * g.ldb;              // columns 0 .. 127 of the tile
        const T* Bg1 = g.B + k0 + td.nB0 * g.ldb;             // columns 128 .. 255 (the same block row for ordinary tiles)

        // Staggered lockstep (transposed product): the members of a group walk the SAME K range, but member j starts `stag * j` K-tiles into it
        // and wraps around.  In phase, the eight members fetch the same 16 rows of eight neighbouring column blocks of A -- addresses a multiple
        // of 128 * lda * sizeof(T) apart, which land on a handful of memory channels (31.2 ms against 30.0 ungrouped at C2); a few tiles apart
        // they hit different channels while the rows of the shared operand Q stay within a ~1 MiB window of L2.  The sum over a segment is taken
        // in the rotated order: fixed per (shape, grid), so results stay reproducible.
        // (32-bit tile counters: there is no scalar 64-bit >=, so with 64-bit ones the wrap-around below -- and every address after it -- was
        // computed per lane in the vector unit, between the MFMAs)
        const int nk_i = (int)nk;
        const int rot = (A_KC && GS > 1 && g.stag > 0) ? (int)(((int64_t)who.member * g.stag) % nk) : 0;
        auto issue = [&](int t, int stage) {   // DMA K-tile t of this segment into ring stage `stage`
            unsigned char* st = smem + stage * STAGE;
            t += rot;
            if (t >= nk_i) t -= nk_i;
            const char* Ap = reinterpret_cast<const char*>(Ag + (int64_t)t * a_step);          // wave-uniform
            const char* Bp = reinterpret_cast<const char*>(Bg + (int64_t)t * BK);
            const char* Bp1 = reinterpret_cast<const char*>(Bg1 + (int64_t)t * BK);
#pragma unroll
            for (int h = 0; h < 2; ++h) glds16(Ap + (h ? as1 : as0), st + (wid + 8 * h) * 1024);
#pragma unroll
            for (int h = 0; h < 4; ++h) glds16((h < 2 ? Bp : Bp1) + b_src[h], st + STAGE_A + (wid + 8 * h) * 1024);
        };

        acc_t acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = acc_t{0, 0, 0, 0};
        // ---- software pipeline (one rendezvous per K-tile, placed in the MIDDLE of the tile):
        //   F0 = fragments of step group 0, F1 = step group 1.  While the MFMAs of F0 run, F1 is fetched; at
        //   the mid-point the wave retires its own DMA pieces of tile t+1 (the only group outstanding) and
        //   meets the other waves: after that barrier tile t+1 is visible to everybody AND everybody has left
        //   tile t-1, so tile t+2 may be DMA'd into t-1's stage and F0 of tile t+1 may be fetched while the
        //   MFMAs of F1 run.  The matrix pipe therefore never waits for a tile boundary.
        frag_t fa0[4], fb0[4], fa1[4], fb1[4];
        const bool do_ssq = (g.ssq_part != nullptr) && (td.nA0 == 0);   // every A element is staged once by tile_n == 0
        auto fetch = [&](const unsigned char* sA, int sg, frag_t (&a)[4], frag_t (&b)[4]) {
            const unsigned char* sB = sA + STAGE_A;
            const int kc = sg ? kc_off1 : kc_off0;
#pragma unroll
            for (int x = 0; x < 4; ++x) b[x] = *reinterpret_cast<const frag_t*>(sB + (wn0 + 16 * x + fr) * 128 + kc);
            if constexpr (A_KC) {
#pragma unroll
                for (int x = 0; x < 4; ++x) a[x] = *reinterpret_cast<const frag_t*>(sA + (wm0 + 16 * x + fr) * 128 + kc);
            } else if constexpr (F64) {
                // MC image: a[2*h + xi] = rows (32*xi + 2*fr, +1) of k-row 8*sg + 2*fk + h
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int xi = 0; xi < 2; ++xi)
                        a[2 * h + xi] = *reinterpret_cast<const frag_t*>(sA + (8 * sg + h) * 1024 + mc_k + mc_row + xi * 256);
            } else {
                // MC image: a[h] = rows 4*fr .. 4*fr+3 of k-row 16*sg + 4*fk + h
#pragma unroll
                for (int h = 0; h < 4; ++h) a[h] = *reinterpret_cast<const frag_t*>(sA + (16 * sg + h) * 512 + mc_k + mc_row);
            }
        };
        auto mma16 = [&](frag_t (&a)[4], frag_t (&b)[4], auto hc) {
            constexpr int h = decltype(hc)::value;     // compile-time element index: a run-time index into a register vector goes through scratch
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    T av;
                    if constexpr (A_KC) av = a[x][h];
                    else if constexpr (F64) av = a[2 * h + (x >> 1)][x & 1];
                    else av = a[h][x];
                    acc[x][u] = S::mma(b[u][h], av, acc[x][u]);
                }
        };

        __builtin_amdgcn_s_barrier();   // nobody still reads the ring (previous segment)
        issue(0, 0);
        if (nk_i > 1) {
            issue(1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();   // tile 0 visible
        fetch(smem, 0, fa0, fb0);
        int st_cur = 0;   // ring stage holding tile t
        for (int t = 0; t < nk_i; ++t) {
            const int st_next = (st_cur == 2) ? 0 : st_cur + 1;
            const int st_prev = (st_cur == 0) ? 2 : st_cur - 1;
            fetch(smem + st_cur * STAGE, 1, fa1, fb1);
            if (do_ssq) {   // the A stage = 16 KiB / 512 threads = two b128 reads per thread (layout-agnostic)
                const frag_t* sa = reinterpret_cast<const frag_t*>(smem + st_cur * STAGE);
                frag_t v0 = sa[tid], v1 = sa[tid + 512];
                if (rag) {
                    if (!ssq_ok_rag[0]) v0 = frag_t{};
                    if (!ssq_ok_rag[1]) v1 = frag_t{};
                }
                if constexpr (F64) {
                    ssq_acc = fma(v0[0], v0[0], fma(v0[1], v0[1], fma(v1[0], v1[0], fma(v1[1], v1[1], ssq_acc))));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ssq_acc = fma((double)v0[e], (double)v0[e], fma((double)v1[e], (double)v1[e], ssq_acc));
                }
            }
            mma16(fa0, fb0, HC<0>{});
            mma16(fa0, fb0, HC<1>{});
            if constexpr (NH == 4) { mma16(fa0, fb0, HC<2>{}); mma16(fa0, fb0, HC<3>{}); }
            if (t + 1 < nk_i) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // unconditional (straight-line code lets hipcc use a counted lgkmcnt for F1 instead of lgkmcnt(0));
            // on the last tile this reads a stale stage and the values are never used
            fetch(smem + st_next * STAGE, 0, fa0, fb0);
            mma16(fa1, fb1, HC<0>{});
            if constexpr (NH == 4) mma16(fa1, fb1, HC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk_i) issue(t + 2, st_prev);   // DMA issue slots hidden behind the MFMAs just queued
            __builtin_amdgcn_sched_barrier(0);
            mma16(fa1, fb1, HC<NH / 2>{});
            if constexpr (NH == 4) mma16(fa1, fb1, HC<3>{});
            st_cur = st_next;
        }

        // ---- epilogue: lane owns C[i = crow(x)][j = 16u + ccol(fk, r)] of its 64 x 64 wave tile
        // C row of fragment x, lane fr (MC images interleave the fragments' rows, see header)
        auto crow = [&](int x) {
            if constexpr (A_KC) return wm0 + 16 * x + fr;
            else if constexpr (F64) return wm0 + 32 * (x >> 1) + 2 * fr + (x & 1);
            else return wm0 + 4 * fr + x;
        };
        const bool whole = (kt0 == 0) && (nk == KT);
        if constexpr (!A_KC && !F64) {
            // fp32 MC image: the lane's four fragments are the four CONSECUTIVE rows 4 fr .. 4 fr + 3 -> one 16-byte access per column
            // (16 lanes cover 256 contiguous bytes of a column of C) instead of four 4-byte accesses 16 bytes apart
            const int64_t i0 = m0 + wm0 + 4 * fr;
            if (whole && !g.tri && g.beta != T(0) && i0 < m_lim && (g.ldc & 3) == 0 && (((uintptr_t)g.C) & 15) == 0) {
                // accumulating tile (BQRRP's C -= V W): the 16-byte pieces of C are read four at a time BEFORE the first is used -- written as
                // `v += beta * C[..]` inside the store loop every piece was its own load -> s_waitcnt vmcnt(0) -> store round trip to HBM, sixteen in
                // a row per thread with the matrix pipe idle behind them (~7 % of a K = 2048 tile)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f4_t cv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) cv[r] = *reinterpret_cast<const f4_t*>(g.C + i0 + (n0 + wn0 + 16 * u + S::ccol(fk, r)) * g.ldc);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<f4_t*>(g.C + i0 + (n0 + wn0 + 16 * u + S::ccol(fk, r)) * g.ldc) =
                            f4_t{acc[0][u][r], acc[1][u][r], acc[2][u][r], acc[3][u][r]} * g.alpha + g.beta * cv[r];
                }
            } else if (whole) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t j = n0 + wn0 + 16 * u + S::ccol(fk, r);
                        f4_t v = f4_t{acc[0][u][r], acc[1][u][r], acc[2][u][r], acc[3][u][r]} * g.alpha;
                        T* dst = g.C + i0 + j * g.ldc;
                        if (i0 >= m_lim) continue;                     // rows of a partial tile that do not exist (4 | M % 128: all four or none)
                        if (g.tri) {                      // (the tri map is only used with A_KC operands; kept for completeness)
#pragma unroll
                            for (int x = 0; x < 4; ++x)
                                if (i0 + x <= j) dst[x] = (g.beta != T(0)) ? v[x] + g.beta * dst[x] : v[x];
                        } else if ((g.ldc & 3) == 0 && (((uintptr_t)g.C) & 15) == 0) {
                            if (g.beta != T(0)) v += g.beta * *reinterpret_cast<const f4_t*>(dst);
                            *reinterpret_cast<f4_t*>(dst) = v;
                        } else {
#pragma unroll
                            for (int x = 0; x < 4; ++x) dst[x] = (g.beta != T(0)) ? v[x] + g.beta * dst[x] : v[x];
                        }
                    }
            } else {
                T* out = g.slab + (2 * w + (unit != first_tile ? 1 : 0)) * (int64_t)SLAB_ELEMS;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<f4_t*>(out + (wm0 + 4 * fr) + (wn0 + 16 * u + S::ccol(fk, r)) * BM) =
                            f4_t{acc[0][u][r], acc[1][u][r], acc[2][u][r], acc[3][u][r]};
            }
        } else
        if (whole && g.beta != T(0)) {
            // accumulating tile: the sixteen entries of a fragment row are read together (entries that are not written read C[0]), then stored
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                T cv[4][4];
                int64_t off[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int64_t i, j;
                        const bool ok = sk_cpos(td, g.tri, crow(x), wn0 + 16 * u + S::ccol(fk, r), i, j) && i < m_lim;
                        off[u][r] = ok ? i + j * g.ldc : (int64_t)-1;
                        cv[u][r] = g.C[ok ? off[u][r] : 0];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (off[u][r] >= 0) g.C[off[u][r]] = g.alpha * acc[x][u][r] + g.beta * cv[u][r];
            }
        } else if (whole) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int64_t i, j;
                        if (!sk_cpos(td, g.tri, crow(x), wn0 + 16 * u + S::ccol(fk, r), i, j) || i >= m_lim) continue;
                        g.C[i + j * g.ldc] = g.alpha * acc[x][u][r];
                    }
        } else {
            T* out = g.slab + (2 * w + (unit != first_tile ? 1 : 0)) * (int64_t)SLAB_ELEMS;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[crow(x) + (wn0 + 16 * u + S::ccol(fk, r)) * BM] = acc[x][u][r];
        }
        pos += nk;
    }
    if (g.clk && blockIdx.x == 0 && tid == 0) { g.clk[2] = __builtin_readcyclecounter(); g.clk[3] = wall_clock64(); }
    if (g.ssq_part) {   // deterministic: fixed lane->element map, fixed reduction tree, one partial per workgroup
        double v = ssq_acc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);
        if (lane == 0) red[wid] = v;
        __syncthreads();
        if (tid == 0) {
            double sacc = 0;
            for (int i = 0; i < 8; ++i) sacc += red[i];
            g.ssq_part[w] = sacc;
        }
    }
}

// Sums the partial slabs of every tile that was cut by a share boundary, in increasing k order.
template <typename T>
__global__ __launch_bounds__(256) void gemm_sk_fixup_kernel(SkArgs<T> g, int64_t P, int by_boundary) {
    const int64_t KT = g.ktiles;
    const int64_t GS = g.gs;
    const int64_t W = ((g.ntiles + GS - 1) / GS) * KT;
    // by_boundary: blockIdx.x = (b - 1) * GS + member enumerates the P - 1 share boundaries -- only a unit with a boundary INSIDE it was cut.
    // (A product with many tiles -- BQRRP's C -= V W: 70656 tiles, 255 of them cut -- used to launch a workgroup per tile and slice, 1.1 M
    // workgroups that returned at once: 1.2 ms per launch.)  The first boundary inside a unit handles the unit.
    int64_t tile = blockIdx.x;
    if (by_boundary) {
        const int64_t b = blockIdx.x / GS + 1, mem = blockIdx.x % GS;
        const int64_t pos = (b * W) / P;
        if (pos % KT == 0) return;                                   // the boundary sits on a unit's edge: nothing was cut there
        const int64_t u = pos / KT;
        if (b >= 2) {
            const int64_t prev = ((b - 1) * W) / P;
            if (prev % KT != 0 && prev / KT == u) return;            // an earlier boundary lies inside the same unit
        }
        tile = u * GS + mem;
        if (tile >= g.ntiles) return;                                // (the last unit may be short)
    }
    const int64_t unit = tile / GS, member = tile - unit * GS;
    const int64_t lo = unit * KT, hi = lo + KT;
    // first / last share (group) intersecting [lo, hi)      (P = number of groups)
    int64_t w0 = (lo * P) / W;
    while (w0 > 0 && (w0 * W) / P > lo) --w0;
    while (((w0 + 1) * W) / P <= lo) ++w0;
    int64_t w1 = w0;
    while (((w1 + 1) * W) / P < hi) ++w1;
    if (w0 == w1) return;   // one workgroup covered the whole tile and wrote C itself
    const SkTile td = sk_tile(g, tile);
    // blockIdx.y slices the tile so that a tile shared by many workgroups (tall Gram matrices) is not summed by one CU
    const int per = SLAB_ELEMS / (int)gridDim.y;
    for (int e = blockIdx.y * per + threadIdx.x; e < (int)(blockIdx.y + 1) * per; e += 256) {
        T s = 0;
        for (int64_t w = w0; w <= w1; ++w) {
            const int64_t first_unit_w = ((w * W) / P) / KT;
            const T* slab = g.slab + (2 * sk_workgroup(w, member, (int)GS) + (unit != first_unit_w ? 1 : 0)) * (int64_t)SLAB_ELEMS;
            s += slab[e];
        }
        int64_t i, j;
        if (!sk_cpos(td, g.tri, e % BM, e / BM, i, j) || (!g.tri && i >= g.M)) continue;
        T v = g.alpha * s;
        if (g.beta != T(0)) v += g.beta * g.C[i + j * g.ldc];
        g.C[i + j * g.ldc] = v;
    }
}

__global__ __launch_bounds__(256) void ssq_sum_kernel(int np, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double sm[256];
    double a = 0;
    for (int i = threadIdx.x; i < np; i += 256) a += part[i];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) sm[threadIdx.x] += sm[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += sm[0];   // accumulates: the caller zero-initialises / adds the peeled rows
}

}  // namespace

namespace rlhip {

// returns 1 if the problem was handled here, 0 if the caller should use the generic kernel, <0 on error
template <typename T>
int gemm_streamk(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda, const T* B,
                 int64_t ldb, T beta, T* C, int64_t ldc, double* ssqA_dev, int tri) {
    constexpr int BK = SkT<T>::BK, EPP = SkT<T>::EPP;
    const int num_cu = c->num_cu;
    if (transB || c->avoid_persistent) return 0;
    // fp32: the kernel carries ONE fma chain per output entry through the whole K of a tile.  Beyond ~16k products the rounding of the
    // growing partial sum (eps * K / sqrt 2) is 2-3x that of a cache-blocked host sgemm or of the split-K generic kernel (measured on
    // BQRRP 65536^2: residual per column 4e-5 -> 7.5e-5; a second accumulator level costs more registers than the kernel has: 137 ->
    // 123 TFLOP/s), so longer contractions are cut into accumulating chunks of 16384 by the caller (gemm.hip).
    if (sizeof(T) == 4 && k > 16384) return 0;
    if (n % BN || k % BK || m < BM || n <= 0 || k <= 0) return 0;
    if (m % BM && (tri || (m % BM) % EPP)) return 0;          // a partial last tile row: whole 16-byte pieces only, never in the tri map
    if (((uintptr_t)A | (uintptr_t)B) % 16 || lda % EPP || ldb % EPP) return 0;    // 16-byte aligned DMA pieces
    const int64_t tiles_m = (m + BM - 1) / BM, tiles_n = n / BN, ktiles = k / BK;
    int64_t ntiles = tiles_m * tiles_n;
    if (tri) {
        if (m != n) return 0;
        ntiles = 0;
        for (int64_t i = 0; i < tiles_m; ++i) ntiles += (tiles_m - i) / 2;      // pairs of 128-blocks from the diagonal block on
        ntiles += (tiles_m / 2 + 1) / 2;                                       // the leftover halves of the odd rows, two per tile
    }
    const int64_t W = ntiles * ktiles;
    if (W < (int64_t)num_cu * 64) return 0;   // too little work to amortise the persistent launch
    // a k x k Gram matrix with k = 256 is TWO tiles: every one of the 256 workgroups would write a partial slab (64 MB) for the fix-up to
    // sum -- 453 + 69 us at 200000 x 256 against ~300 for the tiled kernel with split-K slabs of its 3 upper 128-blocks
    if (tri && ntiles < 8) return 0;
    SkArgs<T> g;
    g.M = m; g.N = n; g.K = k; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta; g.tiles_m = tiles_m; g.tiles_n = tiles_n; g.ktiles = ktiles;
    g.tri = tri; g.ntiles = ntiles;
    int64_t P = num_cu;
    // experiments only (scripts/shard_gemm_ab.py): RLHIP_SK_TUNE="gs_nn,gs_tn,workgroups", 0 keeps the default
    static int tune[4] = {-1, 0, 0, -1};
    if (tune[0] < 0) {
        tune[0] = 0;
        if (const char* e = getenv("RLHIP_SK_TUNE")) sscanf(e, "%d,%d,%d,%d", &tune[0], &tune[1], &tune[2], &tune[3]);    // gs_nn, gs_tn, workgroups, stagger
    }
    // Triangular map with few tiles (18 at n = 1024): with P = ntiles * s workgroups every tile is cut into s EQUAL K-shares, so the ntiles
    // workgroups of one K-share walk the same rows of A at the same time (they meet in the Infinity Cache instead of streaming ntiles unrelated
    // K ranges).  Taken when it idles < 3 % of the CUs.  C3's Gram matrix, 1048576 x 1024: 256 workgroups 17.95 ms at 2329 MHz, 252: 17.81 ms at
    // 2391 MHz (profiles/round6_tri_gram_workgroups_ab.txt; 234 and fewer lose to the idle CUs).
    if (tri) {
        const int64_t per = num_cu / ntiles;
        if (per >= 1 && ntiles * per * 100 >= 97 * (int64_t)num_cu) P = ntiles * per;
    }
    if (tune[2] > 0 && tune[2] <= num_cu) P = tune[2];
    // Lockstep group size (sk_group).  Measured at C2 (200000 x 20000 x 256 fp64, kernel ms / FETCH_SIZE GB against 32.5 GB algorithmic):
    //   Y = A Omega (NN):   1: 29.8 / 72.8   2: 29.9 / 68.8   4: 29.9 / 52.1   8: 30.0 / 34.4   (16, 32: as 8)
    //   B^T = A^T Q (TN):   1: 30.1 / 63.0   2: 30.0 / 58.8   4: 30.6 / 58.7   8: 31.2 / 34.9
    // NN takes 8 (fabric traffic 2.27x -> 1.08x of the algorithmic bytes for 0.5 % of kernel time).  TN loses 4 % at 8 IN PHASE -- its members
    // read the SAME k-rows of neighbouring column blocks of A, 1.6 MB apart, at the same instant, which camps on memory channels -- so its
    // groups of 8 walk staggered (round 4, below).  The triangular map has too few tiles for whole units (18 at n = 1024) and runs ungrouped.
    int gs_want = tri ? 1 : 8;
    if (!tri && tune[transA ? 1 : 0] > 0) gs_want = tune[transA ? 1 : 0];
    g.gs = (gs_want > 1 && P % (8 * gs_want) == 0 && ntiles >= 4 * (int64_t)gs_want) ? gs_want : 1;
    // transposed product: groups of 8 walk their K range 16 tiles apart (see `rot` in the kernel).  Interleaved A/B at C2, best of 4 / median,
    // ms: groups of 2 in phase (round 3) 31.64 / 33.1;  8 in phase 32.1;  8 staggered by 16: 30.87 / 31.6;  4 by 8: 30.85 / 31.6;  8 by 64: 32.2
    g.stag = (transA && g.gs > 1) ? (tune[3] >= 0 ? tune[3] : 16) : 0;
    size_t mark = rlhip_ws_mark(c);
    g.slab = ws_alloc<T>(c, (size_t)2 * P * SLAB_ELEMS);
    if (!g.slab) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    g.ssq_part = ssqA_dev ? ws_alloc<double>(c, (size_t)P) : nullptr;
    static int want_clk = -1;
    if (want_clk < 0) { const char* e = getenv("RLHIP_SK_CLOCK"); want_clk = e ? atoi(e) : 0; }
    g.clk = want_clk ? ws_alloc<unsigned long long>(c, 4) : nullptr;
    constexpr int smem = NSTAGE * STAGE;
    // 32-bit byte offsets for the DMA pieces when the 256 rows a stage takes from either operand span < 4 GiB (ld < 2 M doubles / 4 M floats)
    const int64_t span = (int64_t)sizeof(T) * (256 * (lda > ldb ? lda : ldb) + 512);
    const bool o32 = span < ((int64_t)1 << 32);
#define RLHIP_SK_LAUNCH(KC, O)                                                                                           \
    do {                                                                                                                 \
        RLHIP_FUNC_LDS(c, (gemm_sk_kernel<T, KC, O>), smem);                                                             \
        hipLaunchKernelGGL((gemm_sk_kernel<T, KC, O>), dim3((unsigned)P), dim3(512), smem, c->stream, g);                \
    } while (0)
    if (transA) { if (o32) RLHIP_SK_LAUNCH(true, true); else RLHIP_SK_LAUNCH(true, false); }
    else { if (o32) RLHIP_SK_LAUNCH(false, true); else RLHIP_SK_LAUNCH(false, false); }
#undef RLHIP_SK_LAUNCH
    RLHIP_LAUNCH_CHECK();
    // (few tiles cut into many slabs -- the Gram matrix of a 200000 x 256 factor: 2 tiles x 128 slabs -- get more slices per tile: with 16 the
    // 32 blocks of that fix-up took as long as the product itself, 0.42 ms)
    int64_t fy = 2048 / ntiles;
    fy = fy < 16 ? 16 : (fy > 128 ? 128 : fy);
    while (SLAB_ELEMS % fy) --fy;
    {
        const int64_t Pg = P / g.gs, nbnd = (Pg - 1) * g.gs;        // share boundaries x group members
        const bool by_boundary = nbnd > 0 && ntiles > nbnd;
        hipLaunchKernelGGL(gemm_sk_fixup_kernel<T>, dim3((unsigned)(by_boundary ? nbnd : ntiles), (unsigned)fy), dim3(256), 0, c->stream, g, Pg, by_boundary ? 1 : 0);
    }
    RLHIP_LAUNCH_CHECK();
    if (ssqA_dev) {
        hipLaunchKernelGGL(ssq_sum_kernel, dim3(1), dim3(256), 0, c->stream, (int)P, g.ssq_part, ssqA_dev);
        RLHIP_LAUNCH_CHECK();
    }
    if (g.clk) {   // debug: effective shader clock of this launch (costs a host sync)
        unsigned long long h[4];
        if (hipMemcpyAsync(h, g.clk, sizeof h, hipMemcpyDeviceToHost, c->stream) == hipSuccess && rlhip_stream_sync(c) == hipSuccess && h[3] > h[1])
            fprintf(stderr, "[sk clock] %s m %lld n %lld k %lld: %.1f us at %.0f MHz\n", transA ? "TN" : "NN", (long long)m, (long long)n, (long long)k,
                    (double)(h[3] - h[1]) / 100.0, (double)(h[2] - h[0]) / ((double)(h[3] - h[1]) / 100.0));
    }
    rlhip_ws_release(c, mark);
    c->path_count[sizeof(T) == 8 ? 0 : 1]++;
    return 1;
}
template int gemm_streamk<double>(rlhip_ctx*, int, int, int64_t, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t, double, double*,
                                  int64_t, double*, int);
template int gemm_streamk<float>(rlhip_ctx*, int, int, int64_t, int64_t, int64_t, float, const float*, int64_t, const float*, int64_t, float, float*, int64_t,
                                 double*, int);

}  // namespace rlhip
