// fp32 panel step of the row-pivoted LU (see lu.hip for the algorithm and the host driver, lu_common.h for the shared pieces)
#include "lu_common.h"

namespace rlhip_lu {
namespace {

// ---- fp32 panels of up to 65536 rows (BQRRP's transposed sketch, rl_bqrrp.hh:341-352): the column step rewritten around what the per-phase
// profile of the general step showed (us per column at 65536 x 2048: candidate 1.2, publish 1.3, exchange 1.8, decision 1.1, eliminate
// 1.7 -- the hand-off itself is a quarter; the rest is LOCAL work with one workgroup per CU and nothing to hide latencies behind):
//   * a row is published by 32 lanes with ONE store instruction (the owner lane hands its 32 registers to its own wave through a
//     wave-private LDS line; no workgroup barrier) instead of 32 store instructions from one lane;
//   * (|value|, row) travel as ONE 64-bit key {float bits : 2^32 - 1 - row}: a maximum over keys is LAPACK's first maximum, the wave
//     reduction is four DPP row rotations + four readlanes instead of six dependent cross-lane shuffles of three values;
//   * G <= 64 workgroups, so every WAVE reads all records itself (lane l <- workgroup l) and decides without a workgroup barrier;
//   * two workgroup barriers per column (candidate combine, pivot row staged) instead of four;
//   * rows are never moved between registers: an interchange j <-> p only swaps the two ROW LABELS (st.gr) -- the slot that held row p
//     now is row j (final, no longer eliminated), the slot that held row j carries on as row p -- and every slot is written to the row
//     its label names when the panel is done.  The old diagonal row is therefore never published or fetched, and the 256 conditional
//     moves + 64 LDS reads per thread and column of the value swap are gone (the step was VALU-issue bound: ~500 instructions per thread).
// Same decisions and the same arithmetic per row as the general step: identical pivots and factors.
__device__ __forceinline__ unsigned long long lu_dpp_max_step(unsigned long long k, const int which) {
    int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32), lo2, hi2;
    switch (which) {   // row_ror:n inside each 16-lane row
        case 8: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false); break;
        case 4: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xF, 0xF, false); break;
        case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xF, 0xF, false); break;
        default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, false); break;
    }
    const unsigned long long o = ((unsigned long long)(unsigned)hi2 << 32) | (unsigned)lo2;
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long lu_wave_max_u64(unsigned long long k) {
    k = lu_dpp_max_step(k, 8); k = lu_dpp_max_step(k, 4); k = lu_dpp_max_step(k, 2); k = lu_dpp_max_step(k, 1);
    const int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32);
    unsigned long long r = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const unsigned long long v = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, l) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, l);
        r = v > r ? v : r;
    }
    return r;
}
// key of a candidate: larger value wins, equal values -> smaller row wins; 0 = nothing to offer (also for NaN, which LAPACK's
// strict '>' search never selects either)
__device__ __forceinline__ unsigned long long lu_key(float absval, unsigned row) {
    return (absval == absval) ? (((unsigned long long)__float_as_uint(absval) << 32) | (0xffffffffu - row)) : 0ull;
}

constexpr int LF_RPT = 4;                 // rows per thread: 1024 rows per workgroup
struct LuF32Shared {
    float rb[4][PB];                      // wave-private hand-over lines (owner lane -> 32 lanes)
    unsigned long long key[4];
    float piv[2][PB];                     // by column parity: no barrier needed before the next column rewrites it
};

template <int C>
__device__ __forceinline__ void lu_f32_step(const LuArgs<float>& g, LuRegState<float, LF_RPT>& st, LuF32Shared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, me = (int)blockIdx.x;
    const unsigned m = (unsigned)g.m;
    const unsigned j = (unsigned)g.j0 + C;
    constexpr int par = C & 1;
    const unsigned tag = g.tag_base + C + 1;
    unsigned long long* base = g.tw + (size_t)par * (size_t)(2 * G + G * PB + PB);
    unsigned long long* cw0 = base, *cw1 = cw0 + G, *rw = cw1 + G;
    auto putw = [&](unsigned long long* q, unsigned payload) {
        __hip_atomic_store(q, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // owner lane (row `want`) -> its wave's 32 low lanes -> one store instruction into dst[0..31]
    auto publish_row = [&](unsigned want, unsigned long long* dst) {
#pragma unroll
        for (int q = 0; q < LF_RPT; ++q) {
            const bool own = ((unsigned)st.gr[q] == want);
            if (__builtin_amdgcn_ballot_w64(own)) {                       // wave-uniform
                if (own) {
#pragma unroll
                    for (int c2 = 0; c2 < PB; ++c2) sh.rb[wid][c2] = st.x[q][c2];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): the line is written (same wave)
                __builtin_amdgcn_wave_barrier();
                if (lane < PB) putw(dst + lane, __float_as_uint(sh.rb[wid][lane]));
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    LU_MARK(0)
    // ---- local candidate
    unsigned long long key = 0;
#pragma unroll
    for (int q = 0; q < LF_RPT; ++q) {
        const unsigned r = (unsigned)st.gr[q];
        const unsigned long long k2 = (r >= j && r < m) ? lu_key(fabsf(st.x[q][C]), r) : 0ull;
        key = k2 > key ? k2 : key;
    }
    key = lu_wave_max_u64(key);
    if (lane == 0) sh.key[wid] = key;
    __syncthreads();
    {
        unsigned long long k1 = sh.key[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) k1 = sh.key[w] > k1 ? sh.key[w] : k1;
        key = k1;
    }
    const unsigned lrow = key ? 0xffffffffu - (unsigned)key : m;
    if (tid == 0) { putw(cw0 + me, (unsigned)(key >> 32)); putw(cw1 + me, lrow); }
    if (lrow >= m) { if (tid < PB) putw(rw + (size_t)me * PB + tid, 0u); }        // nothing to offer: a dummy row, readers never wait for one
    else publish_row(lrow, rw + (size_t)me * PB);
    LU_MARK(1)
    // ---- one batch of loads: record of workgroup `lane` (every wave reads all G <= 64 records), element tid % 32 of the diagonal row and
    //      of the candidate rows of workgroups tid / 32 + 8 u
    constexpr int PF = 8, NWD = 2 + PF;
    const unsigned long long* ad[NWD]; bool need[NWD]; unsigned got[NWD];
    {
        const int wl = lane < G ? lane : 0;
        ad[0] = cw0 + wl; need[0] = lane < G;
        ad[1] = cw1 + wl; need[1] = lane < G;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int wu = (tid >> 5) + 8 * u;
            ad[2 + u] = rw + (size_t)(wu < G ? wu : 0) * PB + (tid & 31); need[2 + u] = wu < G;
        }
    }
    lu_tag_get_n<NWD>(ad, need, tag, got, g.info);
    LU_MARK(2)
    // ---- decision, per wave
    const unsigned long long myk = (lane < G && got[1] < m) ? (((unsigned long long)got[0] << 32) | (0xffffffffu - got[1])) : 0ull;
    const unsigned long long gk = lu_wave_max_u64(myk);
    const unsigned p = gk ? 0xffffffffu - (unsigned)gk : j;                 // empty / NaN column: no exchange, (dummy) zero pivot row
    // the winner is the workgroup whose record carries the maximal key (keys are unique: rows are).  NOT (p - j0) / 1024: with
    // interchanges done by label a row lives wherever the slot that received its label is
    const unsigned long long whob = __builtin_amdgcn_ballot_w64(gk != 0ull && myk == gk);
    const int wstar = whob ? (int)__builtin_ctzll(whob) : 0;
    if ((tid >> 5) == (wstar & 7)) {
        unsigned pv = got[2];
#pragma unroll
        for (int u = 1; u < PF; ++u) pv = ((wstar >> 3) == u) ? got[2 + u] : pv;
        sh.piv[par][tid & 31] = __uint_as_float(pv);                        // (p == j: the winner's candidate row IS row j)
    }
    if (me == 0 && tid == 0) g.ipiv[j] = (int64_t)p + 1;
    __syncthreads();
    LU_MARK(3)
    // ---- interchange j <-> p by LABEL, then eliminate the rows below j
    const float* s_piv = sh.piv[par];
    const float piv = s_piv[C];
    const float rp = 1.0f / piv;
#pragma unroll
    for (int q = 0; q < LF_RPT; ++q) {
        unsigned r = (unsigned)st.gr[q];
        if (p != j) { r = (r == j) ? p : (r == p) ? j : r; st.gr[q] = (int64_t)r; }
        if (piv != 0.0f && r > j && r < m) {
            const float l = st.x[q][C] * rp;
            st.x[q][C] = l;
#pragma unroll
            for (int c2 = C + 1; c2 < PB; ++c2) st.x[q][c2] -= l * s_piv[c2];
        }
    }
    if (piv == 0.0f && me == 0 && tid == 0 && *g.info == 0) *g.info = (int)(j + 1);
    LU_MARK(4)
}
template <int C>
__device__ __forceinline__ void lu_f32_steps(const LuArgs<float>& g, LuRegState<float, LF_RPT>& st, LuF32Shared& sh) {
    if constexpr (C < PB) {
        if (C < g.pb) {
            lu_f32_step<C>(g, st, sh);
            lu_f32_steps<C + 1>(g, st, sh);
        }
    }
}
__global__ __launch_bounds__(256) void getrf_panel_f32_kernel(LuArgs<float> g) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    __shared__ LuF32Shared sh;
    const int tid = threadIdx.x;
    const int64_t me = blockIdx.x;
    const int pb = g.pb;
    const int64_t j0 = g.j0, m = g.m;
    const int64_t lo = j0 + me * (256 * LF_RPT);
    LuRegState<float, LF_RPT> st;
#pragma unroll
    for (int q = 0; q < LF_RPT; ++q) {
        st.gr[q] = lo + tid + 256 * q;
        const int64_t rr = st.gr[q] < m ? st.gr[q] : m - 1;
#pragma unroll
        for (int c = 0; c < PB; ++c) st.x[q][c] = g.A[rr + (j0 + (c < pb ? c : pb - 1)) * g.lda];    // clamped row and column: unconditional, all in flight together
    }
    // (the selects come after ALL loads: written as `cond ? load : 0` per entry, hipcc sinks every load into its own branch with an
    // s_waitcnt vmcnt(0) behind it -- 128 dependent L2 round trips = ~30 us per panel launch)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < LF_RPT; ++q) {
#pragma unroll
        for (int c = 0; c < PB; ++c) st.x[q][c] = (st.gr[q] < m && c < pb) ? st.x[q][c] : 0.0f;
    }
#ifdef RLHIP_LU_PROF
    for (int i = 0; i < 5; ++i) st.pf[i] = 0;
    st.pt = wall_clock64();
#endif
    lu_f32_steps<0>(g, st, sh);
#ifdef RLHIP_LU_PROF
    if (me == (int64_t)gridDim.x / 2 && tid == 0) for (int i = 0; i < 5; ++i) atomicAdd((unsigned long long*)(g.diag_data + 2 * PB) + i, (unsigned long long)st.pf[i]);
#endif
#pragma unroll
    for (int q = 0; q < LF_RPT; ++q) {
        if (st.gr[q] < m) {
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (c < pb) g.A[st.gr[q] + (j0 + c) * g.lda] = st.x[q][c];
        }
    }
}

}  // namespace

void launch_getrf_panel_f32(const LuArgs<float>& g, unsigned G, hipStream_t stream) {
    hipLaunchKernelGGL(getrf_panel_f32_kernel, dim3(G), dim3(256), 0, stream, g);
}

}  // namespace rlhip_lu
