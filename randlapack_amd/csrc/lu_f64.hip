// fp64 panel step of the row-pivoted LU: the design of lu_f32.hip (rows published by the lanes of the owner's wave, per-wave decision,
// interchanges by row label, two workgroup barriers per column) with what fp64 changes:
//   * a value is two tagged words, so a row of 32 values is ONE store instruction of all 64 lanes (lane l <- word l);
//   * (|value|, row) do not fit one 64-bit key: the first maximum is found in two DPP reductions -- the maximal |value|, then the smallest
//     row among the lanes that hold it;
//   * 2 rows per thread (128 VGPRs of panel), 512 rows per workgroup, G <= 64 workgroups: up to 32768 rows below the diagonal (taller
//     panels keep the general step of lu.hip).
// BQRRP in fp64 (16384^2, b = 512) spends half its time in these panels (rl_bqrrp.hh:341-352); PLUL (rl_orth.hh:212-230) uses them too.
#include "lu_common.h"

namespace rlhip_lu {
namespace {

__device__ __forceinline__ double ld_dpp_max_step(double k, const int which) {
    int lo = __double2loint(k), hi = __double2hiint(k), lo2, hi2;
    switch (which) {   // row_ror:n inside each 16-lane row
        case 8: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false); break;
        case 4: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xF, 0xF, false); break;
        case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xF, 0xF, false); break;
        default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, false); break;
    }
    const double o = __hiloint2double(hi2, lo2);
    return o > k ? o : k;
}
__device__ __forceinline__ unsigned ld_dpp_min_step(unsigned k, const int which) {
    int o;
    switch (which) {
        case 8: o = __builtin_amdgcn_update_dpp(0, (int)k, 0x128, 0xF, 0xF, false); break;
        case 4: o = __builtin_amdgcn_update_dpp(0, (int)k, 0x124, 0xF, 0xF, false); break;
        case 2: o = __builtin_amdgcn_update_dpp(0, (int)k, 0x122, 0xF, 0xF, false); break;
        default: o = __builtin_amdgcn_update_dpp(0, (int)k, 0x121, 0xF, 0xF, false); break;
    }
    return (unsigned)o < k ? (unsigned)o : k;
}
// first maximum over the wave: (v, r) with v = -1 for "nothing to offer" (NaNs are offered as -1 too: LAPACK's strict '>' never selects them)
__device__ __forceinline__ void ld_wave_best(double& v, unsigned& r) {
    double k = v;
    k = ld_dpp_max_step(k, 8); k = ld_dpp_max_step(k, 4); k = ld_dpp_max_step(k, 2); k = ld_dpp_max_step(k, 1);
    const int lo = __double2loint(k), hi = __double2hiint(k);
    double vmax = -1.0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const double t = __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
        vmax = t > vmax ? t : vmax;
    }
    unsigned rr = (v == vmax && vmax >= 0.0) ? r : 0xffffffffu;
    rr = ld_dpp_min_step(rr, 8); rr = ld_dpp_min_step(rr, 4); rr = ld_dpp_min_step(rr, 2); rr = ld_dpp_min_step(rr, 1);
    unsigned rmin = 0xffffffffu;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)rr, l);
        rmin = t < rmin ? t : rmin;
    }
    v = vmax; r = rmin;
}

constexpr int LD_RPT = 2;                 // rows per thread: 512 rows per workgroup
struct LuF64Shared {
    unsigned rb[4][2 * PB];               // wave-private hand-over lines (owner lane -> the wave): word 2 c = low half of value c
    double bv[4]; unsigned br[4];
    double piv[2][PB];                    // by column parity
};

template <int C>
__device__ __forceinline__ void lu_f64_step(const LuArgs<double>& g, LuRegState<double, LD_RPT>& st, LuF64Shared& sh) {
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, me = (int)blockIdx.x;
    const unsigned m = (unsigned)g.m;
    const unsigned j = (unsigned)g.j0 + C;
    constexpr int par = C & 1;
    const unsigned tag = g.tag_base + C + 1;
    unsigned long long* base = g.tw + (size_t)par * (size_t)(2 * G + G + 2 * G * PB + 2 * PB);
    unsigned long long* cw0 = base, *cw1 = cw0 + 2 * G, *rw = cw1 + G;
    auto putw = [&](unsigned long long* q, unsigned payload) {
        __hip_atomic_store(q, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    LU_MARK(0)
    // ---- local candidate
    double bv = -1.0; unsigned br = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < LD_RPT; ++q) {
        const unsigned r = (unsigned)st.gr[q];
        const double a = fabs(st.x[q][C]);
        if (r >= j && r < m && (a > bv || (a == bv && r < br))) { bv = a; br = r; }      // (a != a fails both tests)
    }
    ld_wave_best(bv, br);
    if (lane == 0) { sh.bv[wid] = bv; sh.br[wid] = br; }
    __syncthreads();
    {
        double v1 = sh.bv[0]; unsigned r1 = sh.br[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const double v2 = sh.bv[w]; const unsigned r2 = sh.br[w];
            if (v2 > v1 || (v2 == v1 && r2 < r1)) { v1 = v2; r1 = r2; }
        }
        bv = v1; br = r1;
    }
    const unsigned lrow = (bv >= 0.0 && br < m) ? br : m;
    if (tid == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(bv);
        putw(cw0 + 2 * me, (unsigned)bits); putw(cw0 + 2 * me + 1, (unsigned)(bits >> 32)); putw(cw1 + me, lrow);
    }
    if (lrow >= m) { if (tid < 2 * PB) putw(rw + (size_t)me * 2 * PB + tid, 0u); }     // nothing to offer: a dummy row, readers never wait for one
    else {
#pragma unroll
        for (int q = 0; q < LD_RPT; ++q) {
            const bool own = ((unsigned)st.gr[q] == lrow);
            if (__builtin_amdgcn_ballot_w64(own)) {                       // wave-uniform
                if (own) {
#pragma unroll
                    for (int c2 = 0; c2 < PB; ++c2) {
                        sh.rb[wid][2 * c2] = (unsigned)__double2loint(st.x[q][c2]);
                        sh.rb[wid][2 * c2 + 1] = (unsigned)__double2hiint(st.x[q][c2]);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): the line is written (same wave)
                __builtin_amdgcn_wave_barrier();
                putw(rw + (size_t)me * 2 * PB + lane, sh.rb[wid][lane]);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    LU_MARK(1)
    // ---- one batch of loads: record of workgroup `lane` (every wave reads all G <= 64 records) and element tid % 32 of the candidate
    //      rows of workgroups tid / 32 + 8 u
    constexpr int PF = 8, NWD = 3 + 2 * PF;
    const unsigned long long* ad[NWD]; bool need[NWD]; unsigned got[NWD];
    {
        const int wl = lane < G ? lane : 0;
        ad[0] = cw0 + 2 * wl; need[0] = lane < G;
        ad[1] = cw0 + 2 * wl + 1; need[1] = lane < G;
        ad[2] = cw1 + wl; need[2] = lane < G;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int wu = (tid >> 5) + 8 * u;
            const unsigned long long* q = rw + ((size_t)(wu < G ? wu : 0) * PB + (tid & 31)) * 2;
            ad[3 + 2 * u] = q; ad[4 + 2 * u] = q + 1;
            need[3 + 2 * u] = wu < G; need[4 + 2 * u] = wu < G;
        }
    }
    lu_tag_get_n<NWD>(ad, need, tag, got, g.info);
    LU_MARK(2)
    // ---- decision, per wave
    double gv = -1.0; unsigned gr = 0xffffffffu;
    if (lane < G && got[2] < m) { gv = __longlong_as_double((long long)(((unsigned long long)got[1] << 32) | got[0])); gr = got[2]; }
    const double myv = gv; const unsigned myr = gr;
    ld_wave_best(gv, gr);
    const bool any = (gv >= 0.0 && gr < m);
    const unsigned p = any ? gr : j;                                        // empty / NaN column: no exchange, (dummy) zero pivot row
    const unsigned long long whob = __builtin_amdgcn_ballot_w64(any && myv == gv && myr == gr);
    const int wstar = whob ? (int)__builtin_ctzll(whob) : 0;
    if ((tid >> 5) == (wstar & 7)) {
        unsigned lo = got[3], hi = got[4];
#pragma unroll
        for (int u = 1; u < PF; ++u) { const bool s = ((wstar >> 3) == u); lo = s ? got[3 + 2 * u] : lo; hi = s ? got[4 + 2 * u] : hi; }
        sh.piv[par][tid & 31] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    if (me == 0 && tid == 0) g.ipiv[j] = (int64_t)p + 1;
    __syncthreads();
    LU_MARK(3)
    // ---- interchange j <-> p by LABEL, then eliminate the rows below j
    const double* s_piv = sh.piv[par];
    const double piv = s_piv[C];
    const double rp = 1.0 / piv;
#pragma unroll
    for (int q = 0; q < LD_RPT; ++q) {
        unsigned r = (unsigned)st.gr[q];
        if (p != j) { r = (r == j) ? p : (r == p) ? j : r; st.gr[q] = (int64_t)r; }
        if (piv != 0.0 && r > j && r < m) {
            const double l = st.x[q][C] * rp;
            st.x[q][C] = l;
#pragma unroll
            for (int c2 = C + 1; c2 < PB; ++c2) st.x[q][c2] -= l * s_piv[c2];
        }
    }
    if (piv == 0.0 && me == 0 && tid == 0 && *g.info == 0) *g.info = (int)(j + 1);
    LU_MARK(4)
}
template <int C>
__device__ __forceinline__ void lu_f64_steps(const LuArgs<double>& g, LuRegState<double, LD_RPT>& st, LuF64Shared& sh) {
    if constexpr (C < PB) {
        if (C < g.pb) {
            lu_f64_step<C>(g, st, sh);
            lu_f64_steps<C + 1>(g, st, sh);
        }
    }
}
__global__ __launch_bounds__(256) void getrf_panel_f64_kernel(LuArgs<double> g) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    __shared__ LuF64Shared sh;
    const int tid = threadIdx.x;
    const int64_t me = blockIdx.x;
    const int pb = g.pb;
    const int64_t j0 = g.j0, m = g.m;
    const int64_t lo = j0 + me * (256 * LD_RPT);
    LuRegState<double, LD_RPT> st;
#pragma unroll
    for (int q = 0; q < LD_RPT; ++q) {
        st.gr[q] = lo + tid + 256 * q;
        const int64_t rr = st.gr[q] < m ? st.gr[q] : m - 1;
#pragma unroll
        for (int c = 0; c < PB; ++c) st.x[q][c] = g.A[rr + (j0 + (c < pb ? c : pb - 1)) * g.lda];    // clamped row and column: unconditional, all in flight together
    }
    // (the selects come after ALL loads: written as `cond ? load : 0` per entry, hipcc sinks every load into its own branch with an
    // s_waitcnt vmcnt(0) behind it -- 128 dependent L2 round trips = ~30 us per panel launch)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < LD_RPT; ++q) {
#pragma unroll
        for (int c = 0; c < PB; ++c) st.x[q][c] = (st.gr[q] < m && c < pb) ? st.x[q][c] : 0.0;
    }
#ifdef RLHIP_LU_PROF
    for (int i = 0; i < 5; ++i) st.pf[i] = 0;
    st.pt = wall_clock64();
#endif
    lu_f64_steps<0>(g, st, sh);
#ifdef RLHIP_LU_PROF
    if (me == (int64_t)gridDim.x / 2 && tid == 0) for (int i = 0; i < 5; ++i) atomicAdd((unsigned long long*)(g.diag_data + 2 * PB) + i, (unsigned long long)st.pf[i]);
#endif
#pragma unroll
    for (int q = 0; q < LD_RPT; ++q) {
        if (st.gr[q] < m) {
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (c < pb) g.A[st.gr[q] + (j0 + c) * g.lda] = st.x[q][c];
        }
    }
}

}  // namespace

void launch_getrf_panel_f64(const LuArgs<double>& g, unsigned G, hipStream_t stream) {
    hipLaunchKernelGGL(getrf_panel_f64_kernel, dim3(G), dim3(256), 0, stream, g);
}

}  // namespace rlhip_lu
