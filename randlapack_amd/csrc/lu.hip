// Row-pivoted LU (lapack::getrf) of a tall-skinny matrix on the device, and the pivot post-processing of BQRRP's
// LU-based qrcp_wide (RandLAPACK/drivers/rl_bqrrp.hh:341-352; rl_bqrrp_gpu.hh:359-364; PLUL rl_orth.hh:212-230).
//
// Partial pivoting is inherently one decision per column, and the decision must equal LAPACK's (first maximum of
// |a(j:m, j)|) for the pivot order to be bit-identical given the same sketch.  Organisation:
//   * blocked right-looking, panel width 32: the trailing update and the U12 solve are MFMA / thread-per-column work;
//   * the panel itself is ONE persistent launch with the rows below the diagonal dealt out to the workgroups.  Two kernels:
//     - register kernel (panels of >= 1024 rows): a thread owns 2 (fp64) / 4 (fp32) rows of the 32-column panel in VGPRs, so the
//       elimination is pure register FMAs against the broadcast pivot row and the local pivot search a wave shuffle reduction.
//       Per column every workgroup publishes its best candidate TOGETHER WITH that row's 32 values (and the owner of the diagonal
//       row publishes that row) as 8-byte words {column tag : 32-bit payload}; readers re-read a word until it carries the tag of
//       the column -- no store drain, no barrier counter, no acquire fence ("flag-less" exchange, bounded spins);
//     - LDS kernel (short panels): the workgroup's piece of the panel lives in LDS and the columns are separated by a grid
//       rendezvous (8-byte write-through stores read after one L1 invalidate, double-buffered by column parity);
//     in both, after the exchange everybody knows the winner, the two owners swap rows and every workgroup eliminates its own rows.
//   * dlaswp on the columns outside the panel is a thread-per-column kernel walking the 32 swaps in order.
#include "rlhip_internal.h"
#include <algorithm>
#include <cstdio>

namespace rlhip {
template <typename T>
int gemm_impl(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
              const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri, double* ssqA_dev = nullptr, int* ssq_done = nullptr);
}

#include "lu_common.h"

namespace {

using namespace rlhip_lu;

template <typename T>
__device__ __forceinline__ void pstore(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void lu_barrier(unsigned* bar, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void getrf_panel_kernel(LuArgs<T> g) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    extern __shared__ __attribute__((aligned(16))) unsigned char lu_smem[];
    T* P = reinterpret_cast<T*>(lu_smem);                 // [pb][rpw] : local rows of the panel, column-major
    __shared__ T s_val[256];
    __shared__ int64_t s_row[256];
    __shared__ int s_w[256];
    __shared__ T s_piv[PB];
    const int tid = threadIdx.x;
    const int64_t G = gridDim.x, me = blockIdx.x;
    const int pb = g.pb;
    const int64_t j0 = g.j0, rpw = g.rpw;
    const int64_t lo = j0 + me * rpw;                      // first global row of this workgroup
    int64_t hi = lo + rpw; if (hi > g.m) hi = g.m;
    const int64_t nloc = hi > lo ? hi - lo : 0;
    // stage
    for (int64_t e = tid; e < (int64_t)pb * rpw; e += 256) {
        const int64_t r = e % rpw; const int c = (int)(e / rpw);
        P[c * rpw + r] = (r < nloc) ? g.A[(lo + r) + (j0 + c) * g.lda] : T(0);
    }
    __syncthreads();
    unsigned epoch = 0;
    for (int c = 0; c < pb; ++c) {
        const int64_t j = j0 + c;                          // global diagonal row / column
        const int par = c & 1;
        // ---- local candidate: first maximum of |P[c][r]| over owned rows >= j
        T bv = T(-1); int64_t br = g.m;
        for (int64_t r = tid; r < nloc; r += 256) {
            const int64_t gr = lo + r;
            if (gr < j) continue;
            T v = fabs(P[c * rpw + r]);
            if (v > bv) { bv = v; br = gr; }               // increasing rows per thread: strict > keeps the first
        }
        s_val[tid] = bv; s_row[tid] = br;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                T v2 = s_val[tid + st]; int64_t r2 = s_row[tid + st];
                if (r2 < g.m && (v2 > s_val[tid] || (v2 == s_val[tid] && r2 < s_row[tid]))) { s_val[tid] = v2; s_row[tid] = r2; }
            }
            __syncthreads();
        }
        const T lbest = s_val[0]; const int64_t lrow = s_row[0];
        if (tid == 0) {
            pstore(g.cand_val + par * G + me, lbest);
            __hip_atomic_store(g.cand_row + par * G + me, lrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lrow < g.m && tid < pb) pstore(g.cand_data + ((int64_t)par * G + me) * PB + tid, P[tid * rpw + (lrow - lo)]);
        if (j >= lo && j < hi && tid < pb) pstore(g.diag_data + par * PB + tid, P[tid * rpw + (j - lo)]);
        __syncthreads();
        lu_barrier(g.bar, (unsigned)(G * (++epoch)));
        // ---- winner (every workgroup, redundantly): max |value|, ties -> smallest row
        {
            T v = T(-1); int64_t r = g.m; int w = 0;
            for (int64_t ww = tid; ww < G; ww += 256) {
                T v2 = g.cand_val[par * G + ww]; int64_t r2 = g.cand_row[par * G + ww];
                if (r2 < g.m && (v2 > v || (v2 == v && r2 < r))) { v = v2; r = r2; w = (int)ww; }
            }
            s_val[tid] = v; s_row[tid] = r; s_w[tid] = w;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if (tid < st) {
                    T v2 = s_val[tid + st]; int64_t r2 = s_row[tid + st];
                    if (r2 < g.m && (v2 > s_val[tid] || (v2 == s_val[tid] && r2 < s_row[tid]))) {
                        s_val[tid] = v2; s_row[tid] = r2; s_w[tid] = s_w[tid + st];
                    }
                }
                __syncthreads();
            }
        }
        int64_t p = s_row[0]; const int wstar = s_w[0];
        if (p >= g.m) p = j;                                 // empty / NaN column: no exchange
        const T* prow = g.cand_data + ((int64_t)par * G + wstar) * PB;   // contents of row p (becomes row j)
        const T* drow = g.diag_data + par * PB;                          // contents of row j (moves to row p)
        __syncthreads();
        if (me == 0 && tid == 0) g.ipiv[j] = p + 1;
        // ---- exchange rows j <-> p inside the LDS pieces
        if (p != j) {
            if (j >= lo && j < hi && tid < pb) P[tid * rpw + (j - lo)] = prow[tid];
            if (p >= lo && p < hi && tid < pb) P[tid * rpw + (p - lo)] = drow[tid];
        }
        __syncthreads();
        // the pivot row (the row that now sits at position j) goes through LDS: one parallel fetch by 32 lanes instead of up
        // to 31 branch-separated global loads per thread
        if (tid < PB) s_piv[tid] = (tid < pb) ? ((p != j) ? prow[tid] : drow[tid]) : T(0);
        __syncthreads();
        const T piv = s_piv[c];
        if (piv == T(0)) {
            if (me == 0 && tid == 0 && *g.info == 0) *g.info = (int)(j + 1);
        } else {
            // ---- eliminate: own rows r > j
            const T rp = T(1) / piv;
            T u[PB];
#pragma unroll
            for (int c2 = 0; c2 < PB; ++c2) u[c2] = (c2 > c) ? s_piv[c2] : T(0);
            for (int64_t r = tid; r < nloc; r += 256) {
                if (lo + r <= j) continue;
                const T l = P[c * rpw + r] * rp;
                P[c * rpw + r] = l;
#pragma unroll
                for (int c2 = 0; c2 < PB; ++c2)
                    if (c2 > c && c2 < pb) P[c2 * rpw + r] -= l * u[c2];
            }
        }
        __syncthreads();
    }
    for (int64_t e = tid; e < (int64_t)pb * rpw; e += 256) {
        const int64_t r = e % rpw; const int c = (int)(e / rpw);
        if (r < nloc) g.A[(lo + r) + (j0 + c) * g.lda] = P[c * rpw + r];
    }
}

// ---- register-resident variant.  A thread owns RPT rows of the panel (rows lo + tid + 256 q) as RPT x 32 values in VGPRs, so the
// elimination is pure register FMAs against the broadcast pivot row, the local pivot search is a wave shuffle reduction, and a
// workgroup covers 256 * RPT rows: the per-column rendezvous has 4x fewer participants than the LDS version above (the atomic
// arrivals at the barrier word serialise in L2, ~15 ns each).  Same decisions, same arithmetic order per row -> same pivots and
// factors bit for bit.
template <typename T>
__device__ __forceinline__ void argmax_take(T& v, int64_t& r, T v2, int64_t r2, int64_t m) {
    if (r2 < m && (v2 > v || (v2 == v && r2 < r))) { v = v2; r = r2; }
}
// spin until the 8-byte word carries `tag` in its upper half (bounded: ~seconds), return the payload.  Kept out of line: it is
// called from 32 fully unrolled column steps.
__device__ __attribute__((noinline)) unsigned lu_tag_get(const unsigned long long* q, unsigned tag, int* info) {
    unsigned long long w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while ((unsigned)(w >> 32) != tag) {
        if (++spins > (1 << 22)) { atomicExch(info, -7); break; }
        __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return (unsigned)w;
}

// one column step with the column index as a template parameter: every x[q][c] index is a compile-time constant, so the panel
// really stays in registers (a runtime-indexed loop put it in scratch)
template <typename T, int RPT, int C, bool TAG>
__device__ __forceinline__ void lu_reg_step(const LuArgs<T>& g, LuRegState<T, RPT>& st, unsigned& epoch, T* s_wv, int64_t* s_wr, int* s_ww, T* s_piv,
                                            T* s_drow) {
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t G = gridDim.x, me = blockIdx.x, m = g.m;
    const int64_t j = g.j0 + C;
    constexpr int par = C & 1;
    // ---- local candidate: first maximum of |x[.][C]| over owned rows >= j (rows increase with q, then with tid)
    T bv = T(-1); int64_t br = m;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const T v = fabs(st.x[q][C]);
        if (st.gr[q] >= j && st.gr[q] < m && v > bv) { bv = v; br = st.gr[q]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T v2 = __shfl_xor(bv, off); const int64_t r2 = __shfl_xor(br, off);
        argmax_take(bv, br, v2, r2, m);
    }
    if (lane == 0) { s_wv[wid] = bv; s_wr[wid] = br; }
    __syncthreads();
    T lbest = s_wv[0]; int64_t lrow = s_wr[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) argmax_take(lbest, lrow, s_wv[w], s_wr[w], m);
    int64_t p; int wstar;
    LU_MARK(0)
    if constexpr (TAG) {
        // ---- flag-less exchange: every published item travels as 8-byte words {tag : 32-bit payload} (a double is two words); a
        //      reader simply re-reads a word until it carries this step's tag.  No store drain, no barrier counter, no acquire fence:
        //      the chain per column is "store lands -> poll sees it" instead of drain + arrive + poll + read.
        //      Slots alternate by column parity; a workgroup publishes column c + 1 only after it has consumed everybody's column c
        //      records, so nobody can still be reading the slot a fast workgroup overwrites two columns later.
        constexpr int W = (int)sizeof(T) / 4;                     // tagged words per value
        const unsigned tag = g.tag_base + C + 1;
        unsigned long long* base = g.tw + (size_t)par * (size_t)(W * G + G + W * G * PB + W * PB);
        unsigned long long* cw0 = base, *cw1 = cw0 + W * G, *rw = cw1 + G, *dw = rw + W * G * PB;
        auto putw = [&](unsigned long long* q, unsigned payload) {
            __hip_atomic_store(q, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto putv = [&](unsigned long long* q, int64_t idx, T v) {
            if constexpr (W == 1) putw(q + idx, __float_as_uint((float)v));
            else { const unsigned long long bits = (unsigned long long)__double_as_longlong((double)v); putw(q + 2 * idx, (unsigned)bits); putw(q + 2 * idx + 1, (unsigned)(bits >> 32)); }
        };
        auto getv = [&](const unsigned long long* q, int64_t idx) -> T {
            if constexpr (W == 1) return (T)__uint_as_float(lu_tag_get(q + idx, tag, g.info));
            else {
                const unsigned lo = lu_tag_get(q + 2 * idx, tag, g.info), hi = lu_tag_get(q + 2 * idx + 1, tag, g.info);
                return (T)__longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            }
        };
        if (tid == 0) { putv(cw0, me, lbest); putw(cw1 + me, (unsigned)lrow); }
        if (lrow >= m && tid < PB) putv(rw, me * PB + tid, T(0));   // nothing to offer: a dummy row, so that readers never wait for one
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            if (st.gr[q] == lrow && lrow < m) {
#pragma unroll
                for (int c2 = 0; c2 < PB; ++c2) putv(rw, me * PB + c2, st.x[q][c2]);
            }
            if (st.gr[q] == j) {
#pragma unroll
                for (int c2 = 0; c2 < PB; ++c2) putv(dw, c2, st.x[q][c2]);
            }
        }
        // Everything a thread needs from the other workgroups in ONE batch of loads: the record of workgroup `tid` (tid < G), element
        // tid % 32 of the diagonal row, and -- the winner's row is needed right after the decision -- element tid % 32 of the candidate
        // rows of workgroups tid / 32 + 8 u, u < 8 (G <= 64 covers 65536 rows in fp32).  Three dependent round trips before (record
        // value, record row, then the winner's row after the decision) are one now.
        constexpr int PF = 8;
        const bool pf_ok = G <= 8 * PF;
        constexpr int NWD = W + 1 + W + PF * W;
        const unsigned long long* ad[NWD]; bool need[NWD]; unsigned got[NWD];
        {
            const int64_t wme = (tid < G) ? tid : 0;
#pragma unroll
            for (int h = 0; h < W; ++h) { ad[h] = cw0 + W * wme + h; need[h] = tid < G; }
            ad[W] = cw1 + wme; need[W] = tid < G;
#pragma unroll
            for (int h = 0; h < W; ++h) { ad[W + 1 + h] = dw + W * (tid & 31) + h; need[W + 1 + h] = true; }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int64_t wu = (tid >> 5) + 8 * u;
#pragma unroll
                for (int h = 0; h < W; ++h) {
                    ad[2 * W + 1 + u * W + h] = rw + W * ((wu < G ? wu : 0) * PB + (tid & 31)) + h;
                    need[2 * W + 1 + u * W + h] = pf_ok && wu < G;
                }
            }
        }
        LU_MARK(1)
        lu_tag_get_n<NWD>(ad, need, tag, got, g.info);
        LU_MARK(2)
        auto dec = [&](int i0) -> T {
            if constexpr (W == 1) return (T)__uint_as_float(got[i0]);
            else return (T)__longlong_as_double((long long)(((unsigned long long)got[i0 + 1] << 32) | got[i0]));
        };
        const T dv_pref = dec(W + 1);
        {
            T v = T(-1); int64_t r = m; int w = 0;
            if (tid < G) {
                const T v2 = dec(0); const int64_t r2 = (int64_t)got[W];
                if (r2 < m) { v = v2; r = r2; w = tid; }
            }
            for (int64_t ww = tid + 256; ww < G; ww += 256) {          // (G > 256 never happens: one workgroup per 512+ rows, <= num_cu)
                const T v2 = getv(cw0, ww); const int64_t r2 = (int64_t)lu_tag_get(cw1 + ww, tag, g.info);
                if (r2 < m && (v2 > v || (v2 == v && r2 < r))) { v = v2; r = r2; w = (int)ww; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const T v2 = __shfl_xor(v, off); const int64_t r2 = __shfl_xor(r, off); const int w2 = __shfl_xor(w, off);
                if (r2 < m && (v2 > v || (v2 == v && r2 < r))) { v = v2; r = r2; w = w2; }
            }
            if (lane == 0) { s_wv[wid] = v; s_wr[wid] = r; s_ww[wid] = w; }
        }
        __syncthreads();
        T gv = s_wv[0]; p = s_wr[0]; wstar = s_ww[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_wr[w] < m && (s_wv[w] > gv || (s_wv[w] == gv && s_wr[w] < p))) { gv = s_wv[w]; p = s_wr[w]; wstar = s_ww[w]; }
        if (p >= m) p = j;
        if (pf_ok) {
            // the thread group (tid / 32) == wstar % 8 holds the winner's row in slot wstar / 8
            if ((tid >> 5) == (wstar & 7)) {
                T pv = dec(2 * W + 1);
#pragma unroll
                for (int u = 1; u < PF; ++u) pv = ((wstar >> 3) == u) ? dec(2 * W + 1 + u * W) : pv;
                s_piv[tid & 31] = (p != j) ? pv : dv_pref;
            }
            if (tid < PB) s_drow[tid] = dv_pref;
        } else if (tid < PB) {
            s_drow[tid] = dv_pref;
            s_piv[tid] = (p != j) ? getv(rw, (int64_t)wstar * PB + tid) : dv_pref;
        }
    } else {
    if (tid == 0) {
        pstore(g.cand_val + par * G + me, lbest);
        __hip_atomic_store(g.cand_row + par * G + me, lrow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the owners of the candidate row and of the diagonal row publish those rows' 32 values
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        if (st.gr[q] == lrow && lrow < m) {
            T* dst = g.cand_data + ((int64_t)par * G + me) * PB;
#pragma unroll
            for (int c2 = 0; c2 < PB; ++c2) pstore(dst + c2, st.x[q][c2]);
        }
        if (st.gr[q] == j) {
            T* dst = g.diag_data + par * PB;
#pragma unroll
            for (int c2 = 0; c2 < PB; ++c2) pstore(dst + c2, st.x[q][c2]);
        }
    }
    lu_barrier(g.bar, (unsigned)(G * (++epoch)));
    // ---- winner (every workgroup, redundantly): thread w looks at workgroup w's candidate.  The winner's row is needed right
    //      after the decision; instead of a dependent second round trip every thread prefetches, together with the candidates,
    //      element (tid % 32) of the candidate rows of workgroups tid / 32, tid / 32 + 8, ... (G <= 64 covers 32768+ rows) and the
    //      decision then picks the winner's copy out of registers.
    constexpr int PF = 8;                                        // 8 x 8 = 64 workgroups prefetched
    T pf[PF];
    const bool pf_ok = G <= 8 * PF;
    if (pf_ok) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int64_t ww = (tid >> 5) + 8 * u;
            pf[u] = g.cand_data[((int64_t)par * G + (ww < G ? ww : G - 1)) * PB + (tid & 31)];
        }
    }
    const T dv_pref = g.diag_data[par * PB + (tid & 31)];
    {
        T v = T(-1); int64_t r = m; int w = 0;
        for (int64_t ww = tid; ww < G; ww += 256) {
            const T v2 = g.cand_val[par * G + ww]; const int64_t r2 = g.cand_row[par * G + ww];
            if (r2 < m && (v2 > v || (v2 == v && r2 < r))) { v = v2; r = r2; w = (int)ww; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T v2 = __shfl_xor(v, off); const int64_t r2 = __shfl_xor(r, off); const int w2 = __shfl_xor(w, off);
            if (r2 < m && (v2 > v || (v2 == v && r2 < r))) { v = v2; r = r2; w = w2; }
        }
        if (lane == 0) { s_wv[wid] = v; s_wr[wid] = r; s_ww[wid] = w; }
    }
    __syncthreads();
    T gv = s_wv[0]; p = s_wr[0]; wstar = s_ww[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (s_wr[w] < m && (s_wv[w] > gv || (s_wv[w] == gv && s_wr[w] < p))) { gv = s_wv[w]; p = s_wr[w]; wstar = s_ww[w]; }
    if (p >= m) p = j;                                           // empty / NaN column: no exchange
    if (pf_ok) {
        // the thread group (tid / 32) == wstar % 8 holds the winner's row in pf[wstar / 8]
        if ((tid >> 5) == (wstar & 7)) {
            T pv = pf[0];
#pragma unroll
            for (int u = 1; u < PF; ++u) pv = ((wstar >> 3) == u) ? pf[u] : pv;
            s_piv[tid & 31] = (p != j) ? pv : dv_pref;
        }
        if (tid < PB) s_drow[tid] = dv_pref;
    } else if (tid < PB) {
        const T* prow = g.cand_data + ((int64_t)par * G + wstar) * PB;   // contents of row p (becomes row j)
        s_drow[tid] = dv_pref;
        s_piv[tid] = (p != j) ? prow[tid] : dv_pref;
    }
    }
    if (me == 0 && tid == 0) g.ipiv[j] = p + 1;
    __syncthreads();
    LU_MARK(3)
    // ---- exchange rows j <-> p in the owners' registers, then eliminate
    const T piv = s_piv[C];
    const T rp = T(1) / piv;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        if (p != j) {
            if (st.gr[q] == j) {
#pragma unroll
                for (int c2 = 0; c2 < PB; ++c2) st.x[q][c2] = s_piv[c2];
            } else if (st.gr[q] == p) {
#pragma unroll
                for (int c2 = 0; c2 < PB; ++c2) st.x[q][c2] = s_drow[c2];
            }
        }
        if (piv != T(0) && st.gr[q] > j && st.gr[q] < m) {
            const T l = st.x[q][C] * rp;
            st.x[q][C] = l;
#pragma unroll
            for (int c2 = C + 1; c2 < PB; ++c2) st.x[q][c2] -= l * s_piv[c2];
        }
    }
    if (piv == T(0) && me == 0 && tid == 0 && *g.info == 0) *g.info = (int)(j + 1);
    __syncthreads();                                              // s_piv / s_drow / s_w* are rewritten next column
    LU_MARK(4)
}
template <typename T, int RPT, int C, bool TAG>
__device__ __forceinline__ void lu_reg_steps(const LuArgs<T>& g, LuRegState<T, RPT>& st, unsigned& epoch, T* s_wv, int64_t* s_wr, int* s_ww,
                                             T* s_piv, T* s_drow) {
    if constexpr (C < PB) {
        if (C < g.pb) {                                           // uniform: pb is a kernel argument
            lu_reg_step<T, RPT, C, TAG>(g, st, epoch, s_wv, s_wr, s_ww, s_piv, s_drow);
            lu_reg_steps<T, RPT, C + 1, TAG>(g, st, epoch, s_wv, s_wr, s_ww, s_piv, s_drow);
        }
    }
}
template <typename T, int RPT, bool TAG>
__global__ __launch_bounds__(256, (sizeof(T) == 8) ? 2 : 1) void getrf_panel_reg_kernel(LuArgs<T> g) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    __shared__ T s_wv[4];
    __shared__ int64_t s_wr[4];
    __shared__ int s_ww[4];
    __shared__ T s_piv[PB], s_drow[PB];
    const int tid = threadIdx.x;
    const int64_t me = blockIdx.x;
    const int pb = g.pb;
    const int64_t j0 = g.j0, m = g.m;
    const int64_t lo = j0 + me * (256 * RPT);
    LuRegState<T, RPT> st;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        st.gr[q] = lo + tid + 256 * q;
        const int64_t rr = st.gr[q] < m ? st.gr[q] : m - 1;           // clamped row: unconditional coalesced loads
#pragma unroll
        for (int c = 0; c < PB; ++c) st.x[q][c] = g.A[rr + (j0 + (c < pb ? c : pb - 1)) * g.lda];    // clamped row and column: unconditional, all in flight together
    }
    // (the selects come after ALL loads: written as `cond ? load : 0` per entry, hipcc sinks every load into its own branch with an
    // s_waitcnt vmcnt(0) behind it -- 128 dependent L2 round trips = ~30 us per panel launch)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
#pragma unroll
        for (int c = 0; c < PB; ++c) st.x[q][c] = (st.gr[q] < m && c < pb) ? st.x[q][c] : T(0);
    }
#ifdef RLHIP_LU_PROF
    for (int i = 0; i < 5; ++i) st.pf[i] = 0;
    st.pt = wall_clock64();
#endif
    unsigned epoch = 0;
    lu_reg_steps<T, RPT, 0, TAG>(g, st, epoch, s_wv, s_wr, s_ww, s_piv, s_drow);
#ifdef RLHIP_LU_PROF
    if (me == (int64_t)gridDim.x / 2 && tid == 0) for (int i = 0; i < 5; ++i) atomicAdd((unsigned long long*)(g.diag_data + 2 * PB) + i, (unsigned long long)st.pf[i]);
#endif
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        if (st.gr[q] < m) {
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (c < pb) g.A[st.gr[q] + (j0 + c) * g.lda] = st.x[q][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Panels taller than the device can hold resident (the register kernels keep every row of the panel in VGPRs: 256 rows x RPT per
// workgroup, every workgroup resident at once because they rendezvous every column): dgetf2 as three small launches per column --
// partial maxima, pivot + interchange, scale + rank-1 update of the panel -- with the panel living in L2 / Infinity Cache
// (200000 x 32 fp64 = 51 MB).  Any number of rows; ~12 us per column instead of ~4, taken only above the resident capacity
// (131072 / 262144 rows, below).  Same arithmetic and the same first-maximum pivot rule as the resident kernels and LAPACK.
template <typename T>
__global__ __launch_bounds__(256) void getf2_colmax_kernel(int64_t m, int64_t j, const T* __restrict__ A, int64_t lda, T* __restrict__ pval,
                                                           int64_t* __restrict__ prow) {
    __shared__ T s_v[4];
    __shared__ int64_t s_r[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    T bv = T(-1); int64_t br = m;
    for (int64_t i = j + (int64_t)blockIdx.x * 256 + tid; i < m; i += (int64_t)gridDim.x * 256) {
        const T v = fabs(A[i + j * lda]);
        if (v > bv) { bv = v; br = i; }                              // rows increase along the walk: ties keep the first
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T v2 = __shfl_xor(bv, off); const int64_t r2 = __shfl_xor(br, off);
        argmax_take(bv, br, v2, r2, m);
    }
    if (lane == 0) { s_v[wid] = bv; s_r[wid] = br; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) argmax_take(s_v[0], s_r[0], s_v[w], s_r[w], m);
        pval[blockIdx.x] = s_v[0]; prow[blockIdx.x] = s_r[0];
    }
}
// one workgroup: the column's pivot from the partial maxima, ipiv[j], the interchange of rows j and p inside the panel, info
template <typename T>
__global__ __launch_bounds__(256) void getf2_pivot_kernel(int64_t m, int64_t j0, int pb, int64_t j, int nparts, T* __restrict__ A, int64_t lda,
                                                          const T* __restrict__ pval, const int64_t* __restrict__ prow, int64_t* __restrict__ ipiv,
                                                          int* __restrict__ info) {
    __shared__ T s_v[4];
    __shared__ int64_t s_r[4];
    __shared__ int64_t s_p;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    T bv = T(-1); int64_t br = m;
    for (int i = tid; i < nparts; i += 256) argmax_take(bv, br, pval[i], prow[i], m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T v2 = __shfl_xor(bv, off); const int64_t r2 = __shfl_xor(br, off);
        argmax_take(bv, br, v2, r2, m);
    }
    if (lane == 0) { s_v[wid] = bv; s_r[wid] = br; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) argmax_take(s_v[0], s_r[0], s_v[w], s_r[w], m);
        const int64_t p = (s_r[0] < m) ? s_r[0] : j;
        s_p = p;
        ipiv[j] = p + 1;
        if (A[p + j * lda] == T(0) && *info == 0) *info = (int)(j + 1);
    }
    __syncthreads();
    const int64_t p = s_p;
    if (p != j && tid < pb) {
        const int64_t c = j0 + tid;
        const T a = A[j + c * lda], b = A[p + c * lda];
        A[j + c * lda] = b; A[p + c * lda] = a;
    }
}
// rows i > j: l = a_ij / pivot (left alone when the pivot is zero, as dgetf2), a_ic -= l * u_c for the panel columns right of j
template <typename T>
__global__ __launch_bounds__(256) void getf2_update_kernel(int64_t m, int64_t j0, int pb, int64_t j, T* __restrict__ A, int64_t lda) {
    __shared__ T s_u[PB];
    const int tid = threadIdx.x;
    const int nc = (int)(j0 + pb - 1 - j);                          // columns right of j inside the panel
    if (tid <= nc) s_u[tid] = A[j + (j + tid) * lda];                // s_u[0] = pivot
    __syncthreads();
    const T piv = s_u[0];
    if (piv == T(0)) return;
    const T rp = T(1) / piv;
    for (int64_t i = j + 1 + (int64_t)blockIdx.x * 256 + tid; i < m; i += (int64_t)gridDim.x * 256) {
        const T l = A[i + j * lda] * rp;
        A[i + j * lda] = l;
        for (int c = 1; c <= nc; ++c) A[i + (j + c) * lda] -= l * s_u[c];
    }
}

// dlaswp on column range [c_lo, c_hi): for j in [j0, j0+pb): swap rows j and ipiv[j]-1.
// Walking the pb swaps in order costs pb dependent load/store round trips per column (34 us per launch at pb = 32).  The swaps of a
// panel touch at most 2 pb rows, so every workgroup first composes them into ONE net permutation "row dst receives the old row
// src" (wave 0, in LDS, ~1 us).  One WAVE then moves one column: lane e gathers source value e and scatters it (two memory round trips
// for the whole column, 64 independent requests each; a thread per column walked its 64 strided elements alone: 32 us per launch on 8
// workgroups).
// SOLVE: the wave then forward-substitutes the column with the panel's unit lower triangle (U12 = L11^-1 A12): lane i holds x_i and row
// i of L11 in registers, step s broadcasts x_s by v_readlane -- the two steps of a panel that act on the columns to its right, in one launch
// (the thread-per-column solve took 14 us on its own).
template <typename T, bool SOLVE>
__global__ __launch_bounds__(256) void laswp_kernel(int64_t c_lo, int64_t c_hi, int64_t j0, int pb, T* A, int64_t lda,
                                                   const int64_t* __restrict__ ipiv) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    __shared__ int64_t s_pos[2 * PB], s_src[2 * PB];
    __shared__ int s_n;
    __shared__ T sL[SOLVE ? PB : 1][PB + 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (SOLVE) {
        for (int e = tid; e < PB * PB; e += 256) {
            const int i = e % PB, j = e / PB;
            sL[i][j] = (i < pb && j < pb && i > j) ? A[(j0 + i) + (j0 + j) * lda] : T(0);
        }
    }
    if (tid < 64) {                                    // one wave: wave-synchronous, no barriers inside
        const int e = tid;
        int64_t pos = -1;
        if (e < pb) pos = j0 + e;
        else if (e < 2 * pb) pos = ipiv[j0 + (e - pb)] - 1;
        s_pos[e < 2 * PB ? e : 0] = pos;
        __builtin_amdgcn_wave_barrier();
        int act = (e < 2 * pb) ? 1 : 0;
        for (int l = 0; l < e && act; ++l)
            if (l < 2 * pb && s_pos[l] == pos) act = 0;          // a later duplicate of a row already listed
        int64_t content = pos;                                     // which ORIGINAL row's value currently sits at `pos`
        for (int q = 0; q < pb; ++q) {
            const int64_t a = j0 + q, b = s_pos[pb + q];
            if (a == b) continue;                                  // uniform branch (LDS value)
            // the two holders exchange contents through the wave
            const int is_a = act && pos == a, is_b = act && pos == b;
            const unsigned long long ma = __ballot(is_a), mb = __ballot(is_b);
            const int la = __ffsll((long long)ma) - 1, lb = __ffsll((long long)mb) - 1;
            const int64_t ca = __shfl(content, la), cb = __shfl(content, lb);
            if (is_a) content = cb;
            if (is_b) content = ca;
        }
        const int moved = act && content != pos;
        const unsigned long long mm = __ballot(moved);
        const int slot = __popcll(mm & ((1ull << e) - 1ull));
        if (moved) { s_pos[slot] = pos; s_src[slot] = content; }   // compaction: slot <= e, and every lane has read s_pos already
        if (e == 0) s_n = __popcll(mm);
    }
    __syncthreads();
    const int nmv = s_n;
    if (!SOLVE && nmv == 0) return;
    const int64_t src = s_src[lane < nmv ? lane : 0], dst = s_pos[lane < nmv ? lane : 0];
    T lrow[SOLVE ? PB : 1];
    if constexpr (SOLVE) {
#pragma unroll
        for (int l = 0; l < PB; ++l) lrow[l] = sL[lane & (PB - 1)][l];
    }
    for (int64_t c = c_lo + (int64_t)blockIdx.x * 4 + wid; c < c_hi; c += (int64_t)gridDim.x * 4) {
        T* col = A + c * lda;
        if (nmv != 0) {
            const T v = col[src];                                      // every lane's load is back before the store instruction issues
            if (lane < nmv) col[dst] = v;
        }
        if constexpr (SOLVE) {
            T* cj = col + j0;
            T x = cj[(lane < pb) ? lane : (pb - 1)];                   // (after the wave's own stores, in program order)
            if (lane >= pb) x = T(0);
#pragma unroll
            for (int st = 0; st < PB; ++st) {
                if (st < pb) {                                         // uniform
                    T xs;
                    if constexpr (sizeof(T) == 4) xs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)x), st));
                    else xs = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint((double)x), st), __builtin_amdgcn_readlane(__double2loint((double)x), st));
                    if (lane > st) x -= lrow[st] * xs;                 // (lrow is zero on and above the diagonal and for lanes >= pb)
                }
            }
            if (lane < pb) cj[lane] = x;
        }
    }
}

// U12 = L11^-1 A12 (unit lower, jb x jb at A[j0,j0]) on the columns [c_lo, c_hi); one column per thread
template <typename T>
__global__ __launch_bounds__(256) void unit_lower_solve_kernel(int64_t c_lo, int64_t c_hi, int64_t j0, int jb, T* __restrict__ A, int64_t lda) {
    __shared__ T sL[PB][PB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < PB * PB; e += 256) {
        int i = e % PB, j = e / PB;
        sL[i][j] = (i < jb && j < jb && i > j) ? A[(j0 + i) + (j0 + j) * lda] : T(0);
    }
    __syncthreads();
    int64_t c = c_lo + (int64_t)blockIdx.x * 256 + tid;
    if (c >= c_hi) return;
    T x[PB];
    T* col = A + j0 + c * lda;
#pragma unroll
    for (int i = 0; i < PB; ++i) { const T t = col[(i < jb) ? i : (jb - 1)]; x[i] = (i < jb) ? t : T(0); }   // clamped, not branched
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        if (i < jb) {
            T s = x[i];
#pragma unroll
            for (int l = 0; l < PB; ++l)
                if (l < i) s -= sL[i][l] * x[l];
            x[i] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
        if (i < jb) col[i] = x[i];
}

// BQRRP's conversion of LU row pivots into a QRCP permutation (rl_bqrrp.hh:345-350; LUQRCP_piv_process_gpu_global,
// rl_cuda_kernels.cuh:203-220): J = iota(1..cols); for i < min(sd, cols): swap(J[ipiv[i]-1], J[i]).  Serial, integer-exact.
__global__ void luqrcp_piv_kernel(int64_t sd, int64_t cols, const int64_t* __restrict__ ipiv, int64_t* __restrict__ J) {
    for (int64_t i = threadIdx.x; i < cols; i += blockDim.x) J[i] = i + 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t lim = sd < cols ? sd : cols;
        for (int64_t i = 0; i < lim; ++i) {
            const int64_t a = ipiv[i] - 1;
            const int64_t t = J[a]; J[a] = J[i]; J[i] = t;
        }
    }
}
// The same conversion with the serial part in LDS (the chain of `lim` dependent global read-modify-writes above costs 0.25 us a step:
// 0.56 ms at lim = 2048): positions below lim live in an LDS array, the at most lim touched positions beyond it in an LDS hash table
// (open addressing, 2 * LQ_MAX slots); one thread walks the swaps in ~20 ns a step, the workgroup writes the touched entries back.
constexpr int LQ_MAX = 2048;
__global__ __launch_bounds__(256) void luqrcp_iota_kernel(int64_t cols, int64_t* __restrict__ J) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cols) J[i] = i + 1;
}
__global__ __launch_bounds__(256) void luqrcp_piv_lds_kernel(int64_t sd, int64_t cols, const int64_t* __restrict__ ipiv, int64_t* __restrict__ J) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    __shared__ int s_low[LQ_MAX];                 // J[i] - 1 for i < lim
    __shared__ int s_piv[LQ_MAX];                 // ipiv[i] - 1
    __shared__ int s_key[2 * LQ_MAX], s_val[2 * LQ_MAX];
    const int lim = (int)(sd < cols ? sd : cols);
    // (J = iota(1 .. cols) is written by luqrcp_iota_kernel in front of this launch: one workgroup filling 65536 entries took 0.3 of this kernel's 0.39 ms)
    for (int i = threadIdx.x; i < lim; i += blockDim.x) { s_low[i] = i; s_piv[i] = (int)(ipiv[i] - 1); }
    for (int i = threadIdx.x; i < 2 * LQ_MAX; i += blockDim.x) s_key[i] = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < lim; ++i) {
            const int a = s_piv[i];
            if (a < lim) { const int t = s_low[a]; s_low[a] = s_low[i]; s_low[i] = t; }
            else {
                unsigned h = ((unsigned)a * 2654435761u) >> 20;           // 12 bits
                while (s_key[h] != -1 && s_key[h] != a) h = (h + 1) & (2 * LQ_MAX - 1);
                const int t = (s_key[h] == a) ? s_val[h] : a;             // an untouched position still holds its own index
                s_key[h] = a; s_val[h] = s_low[i]; s_low[i] = t;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < lim; i += blockDim.x) J[i] = (int64_t)s_low[i] + 1;
    for (int i = threadIdx.x; i < 2 * LQ_MAX; i += blockDim.x)
        if (s_key[i] >= 0) J[s_key[i]] = (int64_t)s_val[i] + 1;
}

__global__ void lu_zero_kernel(unsigned* bar, int* info, int zero_info) { *bar = 0; if (zero_info) *info = 0; }

}  // namespace

namespace rlhip {

// grid of laswp_kernel: one column per wave, four per workgroup; very wide ranges loop
static unsigned laswp_grid(int64_t ncols) {
    int64_t g = (ncols + 3) / 4;
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}


template <typename T>
int getrf(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv_dev, int* info_host, int pivots_only) {
    if (info_host) *info_host = 0;
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (lda < (m > 1 ? m : 1)) return -5;
    const int64_t mn = m < n ? m : n;
    if (mn == 0) return 0;
    const int num_cu = c->num_cu;
    RLHIP_FUNC_LDS(c, getrf_panel_kernel<T>, 128 * 1024);
    size_t mark = rlhip_ws_mark(c);
    const int64_t Gmax = num_cu;
    LuArgs<T> g;
    g.m = m; g.n = n; g.A = A; g.lda = lda; g.ipiv = ipiv_dev;
    g.cand_val = ws_alloc<T>(c, 2 * Gmax); g.cand_row = ws_alloc<int64_t>(c, 2 * Gmax);
    g.cand_data = ws_alloc<T>(c, (size_t)2 * Gmax * PB); g.diag_data = ws_alloc<T>(c, 2 * PB + 64);
    g.bar = ws_alloc<unsigned>(c, 4); g.info = (int*)ws_alloc<int>(c, 4);
    const bool use_tag = m < ((int64_t)1 << 31);
    constexpr size_t TW = sizeof(T) / 4;
    // (the general register kernel may run more workgroups than CUs: the word buffer is sized -- and cleared -- for the largest grid of this call)
    const int64_t Gtw = std::max<int64_t>(Gmax, (m + 511) / 512 + 1);
    const size_t tw_words = 2 * (size_t)(TW * Gtw + Gtw + TW * Gtw * PB + TW * PB);
    g.tw = use_tag ? (unsigned long long*)rlhip_xchg_buffer(c, tw_words * sizeof(unsigned long long)) : nullptr;   // uncached exchange memory of the context
    g.tag_base = 0;
    // tags: (panel index + 1) * 64 + column.  The word buffer is cleared at the start of every call and belongs to this call's workspace, so
    // a tag is unique where it can be seen and never equals the cleared pattern (m < 2^31 keeps 2 * j0 + 64 inside 32 bits)
    if (use_tag) {
        if (!g.tw) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        RLHIP_CHECK(hipMemsetAsync(g.tw, 0, tw_words * sizeof(unsigned long long), c->stream));
    }
    if (!g.cand_val || !g.cand_row || !g.cand_data || !g.diag_data || !g.bar || !g.info) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    // Two-level blocking: the 32-column panel steps then update only the columns of their own
    // OUTER block; everything to the right of it gets the block's interchanges, one block forward substitution and ONE rank-nbo GEMM
    // when the block is finished.  Pays for very tall matrices only (numbers below); round 1 measured it slower everywhere because the
    // panel kernel, not the updates, dominated then.
    // default: with the faster fp32 panel step the HBM-bound K = 32 updates of a very tall matrix are worth saving again -- 65536 x 2048
    // fp32: 32.1 ms (outer = panel), 27.0 (64), 25.0 (128), 25.1 (256); 32768 rows: 20.7 / 20.4 / 20.5 / 21.5; 8192 rows: 16.4 / 17.1 / 17.9
    const int64_t nbo = (m >= 49152 && n >= 256) ? 128 : PB;
#ifdef RLHIP_LU_PROF
    hipMemsetAsync(g.diag_data, 0, (2 * PB + 64) * sizeof(T), c->stream);
#endif
    for (int64_t J0 = 0; J0 < mn; J0 += nbo) {
    const int64_t Jend = (J0 + nbo < mn) ? J0 + nbo : mn;      // columns factored by this outer block
    const int64_t Cin = (J0 + nbo < n) ? J0 + nbo : n;         // columns the panel steps keep up to date
    for (int64_t j0 = J0; j0 < Jend; j0 += PB) {
        const int pb = (int)((Jend - j0 < PB) ? (Jend - j0) : PB);
        const int64_t rows = m - j0;
        // rows per workgroup: at least 64, LDS piece pb*rpw*sizeof(T) <= 96 KiB
        int64_t rpw = (rows + Gmax - 1) / Gmax;
        if (rpw < 256) rpw = 256;   // fewer, fatter workgroups: the per-column rendezvous and winner search shrink with G
        const int64_t rpw_max = (96 * 1024) / (PB * (int64_t)sizeof(T));
        if (rpw > rpw_max && !(use_tag && rows >= 1024)) { rlhip_ws_release(c, mark); return -2; }   // LDS variant only: > num_cu * 384 rows (fp64)
        int64_t G = (rows + rpw - 1) / rpw;
        g.j0 = j0; g.pb = pb; g.rpw = rpw;
        // (the tagged-word kernels never touch the barrier counter: only the first panel of a call needs the launch, for `info`)
        if (j0 == 0 || !(use_tag && rows >= 1024))
            hipLaunchKernelGGL(lu_zero_kernel, dim3(1), dim3(1), 0, c->stream, g.bar, g.info, j0 == 0 ? 1 : 0);
        constexpr int RPT_BIG = (sizeof(T) == 4) ? 4 : 2;             // 128 VGPRs of panel per thread either way
        // resident capacity of the general register kernel (every workgroup spins on the others: all of them must be on the device at once)
        static int64_t reg_cap[64] = {};
        if (!reg_cap[c->device & 63]) {
            int nb = 0;
            RLHIP_CHECK(hipSetDevice(c->device));
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)getrf_panel_reg_kernel<T, RPT_BIG, true>, 256, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
            reg_cap[c->device & 63] = (int64_t)nb * num_cu;
        }
        const int64_t G_reg = (rows + 256 * RPT_BIG - 1) / (256 * RPT_BIG);
        if (G_reg > reg_cap[c->device & 63] && G_reg > 64) {
            // taller than the resident kernels can hold (fp64: 262144 rows, fp32: 262144): column-at-a-time launches on the L2-resident panel
            const int nparts = (int)((rows / 2048 < 1) ? 1 : (rows / 2048 > 1024 ? 1024 : rows / 2048));
            T* pval = ws_alloc<T>(c, 1024); int64_t* prow = ws_alloc<int64_t>(c, 1024);
            if (!pval || !prow) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
            for (int64_t j = j0; j < j0 + pb; ++j) {
                hipLaunchKernelGGL(getf2_colmax_kernel<T>, dim3((unsigned)nparts), dim3(256), 0, c->stream, m, j, A, lda, pval, prow);
                hipLaunchKernelGGL(getf2_pivot_kernel<T>, dim3(1), dim3(256), 0, c->stream, m, j0, pb, j, nparts, A, lda, pval, prow, ipiv_dev, g.info);
                if (j + 1 < m) hipLaunchKernelGGL(getf2_update_kernel<T>, dim3((unsigned)nparts), dim3(256), 0, c->stream, m, j0, pb, j, A, lda);
            }
            c->path_count[7]++;
        } else
        if (use_tag && rows >= 1024) {
            G = (rows + 256 * RPT_BIG - 1) / (256 * RPT_BIG);
            g.tag_base = (unsigned)(j0 / PB + 1) * 64u;
            bool launched = false;
            if constexpr (sizeof(T) == 4) {
                if (G <= 64 && m < ((int64_t)1 << 31)) {        // up to 65536 rows below the diagonal: the step of lu_f32_step
                    rlhip_lu::launch_getrf_panel_f32(g, (unsigned)G, c->stream);
                    launched = true;
                }
            }
            if constexpr (sizeof(T) == 8) {
                if (G <= 64 && m < ((int64_t)1 << 31)) {        // up to 32768 rows below the diagonal: the step of lu_f64_step
                    rlhip_lu::launch_getrf_panel_f64(g, (unsigned)G, c->stream);
                    launched = true;
                }
            }
            if (!launched)
            hipLaunchKernelGGL((getrf_panel_reg_kernel<T, RPT_BIG, true>), dim3((unsigned)G), dim3(256), 0, c->stream, g);
        } else   // short panels (< 1024 rows, at most 4 workgroups): the LDS-resident kernel; one register-kernel instantiation per type keeps the
                 // build of this file (32 unrolled column steps) within minutes
            hipLaunchKernelGGL(getrf_panel_kernel<T>, dim3((unsigned)G), dim3(256), (size_t)pb * rpw * sizeof(T), c->stream, g);
        RLHIP_LAUNCH_CHECK();
        // row interchanges left of the panel.  The columns of this outer block feed its closing GEMM, so they always follow; the
        // columns of earlier blocks only keep L consistent and never feed a later pivot decision
        const int64_t left_lo = pivots_only ? J0 : 0;
        if (j0 > left_lo)
            hipLaunchKernelGGL((laswp_kernel<T, false>), dim3(laswp_grid(j0 - left_lo)), dim3(256), 0, c->stream, left_lo, j0, j0, pb, A, lda, ipiv_dev);
        const int64_t rest = Cin - j0 - pb;
        if (rest > 0) {
            // interchanges + U12 = L11^-1 A12 of the columns right of the panel, one launch
            hipLaunchKernelGGL((laswp_kernel<T, true>), dim3(laswp_grid(rest)), dim3(256), 0, c->stream, j0 + pb, Cin, j0, pb, A, lda, ipiv_dev);
            RLHIP_LAUNCH_CHECK();
            const int64_t mrest = m - j0 - pb;
            if (mrest > 0) {
                int rc = gemm_impl<T>(c, 0, 0, mrest, rest, pb, T(-1), A + (j0 + pb) + j0 * lda, lda, A + j0 + (j0 + pb) * lda, lda, T(1),
                                      A + (j0 + pb) + (j0 + pb) * lda, lda, 0);
                if (rc) { rlhip_ws_release(c, mark); return rc; }
            }
        }
    }
    // ---- the columns right of the outer block: interchanges, U12 = L11^-1 A12 by 32-row blocks, A22 -= L21 U12
    const int64_t right = n - Cin;
    if (right > 0) {
        const unsigned gr = laswp_grid(right);
        const bool one_panel = (Jend - J0 <= PB);                  // the outer block is a single panel: interchanges + its forward substitution in one launch
        for (int64_t q0 = J0; q0 < Jend; q0 += PB) {
            const int cnt = (int)((Jend - q0 < PB) ? (Jend - q0) : PB);
            if (one_panel) hipLaunchKernelGGL((laswp_kernel<T, true>), dim3(gr), dim3(256), 0, c->stream, Cin, n, q0, cnt, A, lda, ipiv_dev);
            else hipLaunchKernelGGL((laswp_kernel<T, false>), dim3(gr), dim3(256), 0, c->stream, Cin, n, q0, cnt, A, lda, ipiv_dev);
        }
        const unsigned gs = (unsigned)((right + 255) / 256);
        for (int64_t s0 = J0; s0 < Jend && !one_panel; s0 += PB) {
            const int sb = (int)((Jend - s0 < PB) ? (Jend - s0) : PB);
            hipLaunchKernelGGL(unit_lower_solve_kernel<T>, dim3(gs), dim3(256), 0, c->stream, Cin, n, s0, sb, A, lda);
            RLHIP_LAUNCH_CHECK();
            const int64_t below = Jend - (s0 + sb);
            if (below > 0) {
                int rc = gemm_impl<T>(c, 0, 0, below, right, sb, T(-1), A + (s0 + sb) + s0 * lda, lda, A + s0 + Cin * lda, lda, T(1),
                                      A + (s0 + sb) + Cin * lda, lda, 0);
                if (rc) { rlhip_ws_release(c, mark); return rc; }
            }
        }
        RLHIP_LAUNCH_CHECK();
        const int64_t mrest = m - Jend;
        if (mrest > 0) {
            int rc = gemm_impl<T>(c, 0, 0, mrest, right, Jend - J0, T(-1), A + Jend + J0 * lda, lda, A + J0 + Cin * lda, lda, T(1),
                                  A + Jend + Cin * lda, lda, 0);
            if (rc) { rlhip_ws_release(c, mark); return rc; }
        }
    }
    }
#ifdef RLHIP_LU_PROF
    {
        unsigned long long pf[5];
        rlhip_stream_sync(c);
        hipMemcpy(pf, (unsigned long long*)(g.diag_data + 2 * PB), sizeof(pf), hipMemcpyDeviceToHost);
        const double cols = (double)mn;
        fprintf(stderr, "[lu prof %ld x %ld] us per column: candidate %.2f  publish %.2f  exchange %.2f  decision %.2f  eliminate %.2f\n", (long)m, (long)n,
                pf[0] / 100.0 / cols, pf[1] / 100.0 / cols, pf[2] / 100.0 / cols, pf[3] / 100.0 / cols, pf[4] / 100.0 / cols);
    }
#endif
    if (info_host) {
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 56, g.info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        *info_host = *(int*)(c->h_mail + 56);
        if (*info_host < 0) { rlhip_ws_release(c, mark); return -9; }   // the flag-less exchange timed out (bounded so that a lost word cannot hang the device)
    }
    rlhip_ws_release(c, mark);
    return 0;
}

// lapack::laswp(n, A, lda, k1, k2, ipiv, incx = 1) with 1-based k1..k2 (rl_orth.hh:226)
template <typename T>
int laswp(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int64_t k1, int64_t k2, const int64_t* ipiv_dev) {
    if (n <= 0 || k2 < k1) return 0;
    for (int64_t q0 = k1 - 1; q0 < k2; q0 += PB) {           // the kernel composes up to PB interchanges at a time, in order
        const int cnt = (int)((k2 - q0 < PB) ? (k2 - q0) : PB);
        hipLaunchKernelGGL((laswp_kernel<T, false>), dim3(laswp_grid(n)), dim3(256), 0, c->stream, (int64_t)0, n, q0, cnt, A, lda, ipiv_dev);
    }
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template int laswp<double>(rlhip_ctx*, int64_t, double*, int64_t, int64_t, int64_t, const int64_t*);
template int laswp<float>(rlhip_ctx*, int64_t, float*, int64_t, int64_t, int64_t, const int64_t*);

int luqrcp_piv(rlhip_ctx* c, int64_t sd, int64_t cols, const int64_t* ipiv_dev, int64_t* J_dev) {
    if (cols <= 0) return 0;
    const int64_t lim = sd < cols ? sd : cols;
    if (lim <= LQ_MAX && cols < ((int64_t)1 << 31)) {
        hipLaunchKernelGGL(luqrcp_iota_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, c->stream, cols, J_dev);
        hipLaunchKernelGGL(luqrcp_piv_lds_kernel, dim3(1), dim3(256), 0, c->stream, sd, cols, ipiv_dev, J_dev);
    }
    else hipLaunchKernelGGL(luqrcp_piv_kernel, dim3(1), dim3(256), 0, c->stream, sd, cols, ipiv_dev, J_dev);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

template int getrf<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, int64_t*, int*, int);
template int getrf<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, int64_t*, int*, int);

}  // namespace rlhip
