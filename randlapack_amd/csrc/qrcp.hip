// Column-pivoted Householder QR of the (small, d x n) sketch on the device: lapack::geqp3 at
// RandLAPACK/drivers/rl_cqrrpt.hh:247 (and rl_bqrrp.hh qrcp_wide = geqp3 option).  This call DEFINES the pivots,
// so the arithmetic follows LAPACK's dlaqp2 step by step (first-maximum pivot search over the partial column
// norms, dlarfg reflector, the |A(k,j)|/vn1(j) norm down-date with the sqrt(eps) recomputation safeguard);
// given the same sketch the pivot vector equals LAPACK's except on exact near-ties of partial norms.
//
// Execution model: ONE persistent launch.  Column position j lives with workgroup j % G (cyclic, so the load
// stays balanced as the factorization advances) and is kept in that workgroup's LDS when it fits.  Per step there
// is ONE grid-wide rendezvous: before it every workgroup publishes its best local candidate (norm, position)
// TOGETHER WITH the finished Householder column it would produce (dlarfg applied speculatively to its own
// candidate -- an m-length pass, cheaper than a second rendezvous), and the owner of position k publishes the
// column that will move away; after it everybody knows the winner, reads the winner's finished column, installs
// the two moved columns, applies H to the columns it owns (one wavefront per column, shuffle reductions, no block
// barrier) and down-dates their norms.  Published buffers are double-buffered by step parity (a workgroup can be
// at most one rendezvous ahead).  Measured 1280 x 1024: two-rendezvous/global-memory version 61 ms -> LDS
// columns 36 ms -> single rendezvous + fence-free publication: see DESIGN.md.
#include "rlhip_internal.h"
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <limits>

namespace {

template <typename T>
struct QrcpArgs {
    int64_t m, n;
    T* A; int64_t lda;
    int64_t* jpvt;            // device, 1-based on exit; entry values are ignored (no "fixed" columns)
    T* tau;
    T* cand_val; int64_t* cand_pos; T* cand_tau;   // 2 x G each (step parity)
    T* slot;                  // 2 x G x m : each workgroup's speculative "finished pivot column"
    T* kcol;                  // 2 x m     : column currently at position k (moves to position p)
    unsigned* bar;            // barrier counter (zeroed by the host)
    T tol3z;
    int use_lds;              // owned columns live in LDS for the whole factorization
    int pivot;                // 0: plain Householder QR (geqr2 order), jpvt untouched
    int64_t max_steps;        // number of columns to factor (< min(m,n): partial factorization, HQRRP's sketch step)
    int hq_formula;           // 1: norm down-date written as (1+t)(1-t) (rl_hqrrp.hh:373), 0: dlaqp2's 1 - t^2
    int v_in_lds;             // the step's reflector is staged in LDS (m fits) or read from its published slot (tall inputs)
};

// ---- cross-workgroup traffic uses agent-scope relaxed atomics on 8-byte granules (sc1 write-through stores /
//      L1-bypassing loads): no cache-maintenance fences are needed around the rendezvous (guide section 6, G16:
//      "8-B agent atomics both sides"), which keeps a step's single grid barrier at a few microseconds.
template <typename T>
__device__ __forceinline__ void pub_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T pub_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's published stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        // one L1 invalidate per step: everything published before the rendezvous was stored write-through (sc1),
        // so after this acquire it can be read with ordinary wide loads
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// the same sum with DPP row rotations (VALU speed) + four readlanes instead of six dependent cross-lane shuffles through the LDS
// crossbar: the reductions sit on the critical path of every column step of the barrier-free kernel below
__device__ __forceinline__ double dpp_ror_add_f64(double v, const int n) {
    int lo = __double2loint(v), hi = __double2hiint(v), lo2, hi2;
    switch (n) {
        case 8: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false); break;
        case 4: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xF, 0xF, false); break;
        case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xF, 0xF, false); break;
        default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, false); break;
    }
    return v + __hiloint2double(hi2, lo2);
}
template <typename T>
__device__ __forceinline__ T wave_sum_fast(T x) {
    double v = (double)x;                                       // (fp32 inputs: the sum itself in double, rounded once)
    v = dpp_ror_add_f64(v, 8); v = dpp_ror_add_f64(v, 4); v = dpp_ror_add_f64(v, 2); v = dpp_ror_add_f64(v, 1);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) r += __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
    return (T)r;
}

// (USE_LDS / V_IN_LDS are template parameters for the reason given at qr_pipe_kernel: a run-time choice between an LDS and a global pointer
//  is a generic pointer, and every access through it a flat_ instruction)
template <typename T, bool USE_LDS, bool V_IN_LDS>
__global__ __launch_bounds__(256) void qrcp_kernel(QrcpArgs<T> g) {
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t G = gridDim.x, me = blockIdx.x;
    const int64_t m = g.m, n = g.n;
    const int64_t kmin = m < n ? m : n;
    const int64_t kmax = (g.max_steps >= 0 && g.max_steps < kmin) ? g.max_steps : kmin;
    __shared__ T s_val[4];
    unsigned epoch = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_smem[];
    const int64_t cpw = (n + G - 1) / G;                 // owned positions: j = me + G*s, s < cpw
    T* l_vn1 = reinterpret_cast<T*>(qr_smem);            // partial norms of the owned positions (local to the owner)
    T* l_vn2 = l_vn1 + cpw;
    T* l_v = l_vn2 + cpw;                                // the step's finished pivot column (m)
    T* lds_cols = l_v + (V_IN_LDS ? m : 0);
    __shared__ T s_cval[256];
    __shared__ int64_t s_cpos[256];
    __shared__ int s_cw[256];
    // column position j -> storage (LDS slot j/G of the owner, or the matrix itself)
    auto colptr = [&](int64_t j) {
        if constexpr (USE_LDS) return lds_cols + (j / G) * m;
        else return g.A + j * g.lda;
    };
    if constexpr (USE_LDS) {
        for (int64_t j = me; j < n; j += G) {
            T* dst = lds_cols + (j / G) * m;
            const T* src = g.A + j * g.lda;
            for (int64_t i = tid; i < m; i += 256) dst[i] = src[i];
        }
        __syncthreads();
    }
    // ---- initial column norms + jpvt (columns me, me+G, ...; one wave per column).  vn1/vn2/jpvt entries of a
    //      position are only ever written by that position's owner, except the swap at the pivot step.
    for (int64_t j = me + G * wid; j < n; j += 4 * G) {
        const T* col = colptr(j);
        T ss = 0;
        for (int64_t i = lane; i < m; i += 64) ss += col[i] * col[i];
        ss = wave_sum(ss);
        if (lane == 0) {
            T nr = sqrt(ss);
            l_vn1[j / G] = nr; l_vn2[j / G] = nr;
            if (g.pivot) __hip_atomic_store(g.jpvt + j, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();

    for (int64_t k = 0; k < kmax; ++k) {
        const int par = (int)(k & 1);
        T* my_slot = g.slot + ((int64_t)par * G + me) * m;
        T* kcol = g.kcol + (int64_t)par * (m + 2);
        // ---- A. local candidate over owned positions >= k (first maximum) and its SPECULATIVE reflector
        T best = T(-1); int64_t bpos = n;
        if (g.pivot) {
            for (int64_t j = me; j < n; j += G) {
                if (j < k) continue;
                T v = l_vn1[j / G];
                if (v > best) { best = v; bpos = j; }   // increasing j: strict > keeps the first maximum
            }
        } else if (me == k % G) {
            best = 0; bpos = k;
        }
        T my_tau = 0;
        if (bpos < n) {
            const T* col = colptr(bpos);
            T ss = 0;
            for (int64_t i = k + 1 + tid; i < m; i += 256) ss += col[i] * col[i];
            ss = wave_sum(ss);
            __syncthreads();
            if (lane == 0) s_val[wid] = ss;
            __syncthreads();
            const T xnorm = sqrt(s_val[0] + s_val[1] + s_val[2] + s_val[3]);
            const T alpha = col[k];
            T beta = alpha, scale = 0;
            if (xnorm != T(0)) {                                    // dlarfg (without the safmin rescaling loop)
                beta = -copysign(hypot(alpha, xnorm), alpha);
                my_tau = (beta - alpha) / beta;
                scale = T(1) / (alpha - beta);
            }
            for (int64_t i = tid; i < m; i += 256) {
                T v = col[i];
                if (i == k) v = beta; else if (i > k) v *= scale;
                pub_store(my_slot + i, v);
            }
        }
        if (tid == 0) {
            pub_store(g.cand_val + par * G + me, best);
            __hip_atomic_store(g.cand_pos + par * G + me, bpos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pub_store(g.cand_tau + par * G + me, my_tau);
        }
        const int64_t own_k = k % G;
        if (me == own_k) {
            const T* col = colptr(k);
            for (int64_t i = tid; i < m; i += 256) pub_store(kcol + i, col[i]);
            if (tid == 0) { pub_store(kcol + m, l_vn1[k / G]); pub_store(kcol + m + 1, l_vn2[k / G]); }
        }
        grid_barrier(g.bar, (unsigned)(G * (++epoch)));                                           // the step's rendezvous
        // ---- B. global pivot (every workgroup, redundantly): max norm, ties -> smallest position
        //      (thread w inspects workgroup w's candidate; tree reduction in LDS -- a serial scan of G atomic loads
        //      by every thread cost ~35 us per step)
        {
            T v = T(-1); int64_t q = n; int w = (int)own_k;
            for (int64_t ww = tid; ww < G; ww += 256) {
                T v2 = g.cand_val[par * G + ww];
                int64_t q2 = g.cand_pos[par * G + ww];
                if (q2 < n && (v2 > v || (v2 == v && q2 < q))) { v = v2; q = q2; w = (int)ww; }
            }
            s_cval[tid] = v; s_cpos[tid] = q; s_cw[tid] = w;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if (tid < st) {
                    T v2 = s_cval[tid + st]; int64_t q2 = s_cpos[tid + st];
                    if (q2 < n && (v2 > s_cval[tid] || (v2 == s_cval[tid] && q2 < s_cpos[tid]))) {
                        s_cval[tid] = v2; s_cpos[tid] = q2; s_cw[tid] = s_cw[tid + st];
                    }
                }
                __syncthreads();
            }
        }
        int64_t p = s_cpos[0]; int64_t wstar = s_cw[0];
        if (p >= n) { p = k; wstar = own_k; }   // nothing comparable left (NaNs): natural order
        const int64_t own_p = p % G;
        const T tauk = g.cand_tau[par * G + wstar];
        const T* vcol = g.slot + ((int64_t)par * G + wstar) * m;    // finished column: R above k, beta at k, v below
        // ---- C. install the moved columns, swap bookkeeping
        if constexpr (V_IN_LDS) {
            for (int64_t i = tid; i < m; i += 256) l_v[i] = vcol[i];
            __syncthreads();
        }
        auto vv_of = [&]() {                          // tall inputs: straight from the published slot (L2)
            if constexpr (V_IN_LDS) return (const T*)l_v;
            else return vcol;
        };
        const auto vv = vv_of();
        if (me == own_k) {
            T* col = colptr(k);
            for (int64_t i = tid; i < m; i += 256) col[i] = vv[i];
            if (tid == 0) g.tau[k] = tauk;
        }
        if (p != k && me == own_p) {
            T* col = colptr(p);
            for (int64_t i = tid; i < m; i += 256) col[i] = kcol[i];
            if (tid == 0) {
                l_vn1[p / G] = kcol[m];
                l_vn2[p / G] = kcol[m + 1];
                int64_t jp = __hip_atomic_load(g.jpvt + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int64_t jk = __hip_atomic_load(g.jpvt + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g.jpvt + p, jk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g.jpvt + k, jp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        // ---- D. apply H = I - tau v v^T (v_k = 1) to owned columns j > k, one wave per column; down-date norms
        for (int64_t j = me + G * wid; j < n; j += 4 * G) {
            if (j <= k) continue;
            T* col = colptr(j);
            if (tauk != T(0)) {
                // eight 64-row slabs per trip: eight independent loads in flight per lane (columns that live in HBM / L2 are latency-bound
                // otherwise: 1.6 TB/s at 16384^2) and eight partial sums
                T w8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int64_t i = k + lane;
                for (; i + 448 < m; i += 512) {
                    T cv[8], vx[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { cv[u] = col[i + 64 * u]; vx[u] = vv[i + 64 * u]; }
                    if (i == k) vx[0] = T(1);
#pragma unroll
                    for (int u = 0; u < 8; ++u) w8[u] += vx[u] * cv[u];
                }
                for (; i < m; i += 64) w8[0] += ((i == k) ? T(1) : vv[i]) * col[i];
                T w = wave_sum(((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7]))) * tauk;
                i = k + lane;
                for (; i + 448 < m; i += 512) {
                    T cv[8], vx[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { cv[u] = col[i + 64 * u]; vx[u] = vv[i + 64 * u]; }
                    if (i == k) vx[0] = T(1);
#pragma unroll
                    for (int u = 0; u < 8; ++u) col[i + 64 * u] = cv[u] - w * vx[u];
                }
                for (; i < m; i += 64) col[i] -= w * ((i == k) ? T(1) : vv[i]);
            }
            if (!g.pivot) continue;
            // dlaqp2: vn1(j) *= sqrt(max(0, 1 - (|A(k,j)|/vn1(j))^2)), recomputed when cancellation is detected
            T v1 = l_vn1[j / G];
            if (v1 != T(0)) {
                T akj = fabs(col[k]);
                T r = akj / v1;
                T temp = g.hq_formula ? (T(1) + r) * (T(1) - r) : T(1) - r * r;
                temp = temp > T(0) ? temp : T(0);
                T q = v1 / l_vn2[j / G];
                T temp2 = temp * q * q;
                if (temp2 <= g.tol3z) {
                    T ss = 0;
                    for (int64_t i = k + 1 + lane; i < m; i += 64) ss += col[i] * col[i];
                    ss = wave_sum(ss);
                    v1 = sqrt(ss);
                    if (lane == 0) { l_vn1[j / G] = v1; l_vn2[j / G] = v1; }
                } else {
                    if (lane == 0) l_vn1[j / G] = v1 * sqrt(temp);
                }
            }
        }
        __syncthreads();
    }
    if constexpr (USE_LDS) {
        __syncthreads();
        for (int64_t j = me; j < n; j += G) {
            const T* src = lds_cols + (j / G) * m;
            T* dst = g.A + j * g.lda;
            for (int64_t i = tid; i < m; i += 256) dst[i] = src[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same factorization with a FLAG-LESS exchange (the scheme of the LU panel kernel, lu.hip): everything that crosses workgroups
// travels as 8-byte words {step tag : 32-bit payload} written with write-through stores and simply re-read until they carry the
// step's tag.  No store drain, no barrier counter whose arrivals serialise in L2, no acquire fence, no speculative reflectors:
//   1. every workgroup publishes (best partial norm, position) of its owned columns; the owner of position k publishes that column;
//      a workgroup whose candidate would have ranked among the first few of the previous step's records finishes it (dlarfg) right
//      away and publishes the finished column in its own slot while the records travel;
//   2. every workgroup reads all G records (thread w <- workgroup w) and reduces them: the winner is known;
//   3. the WINNER forms the reflector of its column and publishes column + tau + jpvt entry -- unless it already has (step 1);
//   4. everybody else spins on the winner's slot (one batch of loads per thread) and stages the column in LDS;
//   5. owners install the two moved columns; 6. everybody applies H to the columns it owns and down-dates their norms.
// One store->load hand-off per step when the guess of step 1 holds (two otherwise) instead of drain + atomic arrival + poll + read.
// A hand-off through memory costs ~3 us on this part (8 XCDs: agent-scope data has to pass the memory side), which is the floor of
// any one-decision-per-column scheme; measured 1280 x 1024 fp64: rendezvous kernel 15.9, this one see DESIGN.md.  The statements
// (candidate order, dlarfg, dlaqp2 down-date) are those of the kernel above; only the dot products of step 6 are summed in four
// partial sums.  The speculation decides WHEN the winner's dlarfg runs, never what it computes.  jpvt entries live with their
// columns in LDS (they move with the swaps) and are written out once at the end: no cross-workgroup ordering is needed for them.
// Words alternate between two buffers by step parity: a workgroup publishes step k + 1 only after it has consumed everybody's step-k
// records, so the words of step k are dead for all readers before anyone overwrites them at step k + 2.
template <typename T>
struct QrcpTagArgs {
    int64_t m, n;
    T* A; int64_t lda;        // input, only READ by the kernel
    T* Aout; int64_t ldo;     // the factored columns leave here (the host copies them over A when info == 0: a lost word leaves A intact)
    int64_t* jpvt; T* tau;    // jpvt: scratch of n entries, copied to the caller's vector on success
    unsigned long long* tw;   // 2 x words_per_parity, zeroed by the host before the launch
    int* info;                // -7: a spin ran out (lost word): the host reports an error instead of hanging the device
    T tol3z;
    int64_t max_steps;
    int hq_formula;
};

constexpr int QT_NV = 8;      // values fetched per thread per batch of polling loads

__device__ __forceinline__ unsigned qt_get_u32(const unsigned long long* q, unsigned tag, int* info) {
    for (int spins = 0;; ++spins) {
        const unsigned long long w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(w >> 32) == tag) return (unsigned)w;
        if (spins > (1 << 22)) { atomicExch(info, -7); return (unsigned)w; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// fetch `cnt` (<= QT_NV) values idx[0..cnt) of the tagged array q -- and optionally one more 32-bit word xq -- in ONE batch of loads.
// The first batch is optimistic (data that is already there costs a single round trip); if a word is still missing the thread waits
// on one word and then takes the whole batch again.
template <typename T>
__device__ __forceinline__ void qt_get(const unsigned long long* q, const int64_t (&idx)[QT_NV], int cnt, unsigned tag, T (&out)[QT_NV], int* info,
                                       const unsigned long long* xq = nullptr, unsigned* xout = nullptr) {
    constexpr int W = (int)sizeof(T) / 4;
    if (cnt <= 0 && !xq) return;
    for (int spins = 0;; ++spins) {
        unsigned long long w[QT_NV * W];
        unsigned long long xw = (unsigned long long)tag << 32;
#pragma unroll
        for (int r = 0; r < QT_NV; ++r) {
#pragma unroll
            for (int h = 0; h < W; ++h) w[r * W + h] = (r < cnt) ? __hip_atomic_load(q + idx[r] * W + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        if (xq) xw = __hip_atomic_load(xq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = (unsigned)(xw >> 32) == tag;
#pragma unroll
        for (int r = 0; r < QT_NV * W; ++r) ok = ok && (r / W >= cnt || (unsigned)(w[r] >> 32) == tag);
        // every failed batch waits (paced, bounded by qt_get_u32's own 2^22 sleeps) on the FIRST word that is still missing, so each trip
        // retires at least one word: QT_NV * W + 2 trips cover any arrival order; a word that never arrives sets info in the wait
        const bool give_up = spins > QT_NV * W + 2 || *(volatile int*)info != 0;
        if (ok || give_up) {
            if (!ok) atomicExch(info, -7);
#pragma unroll
            for (int r = 0; r < QT_NV; ++r) {
                if constexpr (W == 1) out[r] = (T)__uint_as_float((unsigned)w[r]);
                else out[r] = (T)__longlong_as_double((long long)(((unsigned long long)(unsigned)w[2 * r + 1] << 32) | (unsigned)w[2 * r]));
            }
            if (xout) *xout = (unsigned)xw;
            return;
        }
        const unsigned long long* missing = nullptr;
#pragma unroll
        for (int r = QT_NV * W - 1; r >= 0; --r)
            if (r / W < cnt && (unsigned)(w[r] >> 32) != tag) missing = q + idx[r / W] * W + (r % W);
        if (!missing) missing = xq;
        (void)qt_get_u32(missing, tag, info);
    }
}
// one value (both halves of a double) plus one 32-bit word, requested together
template <typename T>
__device__ __forceinline__ T qt_get1(const unsigned long long* q, int64_t i, unsigned tag, int* info, const unsigned long long* xq, unsigned* xout) {
    constexpr int W = (int)sizeof(T) / 4;
    for (int spins = 0;; ++spins) {
        const unsigned long long lo = __hip_atomic_load(q + W * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = (W == 2) ? __hip_atomic_load(q + W * i + (W - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : lo;
        const unsigned long long xw = __hip_atomic_load(xq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag && (unsigned)(xw >> 32) == tag;
        if (ok || spins > (1 << 22)) {
            if (!ok) atomicExch(info, -7);
            *xout = (unsigned)xw;
            if constexpr (W == 1) return (T)__uint_as_float((unsigned)lo);
            else return (T)__longlong_as_double((long long)((hi << 32) | (unsigned)lo));
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void qt_put_u32(unsigned long long* q, unsigned tag, unsigned payload) {
    __hip_atomic_store(q, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void qt_put(unsigned long long* q, int64_t i, unsigned tag, T v) {
    if constexpr (sizeof(T) == 4) qt_put_u32(q + i, tag, __float_as_uint((float)v));
    else {
        const unsigned long long bits = (unsigned long long)__double_as_longlong((double)v);
        qt_put_u32(q + 2 * i, tag, (unsigned)bits);
        qt_put_u32(q + 2 * i + 1, tag, (unsigned)(bits >> 32));
    }
}
// words per parity: records (W G + G), one finished-column slot PER workgroup (W (m + 1) + 1 each), position-k column (W (m + 2) + 1)
template <typename T>
__host__ __device__ inline size_t qt_words(int64_t m, int64_t G) {
    constexpr size_t W = sizeof(T) / 4;
    return W * (size_t)G + (size_t)G + (size_t)G * (W * (size_t)(m + 1) + 1) + W * (size_t)(m + 2) + 1;
}
constexpr int QT_SPEC = 6;    // workgroups whose candidate ranked this high among the previous step's records finish it speculatively

template <typename T>
__global__ __launch_bounds__(256) void qrcp_tag_kernel(QrcpTagArgs<T> g) {
    constexpr int W = (int)sizeof(T) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 32-bit indices throughout (the host guarantees m, n < 2^31 - 2) and owned columns addressed by SLOT s (position me + G s):
    // 64-bit divisions by G on every access cost ~1 us per step
    const int G = (int)gridDim.x, me = (int)blockIdx.x;
    const int m = (int)g.m, n = (int)g.n;
    const int kmin = m < n ? m : n;
    const int kmax = (g.max_steps >= 0 && g.max_steps < kmin) ? (int)g.max_steps : kmin;
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_smem[];
    const int cpw = (n + G - 1) / G;                     // owned positions: j = me + G*s, s < cpw
    int64_t* l_jp = reinterpret_cast<int64_t*>(qr_smem);  // jpvt entries of the owned positions
    T* l_vn1 = reinterpret_cast<T*>(l_jp + cpw);
    T* l_vn2 = l_vn1 + cpw;
    T* l_v = l_vn2 + cpw;                                // the step's finished pivot column (m)
    T* lds_cols = l_v + m;
    __shared__ T s_val[4];
    __shared__ T s_wv[4];
    __shared__ int64_t s_wp[4];
    __shared__ int s_ww[4];
    __shared__ int s_cnt[4];
    __shared__ T s_tau;
    __shared__ unsigned s_jp;
    auto colslot = [&](int sl) -> T* { return lds_cols + (size_t)sl * m; };
    for (int sl = 0, j = me; j < n; ++sl, j += G) {
        T* dst = colslot(sl);
        const T* src = g.A + (int64_t)j * g.lda;
        for (int i = tid; i < m; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    for (int sl = wid; sl < cpw; sl += 4) {
        const int j = me + G * sl;
        if (j >= n) continue;
        const T* col = colslot(sl);
        T ss = 0;
        for (int i = lane; i < m; i += 64) ss += col[i] * col[i];
        ss = wave_sum_fast(ss);
        if (lane == 0) {
            T nr = sqrt(ss);
            l_vn1[sl] = nr; l_vn2[sl] = nr;
            l_jp[sl] = j + 1;
        }
    }
    __syncthreads();
    const size_t PW = qt_words<T>(m, G);
    const size_t CS = (size_t)W * (m + 1) + 1;           // words of one finished-column slot
    T hv_prev = std::numeric_limits<T>::infinity();       // record of workgroup `tid` at the previous step (G <= 256); +inf: nobody speculates at step 0
    // dlarfg on the column at position `pos` (rows k+1.. are scaled, row k becomes beta), staged in l_v and published in this workgroup's slot
    auto finish_column = [&](int psl, int k, unsigned tag, unsigned long long* cw) {
        const T* col = colslot(psl);
        T ss = 0;
        for (int i = k + 1 + tid; i < m; i += 256) ss += col[i] * col[i];
        ss = wave_sum_fast(ss);
        __syncthreads();
        if (lane == 0) s_val[wid] = ss;
        __syncthreads();
        const T xnorm = sqrt(s_val[0] + s_val[1] + s_val[2] + s_val[3]);
        const T alpha = col[k];
        T beta = alpha, scale = 0, my_tau = 0;
        if (xnorm != T(0)) {
            beta = -copysign(hypot(alpha, xnorm), alpha);
            my_tau = (beta - alpha) / beta;
            scale = T(1) / (alpha - beta);
        }
        for (int i = tid; i < m; i += 256) {
            T v = col[i];
            if (i == k) v = beta; else if (i > k) v *= scale;
            l_v[i] = v;
            qt_put<T>(cw, i, tag, v);
        }
        if (tid == 0) { qt_put<T>(cw, m, tag, my_tau); qt_put_u32(cw + (size_t)W * (m + 1), tag, (unsigned)l_jp[psl]); s_tau = my_tau; }
    };
#ifdef RLHIP_QT_PROF
    long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long pt = wall_clock64();
#define QT_MARK(i) { const long long now_ = wall_clock64(); pf[i] += now_ - pt; pt = now_; }
#else
#define QT_MARK(i)
#endif
    int own_k = 0, slot_k = 0;                            // k = own_k + G slot_k
    for (int k = 0; k < kmax; ++k, own_k = (own_k + 1 == G ? 0 : own_k + 1), slot_k += (own_k == 0)) {
        const unsigned tag = (unsigned)(k + 1);
        unsigned long long* base = g.tw + (size_t)(k & 1) * PW;
        unsigned long long* hv = base;                          // [W G]   best partial norm of workgroup w
        unsigned long long* hp = hv + (size_t)W * G;            // [G]     its position (n: nothing to offer); bit 31: column already finished
        unsigned long long* cws = hp + G;                       // [G][CS] finished pivot column of workgroup w: m values, tau, jpvt entry
        unsigned long long* kc = cws + (size_t)G * CS;          // [W (m + 2)] the column at position k, vn1, vn2
        unsigned long long* kcj = kc + (size_t)W * (m + 2);     // [1]     its jpvt entry
        unsigned long long* my_cw = cws + (size_t)me * CS;
        // ---- 1. local candidate over owned positions >= k (first maximum)
        T best = T(-1); int bpos = n, bsl = 0;
        for (int sl = 0, j = me; j < n; ++sl, j += G) {
            if (j < k) continue;
            T v = l_vn1[sl];
            if (v > best) { best = v; bpos = j; bsl = sl; }
        }
        //      Would this candidate have ranked among the first QT_SPEC of the previous step's records?  Then finish it NOW, while the
        //      records travel: if it wins, its column is already on its way when the others learn the winner (one hand-off per step
        //      instead of two).  The guess only decides WHEN the winner's dlarfg runs, never what it computes.
        int ahead = 0;
        {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(tid < G && hv_prev > best);
            if (lane == 0) s_cnt[wid] = __builtin_popcountll(bal);
            __syncthreads();
            ahead = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        }
        const bool spec = (bpos < n) && (ahead < QT_SPEC);
        if (tid == 0) { qt_put<T>(hv, me, tag, best); qt_put_u32(hp + me, tag, (unsigned)bpos | (spec ? 0x80000000u : 0u)); }
        if (me == own_k) {                                      // the owner of position k publishes that column
            const T* col = colslot(slot_k);
            for (int i = tid; i < m; i += 256) qt_put<T>(kc, i, tag, col[i]);
            if (tid == 0) { qt_put<T>(kc, m, tag, l_vn1[slot_k]); qt_put<T>(kc, m + 1, tag, l_vn2[slot_k]); qt_put_u32(kcj, tag, (unsigned)l_jp[slot_k]); }
        }
        QT_MARK(0)
        // a speculating workgroup is the likely winner, and the winner receives the column that sits at position k: fetch it while the
        // records travel (registers; unused if somebody else wins)
        T kpre[QT_NV]; unsigned kpre_jp = 0; bool have_k = false;
        if (spec) {
            finish_column(bsl, k, tag, my_cw);
            if (bpos != k && m + 2 <= 256 * QT_NV) {
                int64_t ix[QT_NV]; int cnt = 0;
#pragma unroll
                for (int r = 0; r < QT_NV; ++r) { ix[r] = tid + 256 * r; if (ix[r] < m + 2) cnt = r + 1; }
                qt_get<T>(kc, ix, cnt, tag, kpre, g.info, tid == 0 ? kcj : nullptr, &kpre_jp);
                have_k = true;
            }
        }
        QT_MARK(1)
        // ---- 2. global pivot (every workgroup, redundantly): max norm, ties -> smallest position
        {
            T v = T(-1); int q = n; int w = own_k; unsigned fl = 0;
            T mine = T(-1);
            for (int ww = tid; ww < G; ww += 256) {
                unsigned pw; const T v2 = qt_get1<T>(hv, ww, tag, g.info, hp + ww, &pw);     // value and position words in one batch of loads
                const int q2 = (int)(pw & 0x7fffffffu);
                mine = v2;
                if (q2 < n && (v2 > v || (v2 == v && q2 < q))) { v = v2; q = q2; w = ww; fl = pw >> 31; }
            }
            hv_prev = (tid < G) ? mine : T(-1);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const T v2 = __shfl_xor(v, off); const int q2 = __shfl_xor(q, off); const int w2 = __shfl_xor(w, off);
                const unsigned f2 = __shfl_xor(fl, off);
                if (q2 < n && (v2 > v || (v2 == v && q2 < q))) { v = v2; q = q2; w = w2; fl = f2; }
            }
            if (lane == 0) { s_wv[wid] = v; s_wp[wid] = q; s_ww[wid] = w | (int)(fl << 30); }
        }
        __syncthreads();
        T gv = s_wv[0]; int p = (int)s_wp[0]; int wf = s_ww[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_wp[w] < n && (s_wv[w] > gv || (s_wv[w] == gv && s_wp[w] < p))) { gv = s_wv[w]; p = (int)s_wp[w]; wf = s_ww[w]; }
        int wstar = wf & 0x3fffffff;
        bool finished = (wf >> 30) & 1;
        int psl = bsl;                                         // slot of position p in the winner's LDS (meaningful in the winner only)
        if (p >= n) { p = k; wstar = own_k; finished = false; psl = slot_k; }   // nothing comparable left (NaNs): natural order
        if (tid == wstar) hv_prev = T(-1);            // the winner's record is consumed; its next candidate is unknown to the others
        QT_MARK(2)
        // ---- 3. / 4. the winner finishes its column unless it already has; the others fetch it from the winner's slot
        const unsigned long long* cw = cws + (size_t)wstar * CS;
        if (me == wstar) {
            if (!finished) finish_column(psl, k, tag, my_cw);
        } else {
            // only the owner of position k needs the rows above k (the finished R entries of the column it installs); everybody else
            // applies the reflector to rows >= k and fetches just those (on average half the column)
            const int first = (me == own_k) ? 0 : k;
            for (int i0 = first; i0 < m + 1; i0 += 256 * QT_NV) {
                int64_t ix[QT_NV]; T vv[QT_NV];
                int cnt = 0;
#pragma unroll
                for (int r = 0; r < QT_NV; ++r) { ix[r] = i0 + tid + 256 * r; if (ix[r] < m + 1) cnt = r + 1; }
                const bool want_jp = (me == own_k && tid == 0 && i0 == 0);          // the owner of position k also takes the jpvt entry
                unsigned xj = 0;
                qt_get<T>(cw, ix, cnt, tag, vv, g.info, want_jp ? cw + (size_t)W * (m + 1) : nullptr, &xj);
                if (want_jp) s_jp = xj;
#pragma unroll
                for (int r = 0; r < QT_NV; ++r)
                    if (r < cnt) { if (ix[r] < m) l_v[ix[r]] = vv[r]; else s_tau = vv[r]; }
            }
        }
        __syncthreads();
        const T tauk = s_tau;
        QT_MARK(3)
        // ---- 5. install the moved columns; the jpvt entries move with them
        int64_t jp_p = 0;
        if (me == own_k && tid == 0) jp_p = (me == wstar) ? l_jp[psl] : (int64_t)s_jp;
        if (p != k && me == wstar) {                           // the winner owns position p: it receives the column that was at k
            T* col = colslot(psl);
            if (have_k) {
#pragma unroll
                for (int r = 0; r < QT_NV; ++r) {
                    const int i = tid + 256 * r;
                    if (i < m) col[i] = kpre[r];
                    else if (i == m) l_vn1[psl] = kpre[r];
                    else if (i == m + 1) l_vn2[psl] = kpre[r];
                }
                if (tid == 0) l_jp[psl] = (int64_t)kpre_jp;
            } else {
                for (int i0 = 0; i0 < m + 2; i0 += 256 * QT_NV) {
                    int64_t ix[QT_NV]; T vv[QT_NV];
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < QT_NV; ++r) { ix[r] = i0 + tid + 256 * r; if (ix[r] < m + 2) cnt = r + 1; }
                    const bool want_jp = (tid == 0 && i0 == 0);
                    unsigned xj = 0;
                    qt_get<T>(kc, ix, cnt, tag, vv, g.info, want_jp ? kcj : nullptr, &xj);
                    if (want_jp) l_jp[psl] = (int64_t)xj;
#pragma unroll
                    for (int r = 0; r < QT_NV; ++r)
                        if (r < cnt) {
                            if (ix[r] < m) col[ix[r]] = vv[r];
                            else if (ix[r] == m) l_vn1[psl] = vv[r];
                            else l_vn2[psl] = vv[r];
                        }
                }
            }
        }
        if (me == own_k) {
            T* col = colslot(slot_k);
            for (int i = tid; i < m; i += 256) col[i] = l_v[i];
            if (tid == 0) { g.tau[k] = tauk; l_jp[slot_k] = jp_p; }
        }
        __syncthreads();
        QT_MARK(4)
        // ---- 6. apply H = I - tau v v^T (v_k = 1) to owned columns j > k, one wave per column; down-date norms.  Rows are walked in
        //      64-row slabs from the slab holding row k; four slabs per trip keep four independent LDS loads / fma chains in flight.
        const int r_lo = (k & ~63) + lane;
        for (int sl = wid; sl < cpw; sl += 4) {
            const int j = me + G * sl;
            if (j <= k || j >= n) continue;
            T* col = colslot(sl);
            if (tauk != T(0)) {
                T w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                int i = r_lo;
                for (; i + 192 < m; i += 256) {
                    const T v0 = (i > k) ? l_v[i] : (i == k ? T(1) : T(0));
                    w0 += v0 * col[i]; w1 += l_v[i + 64] * col[i + 64]; w2 += l_v[i + 128] * col[i + 128]; w3 += l_v[i + 192] * col[i + 192];
                }
                for (; i < m; i += 64) w0 += ((i > k) ? l_v[i] : (i == k ? T(1) : T(0))) * col[i];
                const T w = wave_sum_fast((w0 + w1) + (w2 + w3)) * tauk;
                i = r_lo;
                for (; i + 192 < m; i += 256) {
                    const T v0 = (i > k) ? l_v[i] : (i == k ? T(1) : T(0));
                    col[i] -= w * v0; col[i + 64] -= w * l_v[i + 64]; col[i + 128] -= w * l_v[i + 128]; col[i + 192] -= w * l_v[i + 192];
                }
                for (; i < m; i += 64) col[i] -= w * ((i > k) ? l_v[i] : (i == k ? T(1) : T(0)));
            }
            T v1 = l_vn1[sl];
            if (v1 != T(0)) {
                T akj = fabs(col[k]);
                T r = akj / v1;
                T temp = g.hq_formula ? (T(1) + r) * (T(1) - r) : T(1) - r * r;
                temp = temp > T(0) ? temp : T(0);
                T q = v1 / l_vn2[sl];
                T temp2 = temp * q * q;
                if (temp2 <= g.tol3z) {
                    T ss = 0;
                    for (int i = k + 1 + lane; i < m; i += 64) ss += col[i] * col[i];
                    ss = wave_sum_fast(ss);
                    v1 = sqrt(ss);
                    if (lane == 0) { l_vn1[sl] = v1; l_vn2[sl] = v1; }
                } else {
                    if (lane == 0) l_vn1[sl] = v1 * sqrt(temp);
                }
            }
        }
        __syncthreads();
        QT_MARK(5)
    }
#ifdef RLHIP_QT_PROF
    if (tid == 0 && me == G / 2) for (int i = 0; i < 6; ++i) ((long long*)(g.info + 2))[i] = pf[i];
#endif
    for (int sl = 0, j = me; j < n; ++sl, j += G) {
        const T* src = colslot(sl);
        T* dst = g.Aout + (int64_t)j * g.ldo;
        for (int i = tid; i < m; i += 256) dst[i] = src[i];
        if (tid == 0) g.jpvt[j] = l_jp[sl];
    }
}

__global__ void zero_u32(unsigned* p) { *p = 0; }

// ---------------------------------------------------------------------------------------------------------------------
// Unpivoted Householder QR (geqrf / geqr2 order) as a PIPELINE instead of a step-synchronous sweep.  Without pivoting
// there is no all-to-all decision per step: the owner of column k computes H_k and publishes it (the finished column goes
// to its final place in A with write-through stores, then a per-step flag is raised); every workgroup applies H_0, H_1,
// ... to its own columns in order, waiting only on the flag of the reflector it needs next.  Nobody waits for the slowest
// workgroup of the PREVIOUS step, and the owner of column k+1 updates that column first and publishes H_{k+1} before it
// finishes applying H_k to the rest of its columns (look-ahead).  Critical path per step: flag poll + one column read + one
// column update + reflector + publish (~5 us) against ~10-16 us for the rendezvous version.
template <typename T>
struct QrPipeArgs {
    int64_t m, n;
    T* A; int64_t lda;
    T* tau;
    unsigned* flag;           // kmax entries, zeroed by the host
    int use_lds;
    int v_in_lds;             // the current reflector fits in LDS (m doubles); otherwise it is read from its column of A
    int wg_per_col;           // tall-skinny: few columns per workgroup -> the whole workgroup updates one column at a time
    int chunk;                // columns are dealt to the workgroups in chunks of this many consecutive columns (block-cyclic)
};

// USE_LDS / V_IN_LDS (= USE_LDS / V_IN_LDS) are template parameters, not run-time flags: `flag ? lds_pointer : global_pointer` is a GENERIC
// pointer to hipcc, every access through it a flat_ instruction (218 of them in this kernel before) that is slower on LDS data and, because a flat
// access may land on either side, is waited for with vmcnt(0) AND lgkmcnt(0).
template <typename T, bool USE_LDS, bool V_IN_LDS>
__global__ __launch_bounds__(256) void qr_pipe_kernel(QrPipeArgs<T> g) {
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t G = gridDim.x, me = blockIdx.x;
    const int64_t m = g.m, n = g.n;
    const int64_t kmax = m < n ? m : n;
    extern __shared__ __attribute__((aligned(16))) unsigned char qp_smem[];
    T* l_v = reinterpret_cast<T*>(qp_smem);              // current reflector (m), when it fits
    T* lds_cols = l_v + (V_IN_LDS ? m : 0);
    __shared__ T s_val[4];
    __shared__ T s_tau;
    // Block-cyclic column ownership: chunks of CH consecutive columns go round the workgroups.  Inside a chunk consecutive steps
    // stay in ONE workgroup (the look-ahead below makes H_{k+1} right after applying H_k to column k+1, no flag round trip);
    // only every CH-th step crosses workgroups.  CH = 1 is the plain cyclic layout.
    const int64_t CH = g.chunk;
    const int64_t cpw = (((n + CH - 1) / CH + G - 1) / G) * CH;                 // local column slots per workgroup
    auto owner = [&](int64_t j) -> int64_t { return (j / CH) % G; };
    auto slot = [&](int64_t j) -> int64_t { return (j / (CH * G)) * CH + (j % CH); };
    auto col_of = [&](int64_t c) -> int64_t { return ((c / CH) * G + me) * CH + (c % CH); };   // increasing in c
    auto colptr = [&](int64_t j) {
        if constexpr (USE_LDS) return lds_cols + slot(j) * m;
        else return g.A + j * g.lda;
    };
    if constexpr (USE_LDS) {
        for (int64_t c = 0; c < cpw; ++c) {
            const int64_t j = col_of(c);
            if (j >= n) break;
            T* dst = lds_cols + c * m;
            const T* src = g.A + j * g.lda;
            for (int64_t i = tid; i < m; i += 256) dst[i] = src[i];
        }
        __syncthreads();
    }
    auto vread = [](const T* lv, const T* gv, int64_t i) -> T {
        if constexpr (V_IN_LDS) return lv[i];
        else return gv[i];
    };
    // compute H_k from (already updated) column k, publish it, leave v in l_v and tau in s_tau
    auto make_reflector = [&](int64_t k) {
        T* col = colptr(k);
        T ss = 0;
#pragma unroll 8
        for (int64_t i = k + 1 + tid; i < m; i += 256) ss += col[i] * col[i];
        ss = wave_sum(ss);
        __syncthreads();
        if (lane == 0) s_val[wid] = ss;
        __syncthreads();
        const T xnorm = sqrt(s_val[0] + s_val[1] + s_val[2] + s_val[3]);
        const T alpha = col[k];
        T beta = alpha, scale = 0, tk = 0;
        if (xnorm != T(0)) {
            beta = -copysign(hypot(alpha, xnorm), alpha);
            tk = (beta - alpha) / beta;
            scale = T(1) / (alpha - beta);
        }
        T* gcol = g.A + k * g.lda;
#pragma unroll 8
        for (int64_t i = tid; i < m; i += 256) {
            T v = col[i];
            if (i == k) v = beta; else if (i > k) v *= scale;
            if constexpr (V_IN_LDS) l_v[i] = v;
            if constexpr (USE_LDS) col[i] = v;
            pub_store(gcol + i, v);                       // final content of column k of A (R above, beta, v below)
        }
        if (tid == 0) { pub_store(g.tau + k, tk); s_tau = tk; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(g.flag + k, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            // tall matrices read v back from its column of A with ordinary loads: drop this CU's stale L1 lines of that column
            if (!V_IN_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (!V_IN_LDS) __syncthreads();
    };
    // apply H (v in l_v, tau = tk, pivot row k) to column j
    auto apply_one = [&](int64_t k, T tk, T* col) {       // one wave per column
        const T* gv = g.A + k * g.lda;                    // (tall matrices: v straight from its published column)
        T w = 0;
#pragma unroll 4
        for (int64_t i = k + lane; i < m; i += 64) w += ((i == k) ? T(1) : vread(l_v, gv, i)) * col[i];
        w = wave_sum(w) * tk;
#pragma unroll 4
        for (int64_t i = k + lane; i < m; i += 64) col[i] -= w * ((i == k) ? T(1) : vread(l_v, gv, i));
    };
    auto apply_wg = [&](int64_t k, T tk, T* col) {        // whole workgroup on one (long) column
        const T* gv = g.A + k * g.lda;
        T w = 0;
#pragma unroll 8
        for (int64_t i = k + tid; i < m; i += 256) w += ((i == k) ? T(1) : vread(l_v, gv, i)) * col[i];
        w = wave_sum(w);
        __syncthreads();
        if (lane == 0) s_val[wid] = w;
        __syncthreads();
        w = (s_val[0] + s_val[1] + s_val[2] + s_val[3]) * tk;
#pragma unroll 8
        for (int64_t i = k + tid; i < m; i += 256) col[i] -= w * ((i == k) ? T(1) : vread(l_v, gv, i));
    };
    bool have_next = false;                               // reflector k already made by the look-ahead of step k-1
    for (int64_t k = 0; k < kmax; ++k) {
        const int64_t own_k = owner(k);
        T tk;
        if (me == own_k) {
            if (!have_next) make_reflector(k);            // (k = 0, or G == 1 handled by the look-ahead below)
            tk = s_tau;
        } else {
            if (tid == 0) {
                while (__hip_atomic_load(g.flag + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            const T* gcol = g.A + k * g.lda;
            if constexpr (V_IN_LDS)
                for (int64_t i = k + tid; i < m; i += 256) l_v[i] = gcol[i];
            if (tid == 0) s_tau = g.tau[k];
            __syncthreads();
            tk = s_tau;
        }
        have_next = false;
        if (tk != T(0)) {
            // look-ahead: the owner of column k+1 brings that column up to date first and publishes H_{k+1} right away
            const int64_t kn = k + 1;
            const bool own_next = (kn < kmax) && (me == owner(kn));
            if (own_next) {
                if (g.wg_per_col) apply_wg(k, tk, colptr(kn));
                else if (wid == 0) apply_one(k, tk, colptr(kn));
                __syncthreads();
            }
            // remaining owned columns j > k (all of them when this workgroup does not own k+1)
            // H_k for the remaining owned columns (still from l_v: make_reflector(k+1) overwrites it afterwards)
            const T saved_tau = tk;
            if (g.wg_per_col) {
                for (int64_t c = 0; c < cpw; ++c) {
                    const int64_t j = col_of(c);
                    if (j >= n) break;
                    if (j <= k || (own_next && j == kn)) continue;
                    apply_wg(k, saved_tau, colptr(j));
                }
            } else {
                for (int64_t c = wid; c < cpw; c += 4) {
                    const int64_t j = col_of(c);
                    if (j >= n) break;
                    if (j <= k || (own_next && j == kn)) continue;
                    apply_one(k, saved_tau, colptr(j));
                }
            }
            __syncthreads();
            if (own_next) { make_reflector(kn); have_next = true; }
        } else {
            const int64_t kn = k + 1;
            if ((kn < kmax) && (me == owner(kn))) { make_reflector(kn); have_next = true; }
        }
    }
    // columns that never became a pivot (n > m) or global-path bookkeeping: write back what lives only in LDS
    if constexpr (USE_LDS) {
        __syncthreads();
        for (int64_t c = 0; c < cpw; ++c) {
            const int64_t j = col_of(c);
            if (j >= n) break;
            if (j < kmax) continue;                       // pivot columns were published in place
            const T* src = lds_cols + c * m;
            T* dst = g.A + j * g.lda;
            for (int64_t i = tid; i < m; i += 256) dst[i] = src[i];
        }
    }
}

__global__ void zero_u32_n(unsigned* p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}


}  // namespace

namespace rlhip {

template <typename T>
int gemqrt_lt(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, int64_t nb, const T* V, int64_t ldv, const T* Tm, int64_t ldt, T* C, int64_t ldc);
template <typename T>
int larft_gram(rlhip_ctx* c, int64_t m, int64_t k, const T* V, int64_t ldv, const T* tau, T* Tm, int64_t ldt);
template <typename T>
int gemm(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
         const T* B, int64_t ldb, T beta, T* C, int64_t ldc);

template <typename T>
int geqrf_cholqr(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau, int* done);
template <typename T>
int geqrf_blk(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau_dev);

template <typename T>
static int qr_core(rlhip_ctx* c, int pivot, int64_t m, int64_t n, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev, int64_t max_steps = -1,
                   int hq_formula = 0);

// dynamic-LDS limit (150 KiB) of one instantiation of the cooperative QR kernels, raised once per (device, kernel address)
static hipError_t qr_kernel_lds_limit(rlhip_ctx* c, const void* kern) {
    static std::mutex mu;
    static std::vector<std::pair<int, const void*>> seen;
    std::lock_guard<std::mutex> lk(mu);
    const std::pair<int, const void*> key(c->device, kern);
    for (auto const& e : seen) if (e == key) return hipSuccess;
    hipError_t le = hipSetDevice(c->device);
    if (le == hipSuccess) le = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (le == hipSuccess) seen.push_back(key);
    return le;
}

template <typename T>
int geqp3(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev) {
    return qr_core<T>(c, 1, m, n, A, lda, jpvt_dev, tau_dev);
}

// Pivoted Householder QR restricted to the first `steps` columns, with the norm down-date in HQRRP's form: the device
// counterpart of NoFLA_QRPmod_WY_unb_var4(pivoting = 1, num_stages = steps) (rl_hqrrp.hh:516-770).  jpvt returns the
// whole permutation (1-based) that the column swaps of those steps produce.
template <typename T>
int qrp_partial(rlhip_ctx* c, int64_t m, int64_t n, int64_t steps, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev) {
    return qr_core<T>(c, 1, m, n, A, lda, jpvt_dev, tau_dev, steps, 1);
}
// The first `steps` steps of geqp3 itself (LAPACK's norm down-date form): on exit rows 0 .. steps-1 of R are final for ALL columns, the trailing
// (m - steps) x (n - steps) block carries every reflector so far and jpvt the permutation so far -- geqp3 of that block, its pivots applied to
// the columns of the finished rows and composed into jpvt, completes the factorization (CQRRPT's split QRCP, rl_cqrrpt.hh).
template <typename T>
int geqp3_steps(rlhip_ctx* c, int64_t m, int64_t n, int64_t steps, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev) {
    return qr_core<T>(c, 1, m, n, A, lda, jpvt_dev, tau_dev, steps, 0);
}
template int geqp3_steps<double>(rlhip_ctx*, int64_t, int64_t, int64_t, double*, int64_t, int64_t*, double*);
template int geqp3_steps<float>(rlhip_ctx*, int64_t, int64_t, int64_t, float*, int64_t, int64_t*, float*);
template int qrp_partial<double>(rlhip_ctx*, int64_t, int64_t, int64_t, double*, int64_t, int64_t*, double*);
template int qrp_partial<float>(rlhip_ctx*, int64_t, int64_t, int64_t, float*, int64_t, int64_t*, float*);

// lapack::geqrf: Householder QR without pivoting.  Wide input (n > m, BQRRP's permuted sketch rl_bqrrp.hh:356): only the
// leading m x m block goes through the step-synchronous kernel; the remaining columns get Q^T applied as ONE compact-WY
// block (larft + gemqrt on the MFMA path).
// ---- exponent-range guard of geqrf.  The Householder kernels of this library form column norms as plain sums of squares (one reduction
// round gives the norm, the reflector's action and the larft entries at once) where LAPACK's larfg / nrm2 scale; entries above ~1e19 / below
// ~1e-19 in fp32 (1e154 / 1e-154 in fp64) would turn into inf, NaN or a zero tau.  So every geqrf call measures max |a_ij| (one pass, the
// result stays on the device), factors s A with s the power of two that brings it into [1, 2) when it lies outside the safe window --
// exact: the reflectors and tau do not depend on the scale -- and gives R its scale back.  No host read; inside the window (always, on
// this path's sketches) the two rescale launches find s = 1 and write nothing.  What is NOT protected: a column more than half the
// exponent range below the matrix's largest entry loses its norm to underflow.
// w[0] = max |a_ij| as a bit pattern (non-negative IEEE values order like unsigned integers)
template <typename T>
__global__ __launch_bounds__(256) void geqrf_absmax_kernel(int64_t m, int64_t n, const T* __restrict__ A, int64_t lda, unsigned long long* __restrict__ w) {
    unsigned long long best = 0;
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
#pragma unroll 8
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
            const double v = fabs((double)A[i + j * lda]);
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            best = b > best ? b : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(best, off, 64); best = o > best ? o : best; }
    // ONE atomic per workgroup: the atomics on the single result word serialise (~12-35 ns each) -- with one per wavefront and 2048
    // workgroups they, not the 51 MB of a 200000 x 32 panel, were the kernel's 100 us
    __shared__ unsigned long long wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = wmax[0];
        for (int q = 1; q < 4; ++q) b = wmax[q] > b ? wmax[q] : b;
        if (b) atomicMax(w, b);
    }
}
// the scale this matrix needs (1 inside the safe window, and for zero / non-finite matrices)
template <typename T>
__device__ __forceinline__ double geqrf_scale_of(const unsigned long long* w) {
    const double mx = __longlong_as_double((long long)w[0]);
    const double hi = (sizeof(T) == 8) ? 1e149 : 2.8e14, lo = (sizeof(T) == 8) ? 1e-149 : 7e-15;
    if (!(mx > 0.0) || !(mx < 1.7e308) || (mx < hi && mx > lo)) return 1.0;
    int ex = 0;
    (void)frexp(mx, &ex);
    return ldexp(1.0, 1 - ex);
}
// A *= s (back = 0: the whole matrix) or A(i <= j) /= s (back = 1: the R part of the result)
template <typename T>
__global__ __launch_bounds__(256) void geqrf_rescale_kernel(int64_t m, int64_t n, T* __restrict__ A, int64_t lda, const unsigned long long* __restrict__ w, int back) {
    const double s = geqrf_scale_of<T>(w);
    if (s == 1.0) return;
    const double f = back ? 1.0 / s : s;               // (a power of two: exact both ways)
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256)
            if (!back || i <= j) A[i + j * lda] = (T)((double)A[i + j * lda] * f);
}

template <typename T>
int geqrf_core(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau_dev);

template <typename T>
int geqrf(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau_dev) {
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (lda < (m > 1 ? m : 1)) return -5;
    if (m == 0 || n == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    unsigned long long* w = ws_alloc<unsigned long long>(c, 4);
    if (!w) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    // a thread walks rows of a column: coalesced, no index divisions, eight loads in flight; at most 1024 workgroups (one atomic each)
    int64_t bx = (m + 1023) / 1024, by = n;
    if (bx > 64) bx = 64;
    if (by > 1024 / bx) by = 1024 / bx;
    const dim3 blocks((unsigned)bx, (unsigned)(by < 1 ? 1 : by));
    hipError_t e = hipMemsetAsync(w, 0, sizeof(unsigned long long), c->stream);
    if (e != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(e); }
    hipLaunchKernelGGL(geqrf_absmax_kernel<T>, blocks, dim3(256), 0, c->stream, m, n, A, lda, w);
    hipLaunchKernelGGL(geqrf_rescale_kernel<T>, blocks, dim3(256), 0, c->stream, m, n, A, lda, w, 0);
    int rc = geqrf_core<T>(c, m, n, A, lda, tau_dev);
    if (!rc) {
        hipLaunchKernelGGL(geqrf_rescale_kernel<T>, blocks, dim3(256), 0, c->stream, m, n, A, lda, w, 1);
        e = hipGetLastError();
        if (e != hipSuccess) rc = RLHIP_ERR_HIP(e);
    }
    rlhip_ws_release(c, mark);
    return rc;
}

template <typename T>
int geqrf_core(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau_dev) {
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (lda < (m > 1 ? m : 1)) return -5;
    if (m == 0 || n == 0) return 0;
    // Blocked right-looking Householder QR, panel width 128.  A panel is factored by the BLAS-3 route (Cholesky-QR twice +
    // Householder reconstruction: same reflectors, GEMM speed) whenever it is at least twice as tall as wide and the route can be
    // trusted (verified inside geqrf_cholqr), by the flag-pipelined Householder kernel otherwise; the trailing columns get
    // Q_panel^T as one compact-WY block on the MFMA GEMMs.  One panel covers the tall-skinny case; wide inputs (n > m) simply
    // have trailing columns beyond the last reflector.
    constexpr int64_t NBQ = 256;       // a BLAS-3 panel costs ~1.3 ms of launch latency whatever its width: few, wide panels
    const int64_t kmax = m < n ? m : n;
    // sketch-sized, nearly square or wide (up to 2048 rows, 8 columns per CU): the register-resident block-pipelined kernel (qr_blk.hip)
    // factors the leading kmax columns in one launch -- 2048 x 2048 fp32: 12 ms of pipelined / blocked panels before; the columns beyond
    // the last reflector (BQRRP's permuted sketch is 2048 x cols) then get Q^T in compact-WY blocks of NBQ reflectors on the MFMA GEMMs
    if (kmax >= 64 && 10 * m < 19 * kmax) {
        const int rb = geqrf_blk<T>(c, m, kmax, A, lda, tau_dev);
        if (rb < 0) return rb;
        if (rb == 1) {
            if (n <= kmax) return 0;
            size_t markb = rlhip_ws_mark(c);
            T* Tb = ws_alloc<T>(c, (size_t)NBQ * NBQ);
            if (!Tb) { rlhip_ws_release(c, markb); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
            int rcb = 0;
            for (int64_t j0 = 0; j0 < kmax && !rcb; j0 += NBQ) {
                const int64_t jb = (kmax - j0 < NBQ) ? (kmax - j0) : NBQ;
                T* P = A + j0 + j0 * lda;
                rcb = larft_gram<T>(c, m - j0, jb, P, lda, tau_dev + j0, Tb, jb);
                if (!rcb) rcb = gemqrt_lt<T>(c, m - j0, n - kmax, jb, jb, P, lda, Tb, jb, A + j0 + kmax * lda, lda);
            }
            rlhip_ws_release(c, markb);
            return rcb;
        }
    }
    constexpr int64_t pipe_max = 1280;
    // sketch-sized, nearly square problems: the pipelined kernel alone beats blocking (1280 x 1024: 11.2 vs 12.3 ms); from about twice
    // as tall as wide the CholQR-panel route wins (2000 x 1000: 13.0 vs 14.3 ms, 2560 x 1024: 12.2 vs 17.4 ms)
    if (kmax <= pipe_max && (m < 16 * kmax && (10 * m < 19 * kmax || kmax < 600))) {
        int rc0 = qr_core<T>(c, 0, m, kmax, A, lda, nullptr, tau_dev);
        if (rc0 || n <= kmax) return rc0;
        size_t mark0 = rlhip_ws_mark(c);
        T* T0 = ws_alloc<T>(c, (size_t)kmax * kmax);
        if (!T0) { rlhip_ws_release(c, mark0); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        rc0 = larft_gram<T>(c, m, kmax, A, lda, tau_dev, T0, kmax);
        if (!rc0) rc0 = gemqrt_lt<T>(c, m, n - kmax, kmax, kmax, A, lda, T0, kmax, A + kmax * lda, lda);
        rlhip_ws_release(c, mark0);
        return rc0;
    }
    size_t mark = rlhip_ws_mark(c);
    T* Tm = (n > NBQ || n > kmax) ? ws_alloc<T>(c, (size_t)NBQ * NBQ) : nullptr;
    int rc = 0;
    for (int64_t j0 = 0; j0 < kmax && !rc; j0 += NBQ) {
        const int64_t jb = (kmax - j0 < NBQ) ? (kmax - j0) : NBQ;
        const int64_t rows = m - j0;
        T* P = A + j0 + j0 * lda;
        int done = 0;
        if (rows >= 2 * jb && jb >= 8 && (size_t)rows * jb >= 16384) rc = geqrf_cholqr<T>(c, rows, jb, P, lda, tau_dev + j0, &done);
        if (!rc && !done) rc = qr_core<T>(c, 0, rows, jb, P, lda, nullptr, tau_dev + j0);
        const int64_t rest = n - j0 - jb;
        if (!rc && rest > 0) {
            if (!Tm) { rc = RLHIP_ERR_HIP(hipErrorOutOfMemory); break; }
            rc = larft_gram<T>(c, rows, jb, P, lda, tau_dev + j0, Tm, jb);
            if (!rc) rc = gemqrt_lt<T>(c, rows, rest, jb, jb, P, lda, Tm, jb, A + j0 + (j0 + jb) * lda, lda);
        }
    }
    rlhip_ws_release(c, mark);
    return rc;
}

template <typename T>
__global__ void ungqr_seed_kernel(int64_t n, const T* __restrict__ V, int64_t ldv, T* __restrict__ V1t) {
    // V1t (n x n) = unit-lower(V[0:n, 0:n])^T
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int64_t i = idx % n, j = idx / n;      // V1t[i, j] = V1[j, i]
    T v = 0;
    if (j > i) v = V[j + i * ldv]; else if (j == i) v = 1;
    V1t[idx] = v;
}
template <typename T>
__global__ void ungqr_finish_kernel(int64_t m, int64_t n, T* __restrict__ Q, int64_t ldq, const T* __restrict__ V, int64_t ldv,
                                    const T* __restrict__ Wm) {
    // rows < n of the result: Q[i, j] = delta_ij - sum_l V1[i, l] W[l, j]  (V1 unit lower); one thread per entry.
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int64_t i = idx % n, j = idx / n;
    T s = (i == j) ? T(1) : T(0);
    for (int64_t l = 0; l <= i; ++l) s -= ((l == i) ? T(1) : V[i + l * ldv]) * Wm[l + j * n];
    Q[i + j * ldq] = s;   // NOTE: V's strictly-lower part is stored in Q's own top block; see ungqr() for the ordering
}

// lapack::ungqr(m, n, k = n, A, lda, tau): A (m x n, reflectors below the diagonal) <- Q[:, 0:n]   (rl_orth.hh:162)
// Q E = E - V (T V1^T):  W = T V1^T (n x n x n GEMM), bottom rows  Q2 = -V2 W (m x n x n GEMM, out of place into a
// scratch copy of V2), top block by the finish kernel.
template <typename T>
int ungqr(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, const T* tau_dev) {
    if (m < 0) return -2;
    if (n < 0 || n > m) return -3;
    if (lda < (m > 1 ? m : 1)) return -6;
    if (n == 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    T* Tm = ws_alloc<T>(c, (size_t)n * n);
    T* V1t = ws_alloc<T>(c, (size_t)n * n);
    T* Wm = ws_alloc<T>(c, (size_t)n * n);
    T* Qtop = ws_alloc<T>(c, (size_t)n * n);
    T* V2 = (m > n) ? ws_alloc<T>(c, (size_t)(m - n) * n) : nullptr;
    if (!Tm || !V1t || !Wm || !Qtop || (m > n && !V2)) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    int rc = larft_gram<T>(c, m, n, A, lda, tau_dev, Tm, n);
    if (rc) { rlhip_ws_release(c, mark); return rc; }
    const unsigned nb2 = (unsigned)((n * n + 255) / 256);
    hipLaunchKernelGGL(ungqr_seed_kernel<T>, dim3(nb2), dim3(256), 0, c->stream, n, A, lda, V1t);
    RLHIP_LAUNCH_CHECK();
    // T is upper triangular but larft leaves the strictly lower part untouched: Tm was freshly laset to I before the solve
    rc = gemm<T>(c, 0, 0, n, n, n, T(1), Tm, n, V1t, n, T(0), Wm, n);
    if (!rc) {
        hipLaunchKernelGGL(ungqr_finish_kernel<T>, dim3(nb2), dim3(256), 0, c->stream, m, n, Qtop, n, A, lda, Wm);
        RLHIP_LAUNCH_CHECK();
    }
    if (!rc && m > n) {
        RLHIP_CHECK(hipMemcpy2DAsync(V2, (size_t)(m - n) * sizeof(T), A + n, (size_t)lda * sizeof(T), (size_t)(m - n) * sizeof(T), (size_t)n,
                                     hipMemcpyDeviceToDevice, c->stream));
        rc = gemm<T>(c, 0, 0, m - n, n, n, T(-1), V2, m - n, Wm, n, T(0), A + n, lda);
    }
    if (!rc)
        RLHIP_CHECK(hipMemcpy2DAsync(A, (size_t)lda * sizeof(T), Qtop, (size_t)n * sizeof(T), (size_t)n * sizeof(T), (size_t)n,
                                     hipMemcpyDeviceToDevice, c->stream));
    rlhip_ws_release(c, mark);
    return rc;
}

template <typename T>
static int qr_core(rlhip_ctx* c, int pivot, int64_t m, int64_t n, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev, int64_t max_steps,
                   int hq_formula) {
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (lda < (m > 1 ? m : 1)) return -5;
    if (m == 0 || n == 0) return 0;
    const int num_cu = c->num_cu;
    // Fewer, fatter workgroups make the two rendezvous per step cheaper; the owned columns are kept in LDS when
    // they fit (<= 150 KiB per workgroup), which also keeps the release fences of the grid barrier clean.
    int64_t G = (n + 7) / 8;           // ~8 columns (two per wave) per workgroup (4 and 16 measured: no better)
    if (G > num_cu) G = num_cu;        // wide sketches: one workgroup per CU (2560 x 2048: 94 -> 78 ms against num_cu / 2)
    if (G < 1) G = 1;
    const int64_t cols_per_wg = (n + G - 1) / G;
    size_t lds_bytes = (size_t)cols_per_wg * (size_t)m * sizeof(T);
    int use_lds = lds_bytes + (size_t)m * sizeof(T) <= 140 * 1024;
    if (!use_lds) {
        // try more workgroups before giving up on LDS residency
        int64_t G2 = num_cu;
        int64_t cpw2 = (n + G2 - 1) / G2;
        if ((size_t)(cpw2 + 1) * m * sizeof(T) <= 140 * 1024) { G = G2; lds_bytes = (size_t)cpw2 * m * sizeof(T); use_lds = 1; }
        else lds_bytes = 0;
    }
    if (!pivot && max_steps < 0) {
        size_t mark2 = rlhip_ws_mark(c);
        const int64_t kmax = m < n ? m : n;
        QrPipeArgs<T> pa;
        pa.m = m; pa.n = n; pa.A = A; pa.lda = lda; pa.tau = tau_dev; pa.use_lds = use_lds;
        pa.v_in_lds = use_lds || ((size_t)m * sizeof(T) <= 64 * 1024);
        pa.wg_per_col = 0;
        pa.chunk = 8;                          // columns dealt to a workgroup at a time (measured best of 1 / 4 / 8 / 16)
        int64_t Gp = G;
        if (!use_lds && m > 8 * n) {          // tall-skinny: spread the columns over as many workgroups as there are CUs
            Gp = n < num_cu ? n : num_cu;
            pa.wg_per_col = 1;
        }
        pa.flag = ws_alloc<unsigned>(c, (size_t)kmax + 4);
        if (!pa.flag) { rlhip_ws_release(c, mark2); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
        hipLaunchKernelGGL(zero_u32_n, dim3((unsigned)((kmax + 255) / 256)), dim3(256), 0, c->stream, pa.flag, kmax);
        // local column slots under the chunked layout; if the rounding up to whole chunks no longer fits LDS, fall back to chunk = 1
        auto slots = [&](int64_t ch) { return (size_t)((((n + ch - 1) / ch + Gp - 1) / Gp) * ch); };
        if (pa.wg_per_col) pa.chunk = 1;
        if (use_lds && (slots(pa.chunk) + 1) * (size_t)m * sizeof(T) > 150 * 1024) pa.chunk = 1;
        const size_t cpw2 = slots(pa.chunk);
        const size_t dyn2 = (pa.v_in_lds ? (size_t)m * sizeof(T) : 0) + (use_lds ? cpw2 * (size_t)m * sizeof(T) : 0);
        // every workgroup spins on flags raised by others: the grid must be co-resident.  A cooperative launch checks the grid against the
        // device's occupancy and is gang-scheduled (serialised against other processes' cooperative kernels), so a GPU shared between ranks
        // or with busy CUs cannot strand part of the grid.
        {
            void* kargs[] = {(void*)&pa};
            const void* kern = use_lds ? (const void*)qr_pipe_kernel<T, true, true>            // (columns in LDS implies the reflector in LDS)
                                       : (pa.v_in_lds ? (const void*)qr_pipe_kernel<T, false, true> : (const void*)qr_pipe_kernel<T, false, false>);
            hipError_t le = qr_kernel_lds_limit(c, kern);
            if (le == hipSuccess) le = hipLaunchCooperativeKernel(kern, dim3((unsigned)Gp), dim3(256), kargs, (unsigned)dyn2, c->stream);
            if (le != hipSuccess) { rlhip_ws_release(c, mark2); return RLHIP_ERR_HIP(le); }
        }
        rlhip_ws_release(c, mark2);
        return 0;
    }
    if (pivot && use_lds && m < ((int64_t)1 << 31) - 2 && n < ((int64_t)1 << 31) - 2) {
        // the exchange no longer pays per participant, so the columns are spread thinner than for the rendezvous kernel: 4 per workgroup
        // (rlhip_set_qrcp_cols: a caller that runs this factorization BESIDE another kernel asks for
        // fewer, fuller workgroups -- 1280 x 1024: 7.7 ms with 4 columns per workgroup, 8.7 with 8; 768 x 512: 3.5 / 3.9 / 4.9 ms with 4 / 8 / 16)
        int64_t g_env = 4;
        if (c->qrcp_cols_per_wg > 0) g_env = c->qrcp_cols_per_wg;
        int64_t Gt = (n + g_env - 1) / g_env;
        if (Gt > num_cu) Gt = num_cu;
        if (Gt > 256) Gt = 256;           // the speculation bookkeeping of the kernel holds one record per thread of a 256-thread workgroup
        if (Gt < 1) Gt = 1;
        const size_t cpw_t = (size_t)((n + Gt - 1) / Gt);
        const size_t dyn = cpw_t * sizeof(int64_t) + (2 * cpw_t + (size_t)m) * sizeof(T) + cpw_t * (size_t)m * sizeof(T);
        if (dyn <= 150 * 1024) {
            RLHIP_FUNC_LDS(c, qrcp_tag_kernel<T>, 150 * 1024);
            size_t mark = rlhip_ws_mark(c);
            QrcpTagArgs<T> t;
            t.m = m; t.n = n; t.A = A; t.lda = lda; t.tau = tau_dev;
            t.Aout = ws_alloc<T>(c, (size_t)m * n); t.ldo = m;
            t.jpvt = ws_alloc<int64_t>(c, (size_t)n);
            const size_t words = 2 * qt_words<T>(m, Gt);
            t.tw = (unsigned long long*)rlhip_xchg_buffer(c, words * sizeof(unsigned long long));
            t.info = (int*)ws_alloc<int>(c, 32);
            // (the scratch copy of the output is this path's own need: when it cannot be had, the rendezvous kernel below -- which works in
            // place -- takes the problem instead of an out-of-memory error; every error exit releases the arena mark)
            const bool have = t.tw && t.info && t.Aout && t.jpvt;
            t.tol3z = std::sqrt(std::numeric_limits<T>::epsilon() / 2);
            t.max_steps = max_steps; t.hq_formula = hq_formula;
            hipError_t te = hipSuccess;
            if (have) te = hipMemsetAsync(t.tw, 0, words * sizeof(unsigned long long), c->stream);
            if (have && te == hipSuccess) te = hipMemsetAsync(t.info, 0, sizeof(int), c->stream);
            const double kq = (double)((max_steps >= 0 && max_steps < (m < n ? m : n)) ? max_steps : (m < n ? m : n)); (void)kq;
            void* kargs[] = {(void*)&t};
            bool launched = false;
            if (have && te == hipSuccess) {
                if (hipLaunchCooperativeKernel((const void*)qrcp_tag_kernel<T>, dim3((unsigned)Gt), dim3(256), kargs, (unsigned)dyn, c->stream) == hipSuccess) launched = true;
                else (void)hipGetLastError();                          // not resident here: the rendezvous kernel decides for itself
            }
            if (launched) {
                te = hipMemcpyAsync(c->h_mail + 56, t.info, sizeof(int), hipMemcpyDeviceToHost, c->stream);
                if (te == hipSuccess) te = rlhip_stream_sync(c);
            }
            if (te != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(te); }
#ifdef RLHIP_QT_PROF
            {
                long long pf[6];
                RLHIP_CHECK(hipMemcpy(pf, t.info + 2, sizeof(pf), hipMemcpyDeviceToHost));
                fprintf(stderr, "[qrcp prof %ldx%ld G=%ld] per step (us): cand+put %.2f  spec %.2f  records %.2f  column %.2f  install %.2f  apply %.2f\n", (long)m, (long)n, (long)Gt,
                        pf[0] / 100.0 / kq, pf[1] / 100.0 / kq, pf[2] / 100.0 / kq, pf[3] / 100.0 / kq, pf[4] / 100.0 / kq, pf[5] / 100.0 / kq);
            }
#endif
            if (launched && *(int*)(c->h_mail + 56) == 0) {
                // the kernel only read A: its results are taken over now (10 MB at 1280 x 1024: microseconds)
                te = hipMemcpy2DAsync(A, (size_t)lda * sizeof(T), t.Aout, (size_t)m * sizeof(T), (size_t)m * sizeof(T), (size_t)n, hipMemcpyDeviceToDevice, c->stream);
                if (te == hipSuccess) te = hipMemcpyAsync(jpvt_dev, t.jpvt, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream);
                rlhip_ws_release(c, mark);
                return te == hipSuccess ? 0 : RLHIP_ERR_HIP(te);
            }
            // no scratch, no residency, or a published word never arrived (bounded spins).  A and jpvt are untouched: the rendezvous kernel below.
            rlhip_ws_release(c, mark);
        }
    }
    size_t mark = rlhip_ws_mark(c);
    QrcpArgs<T> g;
    g.m = m; g.n = n; g.A = A; g.lda = lda; g.jpvt = jpvt_dev; g.tau = tau_dev;
    g.cand_val = ws_alloc<T>(c, 2 * G); g.cand_pos = ws_alloc<int64_t>(c, 2 * G); g.cand_tau = ws_alloc<T>(c, 2 * G);
    g.slot = ws_alloc<T>(c, (size_t)2 * G * m); g.kcol = ws_alloc<T>(c, (size_t)2 * (m + 2));
    g.bar = ws_alloc<unsigned>(c, 4);
    g.tol3z = std::sqrt(std::numeric_limits<T>::epsilon() / 2);   // SQRT(DLAMCH('Epsilon')): LAPACK's eps is the rounding unit
    g.use_lds = use_lds; g.pivot = pivot; g.max_steps = max_steps; g.hq_formula = hq_formula;
    if (!g.cand_val || !g.cand_pos || !g.cand_tau || !g.slot || !g.kcol || !g.bar) {
        rlhip_ws_release(c, mark);
        return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    }
    hipLaunchKernelGGL(zero_u32, dim3(1), dim3(1), 0, c->stream, g.bar);
    const size_t cpw_final = (size_t)((n + G - 1) / G);
    g.v_in_lds = use_lds || ((2 * cpw_final + (size_t)m) * sizeof(T) <= 140 * 1024);
    const size_t dyn = (2 * cpw_final + (g.v_in_lds ? (size_t)m : 0)) * sizeof(T) + (use_lds ? cpw_final * (size_t)m * sizeof(T) : 0);
    if (dyn > 150 * 1024) { rlhip_ws_release(c, mark); return -2; }   // only the per-column norms left: n / G > ~9000 columns per workgroup
    {   // grid barrier inside: cooperative launch (see qr_pipe_kernel above)
        void* kargs[] = {(void*)&g};
        const void* kern = use_lds ? (const void*)qrcp_kernel<T, true, true>
                                   : (g.v_in_lds ? (const void*)qrcp_kernel<T, false, true> : (const void*)qrcp_kernel<T, false, false>);
        hipError_t le = qr_kernel_lds_limit(c, kern);
        if (le == hipSuccess) le = hipLaunchCooperativeKernel(kern, dim3((unsigned)G), dim3(256), kargs, (unsigned)dyn, c->stream);
        if (le != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(le); }
    }
    rlhip_ws_release(c, mark);
    return 0;
}

template int geqp3<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, int64_t*, double*);
template int geqp3<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, int64_t*, float*);
template int geqrf<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*);
template int geqrf<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*);
template int ungqr<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, const double*);
template int ungqr<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, const float*);

}  // namespace rlhip
