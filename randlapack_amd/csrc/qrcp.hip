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

template <typename T>
__global__ __launch_bounds__(256) void qrcp_kernel(QrcpArgs<T> g) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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
    T* lds_cols = l_v + (g.v_in_lds ? m : 0);
    __shared__ T s_cval[256];
    __shared__ int64_t s_cpos[256];
    __shared__ int s_cw[256];
    // column position j -> storage (generic pointer: LDS slot j/G of the owner, or the matrix itself)
    auto colptr = [&](int64_t j) -> T* { return g.use_lds ? (lds_cols + (j / G) * m) : (g.A + j * g.lda); };
    if (g.use_lds) {
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
        if (g.v_in_lds) {
            for (int64_t i = tid; i < m; i += 256) l_v[i] = vcol[i];
            __syncthreads();
        }
        const T* vv = g.v_in_lds ? l_v : vcol;       // tall inputs: straight from the published slot (L2)
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
                T w = 0;
                for (int64_t i = k + lane; i < m; i += 64) w += ((i == k) ? T(1) : vv[i]) * col[i];
                w = wave_sum(w) * tauk;
                for (int64_t i = k + lane; i < m; i += 64) col[i] -= w * ((i == k) ? T(1) : vv[i]);
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
    if (g.use_lds) {
        __syncthreads();
        for (int64_t j = me; j < n; j += G) {
            const T* src = lds_cols + (j / G) * m;
            T* dst = g.A + j * g.lda;
            for (int64_t i = tid; i < m; i += 256) dst[i] = src[i];
        }
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

template <typename T>
__global__ __launch_bounds__(256) void qr_pipe_kernel(QrPipeArgs<T> g) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t G = gridDim.x, me = blockIdx.x;
    const int64_t m = g.m, n = g.n;
    const int64_t kmax = m < n ? m : n;
    extern __shared__ __attribute__((aligned(16))) unsigned char qp_smem[];
    T* l_v = reinterpret_cast<T*>(qp_smem);              // current reflector (m), when it fits
    T* lds_cols = l_v + (g.v_in_lds ? m : 0);
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
    auto colptr = [&](int64_t j) -> T* { return g.use_lds ? (lds_cols + slot(j) * m) : (g.A + j * g.lda); };
    if (g.use_lds) {
        for (int64_t c = 0; c < cpw; ++c) {
            const int64_t j = col_of(c);
            if (j >= n) break;
            T* dst = lds_cols + c * m;
            const T* src = g.A + j * g.lda;
            for (int64_t i = tid; i < m; i += 256) dst[i] = src[i];
        }
        __syncthreads();
    }
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
            if (g.v_in_lds) l_v[i] = v;
            if (g.use_lds) col[i] = v;
            pub_store(gcol + i, v);                       // final content of column k of A (R above, beta, v below)
        }
        if (tid == 0) { pub_store(g.tau + k, tk); s_tau = tk; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(g.flag + k, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            // tall matrices read v back from its column of A with ordinary loads: drop this CU's stale L1 lines of that column
            if (!g.v_in_lds) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (!g.v_in_lds) __syncthreads();
    };
    // apply H (v in l_v, tau = tk, pivot row k) to column j
    auto apply_one = [&](int64_t k, T tk, T* col) {       // one wave per column
        const T* gv = g.A + k * g.lda;                    // (tall matrices: v straight from its published column)
        T w = 0;
#pragma unroll 4
        for (int64_t i = k + lane; i < m; i += 64) w += ((i == k) ? T(1) : (g.v_in_lds ? l_v[i] : gv[i])) * col[i];
        w = wave_sum(w) * tk;
#pragma unroll 4
        for (int64_t i = k + lane; i < m; i += 64) col[i] -= w * ((i == k) ? T(1) : (g.v_in_lds ? l_v[i] : gv[i]));
    };
    auto apply_wg = [&](int64_t k, T tk, T* col) {        // whole workgroup on one (long) column
        const T* gv = g.A + k * g.lda;
        T w = 0;
#pragma unroll 8
        for (int64_t i = k + tid; i < m; i += 256) w += ((i == k) ? T(1) : (g.v_in_lds ? l_v[i] : gv[i])) * col[i];
        w = wave_sum(w);
        __syncthreads();
        if (lane == 0) s_val[wid] = w;
        __syncthreads();
        w = (s_val[0] + s_val[1] + s_val[2] + s_val[3]) * tk;
#pragma unroll 8
        for (int64_t i = k + tid; i < m; i += 256) col[i] -= w * ((i == k) ? T(1) : (g.v_in_lds ? l_v[i] : gv[i]));
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
            if (g.v_in_lds)
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
    if (g.use_lds) {
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
static int qr_core(rlhip_ctx* c, int pivot, int64_t m, int64_t n, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev, int64_t max_steps = -1,
                   int hq_formula = 0);

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
template int qrp_partial<double>(rlhip_ctx*, int64_t, int64_t, int64_t, double*, int64_t, int64_t*, double*);
template int qrp_partial<float>(rlhip_ctx*, int64_t, int64_t, int64_t, float*, int64_t, int64_t*, float*);

// lapack::geqrf: Householder QR without pivoting.  Wide input (n > m, BQRRP's permuted sketch rl_bqrrp.hh:356): only the
// leading m x m block goes through the step-synchronous kernel; the remaining columns get Q^T applied as ONE compact-WY
// block (larft + gemqrt on the MFMA path).
template <typename T>
int geqrf(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau_dev) {
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
    static int64_t pipe_max = -1;
    if (pipe_max < 0) { const char* e = getenv("RLHIP_GEQRF_PIPE_MAX"); pipe_max = e ? atoll(e) : 1280; }
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
    RLHIP_FUNC_LDS(c, qrcp_kernel<T>, 150 * 1024);
    RLHIP_FUNC_LDS(c, qr_pipe_kernel<T>, 150 * 1024);
    static int pipe_on = -1;
    if (pipe_on < 0) { const char* e = getenv("RLHIP_QR_PIPE"); pipe_on = (e && atoi(e) == 0) ? 0 : 1; }
    if (!pivot && max_steps < 0 && pipe_on) {
        size_t mark2 = rlhip_ws_mark(c);
        const int64_t kmax = m < n ? m : n;
        QrPipeArgs<T> pa;
        pa.m = m; pa.n = n; pa.A = A; pa.lda = lda; pa.tau = tau_dev; pa.use_lds = use_lds;
        pa.v_in_lds = use_lds || ((size_t)m * sizeof(T) <= 64 * 1024);
        pa.wg_per_col = 0;
        static int chunk_env = -1;
        if (chunk_env < 0) { const char* e = getenv("RLHIP_QR_CHUNK"); chunk_env = e ? atoi(e) : 8; if (chunk_env < 1) chunk_env = 1; }
        pa.chunk = chunk_env;
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
            RLHIP_CHECK(hipLaunchCooperativeKernel((const void*)qr_pipe_kernel<T>, dim3((unsigned)Gp), dim3(256), kargs, (unsigned)dyn2, c->stream));
        }
        rlhip_ws_release(c, mark2);
        return 0;
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
        RLHIP_CHECK(hipLaunchCooperativeKernel((const void*)qrcp_kernel<T>, dim3((unsigned)G), dim3(256), kargs, (unsigned)dyn, c->stream));
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
