// Column-pivoted Householder QR of the (small, d x n) sketch on the device: lapack::geqp3 at
// RandLAPACK/drivers/rl_cqrrpt.hh:247 (and rl_bqrrp.hh qrcp_wide = geqp3 option).  This call DEFINES the pivots,
// so the arithmetic follows LAPACK's dlaqp2 step by step (first-maximum pivot search over the partial column
// norms, dlarfg reflector, the |A(k,j)|/vn1(j) norm down-date with the sqrt(eps) recomputation safeguard);
// given the same sketch the pivot vector equals LAPACK's except on exact near-ties of partial norms.
//
// Execution model: ONE persistent launch.  Column position j lives with workgroup j % G (cyclic, so the load
// stays balanced as the factorization advances).  Per step there are two grid-wide rendezvous:
//   R1  every workgroup has published its best local candidate (norm, position)        -> pivot p is known
//   R2  the owner of p has turned that column into the Householder vector (dlarfg) and published it,
//       the owner of k has published the column that moves to position p              -> everybody updates
// after R2 each wave applies H to whole columns it owns (wavefront DPP reductions, no block barrier),
// down-dates their norms and the workgroup publishes its candidate for the next step.
// The grid barrier is a monotonic counter with agent-scope release/acquire (guide section 6, G16); the
// grid is sized to the CU count so all workgroups are co-resident.
#include "rlhip_internal.h"
#include <cmath>
#include <limits>

namespace {

template <typename T>
struct QrcpArgs {
    int64_t m, n;
    T* A; int64_t lda;
    int64_t* jpvt;            // device, 1-based on exit; entries != 0 on entry are NOT treated as fixed (caller passes zeros)
    T* tau;
    T* vn1; T* vn2;           // n each
    T* cand_val; int64_t* cand_pos;   // G each
    T* pcol; T* kcol;         // m each
    T* scal;                  // [0] = tau_k
    unsigned* bar;            // barrier counter (zeroed by the host)
    T tol3z;
};

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
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
    const int64_t kmax = m < n ? m : n;
    __shared__ T s_val[4];
    __shared__ int64_t s_pos[4];
    unsigned epoch = 0;

    // ---- initial column norms + jpvt + first candidates (columns me, me+G, ...; one wave per column)
    for (int64_t j = me + G * wid; j < n; j += 4 * G) {
        const T* col = g.A + j * g.lda;
        T ss = 0;
        for (int64_t i = lane; i < m; i += 64) ss += col[i] * col[i];
        ss = wave_sum(ss);
        if (lane == 0) { T nr = sqrt(ss); g.vn1[j] = nr; g.vn2[j] = nr; g.jpvt[j] = j + 1; }
    }
    __syncthreads();

    for (int64_t k = 0; k < kmax; ++k) {
        // ---- publish local candidate over owned positions >= k (first maximum: smallest position wins ties)
        {
            T best = T(-1); int64_t bpos = n;
            for (int64_t j = me; j < n; j += G) {
                if (j < k) continue;
                T v = g.vn1[j];
                if (v > best) { best = v; bpos = j; }   // increasing j: strict > keeps the first maximum
            }
            // the scan above is done redundantly by every thread (few columns per workgroup)
            if (tid == 0) { g.cand_val[me] = best; g.cand_pos[me] = bpos; }
        }
        grid_barrier(g.bar, (unsigned)(G * (++epoch)));                                           // R1
        // ---- global pivot (every workgroup, redundantly)
        T pbest = T(-1); int64_t p = n;
        for (int64_t w = 0; w < G; ++w) {
            T v = g.cand_val[w]; int64_t q = g.cand_pos[w];
            if (v > pbest || (v == pbest && q < p)) { pbest = v; p = q; }
        }
        if (p >= n) p = k;   // all remaining norms are NaN/negative: keep the natural order
        const int64_t own_p = p % G, own_k = k % G;
        // ---- owner of p: build the reflector from column p (it becomes column k) and publish it
        if (me == own_p) {
            T* col = g.A + p * g.lda;
            // xnorm over rows k+1..m-1
            T ss = 0;
            for (int64_t i = k + 1 + tid; i < m; i += 256) ss += col[i] * col[i];
            ss = wave_sum(ss);
            if (lane == 0) s_val[wid] = ss;
            __syncthreads();
            const T xnorm = sqrt(s_val[0] + s_val[1] + s_val[2] + s_val[3]);
            const T alpha = col[k];
            T tauk = 0, beta = alpha, scale = 0;
            if (xnorm != T(0)) {                                    // dlarfg (without the safmin rescaling loop)
                beta = -copysign(hypot(alpha, xnorm), alpha);
                tauk = (beta - alpha) / beta;
                scale = T(1) / (alpha - beta);
            }
            __syncthreads();
            for (int64_t i = tid; i < m; i += 256) {
                T v = col[i];
                if (i == k) v = beta; else if (i > k) v *= scale;
                g.pcol[i] = v;
            }
            if (tid == 0) { g.scal[0] = tauk; g.tau[k] = tauk; }
        }
        if (me == own_k && p != k) {
            const T* col = g.A + k * g.lda;
            for (int64_t i = tid; i < m; i += 256) g.kcol[i] = col[i];
        }
        grid_barrier(g.bar, (unsigned)(G * (++epoch)));                                           // R2
        // ---- install the moved columns, swap bookkeeping
        if (me == own_k) {
            T* col = g.A + k * g.lda;
            for (int64_t i = tid; i < m; i += 256) col[i] = g.pcol[i];
        }
        if (p != k && me == own_p) {
            T* col = g.A + p * g.lda;
            for (int64_t i = tid; i < m; i += 256) col[i] = g.kcol[i];
            if (tid == 0) {
                g.vn1[p] = g.vn1[k]; g.vn2[p] = g.vn2[k];
                int64_t t = g.jpvt[p]; g.jpvt[p] = g.jpvt[k]; g.jpvt[k] = t;
            }
        }
        __syncthreads();
        // ---- apply H = I - tau v v^T (v_k = 1, v below from pcol) to owned columns j > k, one wave per column
        const T tauk = g.scal[0];
        for (int64_t j = me + G * wid; j < n; j += 4 * G) {
            if (j <= k) continue;
            T* col = g.A + j * g.lda;
            if (tauk != T(0)) {
                T w = 0;
                for (int64_t i = k + lane; i < m; i += 64) w += ((i == k) ? T(1) : g.pcol[i]) * col[i];
                w = wave_sum(w) * tauk;
                for (int64_t i = k + lane; i < m; i += 64) col[i] -= w * ((i == k) ? T(1) : g.pcol[i]);
            }
            // norm down-date (dlaqp2): vn1(j) *= sqrt(max(0, 1 - (|A(k,j)|/vn1(j))^2)) with recomputation safeguard
            T v1 = g.vn1[j];
            if (v1 != T(0)) {
                T akj = fabs(col[k]);
                T r = akj / v1;
                T temp = T(1) - r * r;
                temp = temp > T(0) ? temp : T(0);
                T q = v1 / g.vn2[j];
                T temp2 = temp * q * q;
                if (temp2 <= g.tol3z) {
                    T ss = 0;
                    for (int64_t i = k + 1 + lane; i < m; i += 64) ss += col[i] * col[i];
                    ss = wave_sum(ss);
                    v1 = sqrt(ss);
                    if (lane == 0) { g.vn1[j] = v1; g.vn2[j] = v1; }
                } else {
                    if (lane == 0) g.vn1[j] = v1 * sqrt(temp);
                }
            }
        }
        __syncthreads();
    }
    // tau for any remaining min(m,n) entries is already written; nothing else to do
}

__global__ void zero_u32(unsigned* p) { *p = 0; }

}  // namespace

namespace rlhip {

template <typename T>
int geqp3(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev) {
    if (m < 0) return -2;
    if (n < 0) return -3;
    if (lda < (m > 1 ? m : 1)) return -5;
    if (m == 0 || n == 0) return 0;
    static int num_cu = 0;
    if (!num_cu) {
        hipDeviceProp_t prop;
        num_cu = (hipGetDeviceProperties(&prop, c->device) == hipSuccess) ? prop.multiProcessorCount : 256;
        if (num_cu <= 0) num_cu = 256;
    }
    int64_t G = (n + 3) / 4;           // ~4 columns (one per wave) per workgroup
    if (G > num_cu) G = num_cu;        // co-residency: one workgroup per CU at most
    if (G < 1) G = 1;
    size_t mark = rlhip_ws_mark(c);
    QrcpArgs<T> g;
    g.m = m; g.n = n; g.A = A; g.lda = lda; g.jpvt = jpvt_dev; g.tau = tau_dev;
    g.vn1 = ws_alloc<T>(c, n); g.vn2 = ws_alloc<T>(c, n);
    g.cand_val = ws_alloc<T>(c, G); g.cand_pos = ws_alloc<int64_t>(c, G);
    g.pcol = ws_alloc<T>(c, m); g.kcol = ws_alloc<T>(c, m);
    g.scal = ws_alloc<T>(c, 4);
    g.bar = ws_alloc<unsigned>(c, 4);
    g.tol3z = std::sqrt(std::numeric_limits<T>::epsilon());
    if (!g.vn1 || !g.vn2 || !g.cand_val || !g.cand_pos || !g.pcol || !g.kcol || !g.scal || !g.bar) {
        rlhip_ws_release(c, mark);
        return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    }
    hipLaunchKernelGGL(zero_u32, dim3(1), dim3(1), 0, c->stream, g.bar);
    hipLaunchKernelGGL(qrcp_kernel<T>, dim3((unsigned)G), dim3(256), 0, c->stream, g);
    RLHIP_LAUNCH_CHECK();
    rlhip_ws_release(c, mark);
    return 0;
}

template int geqp3<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, int64_t*, double*);
template int geqp3<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, int64_t*, float*);

}  // namespace rlhip
