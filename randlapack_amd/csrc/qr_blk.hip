// Unpivoted Householder QR of a sketch-sized, nearly square matrix (up to 2048 rows, up to 8 columns per CU) in ONE cooperative launch:
// lapack::geqrf on BQRRP's permuted sketch (rl_bqrrp.hh:356; 2048 x 2048 of the 2048 x cols fp32 sketch at BASELINE configs[3]) and on the
// other d x n sketches of the path.
//
// The flag-pipelined kernel of qrcp.hip hands ONE reflector at a time from workgroup to workgroup (5.9 us per column: every workgroup has
// to take in, and apply, every single reflector), the blocked host loop pays ~1.5 ms of small launches per 256 columns.  Here
//   * workgroup w owns the 8-column chunk w of the matrix and keeps it in REGISTERS for the whole launch: thread t holds rows t, t + NT, ...
//     of all eight columns (no LDS image of the matrix at all: LDS only carries the reductions);
//   * chunk k is factored by its owner alone -- one reduction round per column: the raw inner products of column cc with all eight
//     columns of the chunk give its norm (k = cc), the reflector's action on the columns to its right (k > cc) and the entries of the
//     triangular factor T of the chunk's compact-WY form (k < cc) at once -- and is published as a BLOCK reflector: V in place in A
//     (write-through stores), T (8 x 8) and tau next to a flag;
//   * every later chunk applies it as  C -= V (T^T (V^T C)):  the thread reads its rows of V straight from A (L2-resident), the 8 x 8
//     product V^T C is reduced over the wave by a transposing butterfly (64 sums for 63 exchanges) and over the waves through LDS.
// The critical path is one chunk factorization + one hand-over + one block application per 8 columns instead of per column.
// Same reflectors as LAPACK's geqr2 / larfg (beta = -sign(alpha) ||x||, tau = (beta - alpha) / beta, v = x / (alpha - beta)), same T as larft.
#include "rlhip_internal.h"
#include <cstdlib>
#include <cstdio>

namespace {

template <typename T>
struct QbArgs {
    int64_t m, n;             // n <= m columns are factored, n <= 8 * gridDim.x
    T* A; int64_t lda;
    T* tau;
    T* Tx;                    // gridDim.x blocks of 64: the 8 x 8 upper triangular T of every chunk, column-major
    unsigned* flag;           // gridDim.x entries, zeroed by the host
#ifdef RLHIP_QB_PROF
    unsigned long long* prof; // [0] last block application, [1] chunk factorization, [2] publication (wall-clock ticks, summed over the workgroups)
#endif
};

template <typename T>
__device__ __forceinline__ void qb_pub(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The same write-through (sc1) store, but issued without the s_waitcnt vmcnt(0) hipcc puts in front of every agent-scope atomic store: a
// thread's 32 stores of a chunk would otherwise go out one memory round trip at a time.  The workgroup drains them once, before its flag.
__device__ __forceinline__ void qb_store_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void qb_store_wt(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// value of lane ^ D for D = 1, 2, 4, 8 on the VALU (DPP quad permutations / row rotations; no LDS crossbar round trip)
template <int D>
__device__ __forceinline__ int qb_fetch_i(int x) {
    if constexpr (D == 1) return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);            // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);       // quad_perm [2,3,0,1]
    else if constexpr (D == 4)                                                                          // row_half_mirror (l ^ 7) then quad_perm [3,2,1,0] (l ^ 3)
        return __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false), 0x1B, 0xF, 0xF, false);
    else return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false);                             // row_ror:8
}
template <int D> __device__ __forceinline__ float qb_fetch(float v) { return __int_as_float(qb_fetch_i<D>(__float_as_int(v))); }
template <int D> __device__ __forceinline__ double qb_fetch(double v) {
    return __hiloint2double(qb_fetch_i<D>(__double2hiint(v)), qb_fetch_i<D>(__double2loint(v)));
}
// workgroup rendezvous that orders LDS traffic only: the write-through stores of finished columns stay in flight across it
// (__syncthreads() would drain them: its fence waits for vmcnt(0))
__device__ __forceinline__ void qb_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// value held by lane l (compile-time constant after unrolling): v_readlane, a scalar
__device__ __forceinline__ float qb_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double qb_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// One exchange step across lane bit log2(D): lanes with the bit clear receive  a(lane) + a(lane ^ D), lanes with it set  b(lane) + b(lane ^ D).
// D = 16, 32: v_permlane16_swap / v_permlane32_swap (gfx950) trade the odd rows (upper half) of the first operand for the even rows (lower half)
// of the second one, which is exactly this step in one instruction per 32 bits; D <= 8: select + DPP fetch.
template <int D>
__device__ __forceinline__ float qb_pair_sum(float a, float b, bool up) {
    if constexpr (D == 32) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    else if constexpr (D == 16) { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false); return __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    else { const float keep = up ? b : a, send = up ? a : b; return keep + qb_fetch<D>(send); }
}
template <int D>
__device__ __forceinline__ double qb_pair_sum(double a, double b, bool up) {
    if constexpr (D >= 16) {
        const unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a), bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
        if constexpr (D == 32) {
            auto rl = __builtin_amdgcn_permlane32_swap(al, bl, false, false); auto rh = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
            return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
        } else {
            auto rl = __builtin_amdgcn_permlane16_swap(al, bl, false, false); auto rh = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
            return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
        }
    } else { const double keep = up ? b : a, send = up ? a : b; return keep + qb_fetch<D>(send); }
}

// Sums of N per-lane values over the 64 lanes of a wave by a transposing butterfly: every exchange halves the number of values a lane
// still carries, so N sums cost N - 1 exchanges (+ log2(64 / N) plain butterfly steps when N < 64) instead of 6 N.  Returns the wave-wide
// sum of entry  lane / (64 / N).
template <typename T, int N, int D, int STOP = 0>
struct QbRed {
    static __device__ __forceinline__ T run(T (&x)[N], int lane) {
        if constexpr (D == STOP) return x[0];
        else if constexpr (N == 1) {
            T y[1] = {qb_pair_sum<D>(x[0], x[0], (lane & D) != 0)};
            return QbRed<T, 1, D / 2, STOP>::run(y, lane);
        } else {
            const bool up = (lane & D) != 0;
            T y[N / 2];
#pragma unroll
            for (int k = 0; k < N / 2; ++k) y[k] = qb_pair_sum<D>(x[k], x[k + N / 2], up);
            return QbRed<T, N / 2, D / 2, STOP>::run(y, lane);
        }
    }
};

// NT threads, RPT rows per thread (NT * RPT >= m), IW = columns of V per reduction pass (8 / IW passes per block: the 8 x 8 product of a pass
// lives in IW * 8 registers))
template <typename T, int NT, int RPT, int IW>
__global__ __launch_bounds__(NT) void qr_blk_kernel(QbArgs<T> g) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    constexpr int NW = NT / 64, NP = 8 / IW, NV = IW * 8, LPE = 64 / NV;
    static_assert(NW <= 8, "the column rounds hand one partial sum to every lane of a wave");
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int me = blockIdx.x;
    const int64_t m = g.m, lda = g.lda;
    __shared__ T s_red[NP][NW][NV];
    __shared__ T s_Wp[64];
    __shared__ T s_col[2][8][8];
    __shared__ T s_bc[2][8];
    __shared__ T s_tau[8], s_z[8][8];
    const int j0m = me * 8;
    const int cw = (int)((g.n - j0m < 8) ? (g.n - j0m) : 8);
    const int mi = (int)m;                                 // (m <= NT * RPT <= 2048: row indices are 32-bit throughout)
    T c[RPT][8];
    T* rowp[RPT];                                          // &A[r][0] of the thread's rows (row NT q + tid; clamped for the rows that do not exist)
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + NT * q;
        rowp[q] = g.A + (r < mi ? r : mi - 1);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) c[q][cc] = (r < mi && cc < cw) ? rowp[q][(int64_t)(j0m + cc) * lda] : T(0);
    }
    if constexpr (NW < 8) {                               // (rows of waves that do not exist stay zero)
        if (tid < 2 * 64) (&s_col[0][0][0])[tid] = T(0);
        __syncthreads();
    }
#ifdef RLHIP_QB_PROF
    long long t_seen = wall_clock64();
#endif
    // rows of block reflector k held by this thread / entries T(l, i), l = 0..7, of its factor held by thread 8 i + j < 64
    auto load_block = [&](int k, T (&v)[RPT][8], T (&tk)[8]) {
        const int j0 = k * 8;
        const int64_t cb = (int64_t)j0 * lda;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const T* vp = rowp[q] + cb;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[q][i] = vp[i * lda];    // (unconditional: the clamped row pointer keeps it in bounds; all 32 in flight together)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + NT * q;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[q][i] = (r > j0 + i && r < mi) ? v[q][i] : ((r == j0 + i) ? T(1) : T(0));
        }
        if (tid < 64) {
            const T* Tk = g.Tx + (int64_t)k * 64 + 8 * (tid >> 3);
#pragma unroll
            for (int l = 0; l < 8; ++l) tk[l] = Tk[l];
        }
    };
    // ---- the block reflectors of the chunks to the left, in order
    T v[RPT][8], tk[8];
    for (int k = 0; k < me; ++k) {
        if (tid == 0) {
            while (__hip_atomic_load(g.flag + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
#ifdef RLHIP_QB_PROF
        t_seen = wall_clock64();
#endif
        load_block(k, v, tk);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            T w[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) w[e] = T(0);
#pragma unroll
            for (int q = 0; q < RPT; ++q)
#pragma unroll
                for (int il = 0; il < IW; ++il)
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[il * 8 + j] += v[q][p * IW + il] * c[q][j];
            const T s = QbRed<T, NV, 32>::run(w, lane);
            if ((lane % LPE) == 0) s_red[p][wid][lane / LPE] = s;
        }
        __syncthreads();
        if (tid < 64) {                                    // W = V^T C, entry (i, j) in lane 8 i + j of the first wave;  W' = T^T W without leaving the wave
            const int p = tid / NV, e = tid % NV;
            T ws = T(0);
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) ws += s_red[p][w2][e];
            T acc = T(0);
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const T wl = __shfl(ws, 8 * l + (lane & 7), 64);         // W(l, j)
                acc += (l <= (lane >> 3)) ? tk[l] * wl : T(0);           // T(l, i), l <= i
            }
            s_Wp[tid] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            T wp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wp[j] = s_Wp[i * 8 + j];
#pragma unroll
            for (int q = 0; q < RPT; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) c[q][j] -= v[q][i] * wp[j];
        }
    }
#ifdef RLHIP_QB_PROF
    const long long t_applied = wall_clock64();
#endif
    // (every load of this thread has landed: said here, once, so that hipcc does not place a conservative s_waitcnt vmcnt(0) in front of each
    // first use below -- those would drain the write-through stores the rounds leave in flight)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // ---- this chunk: eight Householder steps, one reduction round each
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        if (cc < cw) {
            const int j = j0m + cc;
            const int par = cc & 1;
            T d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = T(0);
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + NT * q;
                const T x = (r > j && r < mi) ? c[q][cc] : T(0);           // rows below the diagonal only (branch-free)
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] += x * c[q][k];
                if (r == j) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) s_bc[par][k] = c[q][k];
                }
            }
            const T s = QbRed<T, 8, 32>::run(d, lane);    // lanes 8 e .. 8 e + 7 hold entry e
            if ((lane & 7) == 0) s_col[par][wid][lane >> 3] = s;
            qb_barrier_lds();
            // lane l of every wave takes the partial of wave l >> 3 for entry l & 7; three exchange steps leave the total of entry l & 7 in every lane
            T tot[1] = {s_col[par][lane >> 3][lane & 7]};
            const T totv = QbRed<T, 1, 32, 4>::run(tot, lane);     // exchange steps across lane bits 5, 4, 3 only
            T ds[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) ds[k] = qb_lane(totv, k);
            const T alpha = s_bc[par][cc];
            T beta = alpha, tcc = T(0), scale = T(0);
            if (ds[cc] != T(0)) {
                // (ds[cc] is a plain sum of squares already: the scaled hypot of larfg / lapy2 would protect nothing here; the exponent range
                // is the business of rlhip::geqrf, which hands this kernel a matrix whose largest entry lies in [1, 2))
                const T nrm = sqrt(alpha * alpha + ds[cc]);
                beta = (alpha >= T(0)) ? -nrm : nrm;
                tcc = (beta - alpha) / beta;
                scale = T(1) / (alpha - beta);
            }
            T wk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wk[k] = tcc * (s_bc[par][k] + scale * ds[k]);
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + NT * q;
                // v_r below the diagonal, 1 on it, 0 above: one formula for every row
                const T vv = (r > j && r < mi) ? c[q][cc] * scale : ((r == j) ? T(1) : T(0));
#pragma unroll
                for (int k = cc + 1; k < 8; ++k) c[q][k] -= vv * wk[k];
                c[q][cc] = (r > j) ? vv : ((r == j) ? beta : c[q][cc]);
                // column cc is final (R above the diagonal, beta on it, v below): it leaves now, and its stores drain behind the remaining rounds
                if (r < mi) qb_store_wt(rowp[q] + (int64_t)j * lda, c[q][cc]);
            }
            if (tid == 0) {
                s_tau[cc] = tcc;
#pragma unroll
                for (int i = 0; i < cc; ++i) s_z[i][cc] = s_bc[par][i] + scale * ds[i];      // V_i^T v_cc
            }
        }
    }
#ifdef RLHIP_QB_PROF
    const long long t_factored = wall_clock64();
#endif
    // ---- publish T and tau, then -- once every store of this workgroup has landed -- the flag
    __syncthreads();                                      // s_tau, s_z complete
    if (tid < 8) {                                        // row tid of T (larft, forward columnwise): T(i, cc) = -tau_cc sum_{l=i}^{cc-1} T(i, l) z(l, cc)
        const int i = tid;
        T row[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            T val = T(0);
            if (cc < cw) {
                if (cc == i) val = s_tau[cc];
                else if (cc > i) {
                    T a = T(0);
#pragma unroll
                    for (int l = 0; l < 8; ++l)
                        if (l >= i && l < cc) a += row[l] * s_z[l][cc];
                    val = -s_tau[cc] * a;
                }
            }
            row[cc] = val;
            qb_pub(g.Tx + (int64_t)me * 64 + i + 8 * cc, val);
        }
        if (i < cw) qb_pub(g.tau + j0m + i, s_tau[i]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(g.flag + me, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#ifdef RLHIP_QB_PROF
    if (tid == 0) {
        const long long t_pub = wall_clock64();
        atomicAdd(g.prof + 0, (unsigned long long)(t_applied - t_seen));
        atomicAdd(g.prof + 1, (unsigned long long)(t_factored - t_applied));
        atomicAdd(g.prof + 2, (unsigned long long)(t_pub - t_factored));
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same ownership for the sign-modified LU WITHOUT pivoting of Householder reconstruction (lapack::orhr_col -> dlaorhr_col_getrfnp,
// rl_bqrrp.hh:480 / rl_cqrrt.hh): for i: D(i) = -sign(a_ii), a_ii -= D(i), column below /= a_ii, trailing -= column * row.
// No pivot search, so a column step needs NO reduction at all -- the thread that holds the diagonal row broadcasts it through LDS -- and
// a block application is  U_kc = L_kk^-1 C(block rows),  C(below) -= L_k U_kc  with the 8 x 8 pieces handed over by the eight threads
// that hold the block's rows.  2048 x 2048 fp32 took 64 panels x (panel kernel + trsm + GEMM) = 5.5 ms of launches before.
template <typename T>
struct LbArgs {
    int64_t n;                // n x n block, n <= 8 * gridDim.x
    T* A; int64_t lda;
    T* D;                     // n signs
    unsigned* flag;
};

template <typename T, int NT, int RPT>
__global__ __launch_bounds__(NT) void lunp_blk_kernel(LbArgs<T> g) {
    __builtin_amdgcn_s_setprio(3);          // latency-bound: when a look-ahead runs this beside a GEMM on the same CUs, its waves issue first
    const int tid = threadIdx.x;
    const int me = blockIdx.x;
    const int64_t lda = g.lda;
    const int ni = (int)g.n;
    __shared__ T s_piv[2][8];
    __shared__ T s_d[8];
    __shared__ T s_x[8][8], s_l[8][8], s_u[8][8];
    const int j0m = me * 8;
    const int cw = (ni - j0m < 8) ? (ni - j0m) : 8;
    T c[RPT][8];
    T* rowp[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + NT * q;
        rowp[q] = g.A + (r < ni ? r : ni - 1);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) c[q][cc] = (r < ni && cc < cw) ? rowp[q][(int64_t)(j0m + cc) * lda] : T(0);
    }
    // ---- the blocks to the left, in order
    for (int k = 0; k < me; ++k) {
        const int j0 = k * 8;
        if (tid == 0) {
            while (__hip_atomic_load(g.flag + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        T lv[RPT][8];
        const int64_t cb = (int64_t)j0 * lda;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const T* lp = rowp[q] + cb;
#pragma unroll
            for (int i = 0; i < 8; ++i) lv[q][i] = lp[i * lda];
        }
        __builtin_amdgcn_sched_barrier(0);
        // the eight threads that hold rows j0 .. j0 + 7 hand over their rows of this chunk (X) and of the block (L_kk below its diagonal)
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + NT * q;
            if (r >= j0 && r < j0 + 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s_x[r - j0][j] = c[q][j]; s_l[r - j0][j] = lv[q][j]; }
            }
        }
        __syncthreads();
        if (tid < 8) {                                     // column tid of U_kc = L_kk^-1 X (unit lower: forward substitution)
            T u[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                T a = s_x[i][tid];
#pragma unroll
                for (int l = 0; l < 8; ++l)
                    if (l < i) a -= s_l[i][l] * u[l];
                u[i] = a;
                s_u[i][tid] = a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + NT * q;
            const bool below = (r >= j0 + 8) && (r < ni);
            const bool inblk = (r >= j0) && (r < j0 + 8);
            const int ri = inblk ? (r - j0) : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                T a = c[q][j];
#pragma unroll
                for (int i = 0; i < 8; ++i) a -= (below ? lv[q][i] : T(0)) * s_u[i][j];
                c[q][j] = inblk ? s_u[ri][j] : a;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // (see qr_blk_kernel: no conservative vmcnt(0) in front of the rounds' stores)
    // ---- this chunk: eight elimination steps, one rendezvous each, no reduction
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
        if (cc < cw) {
            const int j = j0m + cc;
            const int par = cc & 1;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + NT * q;
                if (r == j) {
                    const T a = c[q][cc];
                    const T dd = (a == T(0)) ? T(1) : ((a > T(0)) ? T(-1) : T(1));
                    c[q][cc] = a - dd;
                    s_d[cc] = dd;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s_piv[par][k] = c[q][k];
                }
            }
            qb_barrier_lds();
            const T piv = s_piv[par][cc];
            T urow[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) urow[k] = s_piv[par][k];
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = tid + NT * q;
                const bool act = (r > j) && (r < ni);
                const T l = act ? c[q][cc] / piv : T(0);
#pragma unroll
                for (int k = cc + 1; k < 8; ++k) c[q][k] -= l * urow[k];
                c[q][cc] = act ? l : c[q][cc];
                if (r < ni) qb_store_wt(rowp[q] + (int64_t)j * lda, c[q][cc]);
            }
        }
    }
    __syncthreads();
    if (tid < cw) qb_pub(g.D + j0m + tid, s_d[tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(g.flag + me, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

namespace rlhip {

// Householder QR of the leading n <= m columns of A (m x n, column-major) in the geqrf output format.  Returns 1 when the problem was
// factored here, 0 when it does not fit this kernel (the caller carries on with its other routes), < 0 on error.
template <typename T>
int geqrf_blk(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau_dev) {
    constexpr int NT = 512, RPT = 4, IW = (sizeof(T) == 8) ? 4 : 8;
    if (n > m || n < 1 || m > (int64_t)NT * RPT) return 0;
    const int64_t G = (n + 7) / 8;
    if (G > c->num_cu) return 0;                           // one chunk per workgroup, one workgroup per CU: all of them are resident
    size_t mark = rlhip_ws_mark(c);
    QbArgs<T> g;
    g.m = m; g.n = n; g.A = A; g.lda = lda; g.tau = tau_dev;
    g.Tx = ws_alloc<T>(c, (size_t)G * 64);
    g.flag = ws_alloc<unsigned>(c, (size_t)G + 4);
    if (!g.Tx || !g.flag) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    {
        const hipError_t me = hipMemsetAsync(g.flag, 0, (size_t)G * sizeof(unsigned), c->stream);
        if (me != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(me); }
    }
#ifdef RLHIP_QB_PROF
    g.prof = (unsigned long long*)ws_alloc<double>(c, 4);
    RLHIP_CHECK(hipMemsetAsync(g.prof, 0, 4 * sizeof(double), c->stream));
#endif
    // every workgroup waits for flags raised by others: the grid must be co-resident (cooperative launch: checked against the device's
    // occupancy and gang-scheduled)
    void* kargs[] = {(void*)&g};
    if (hipLaunchCooperativeKernel((const void*)qr_blk_kernel<T, NT, RPT, IW>, dim3((unsigned)G), dim3(NT), kargs, 0, c->stream) != hipSuccess) {
        (void)hipGetLastError();                           // the grid cannot be made resident here (shared or partitioned device): the caller's other routes
        rlhip_ws_release(c, mark);
        return 0;
    }
#ifdef RLHIP_QB_PROF
    {
        unsigned long long pf[3];
        rlhip_stream_sync(c);
        hipMemcpy(pf, g.prof, sizeof(pf), hipMemcpyDeviceToHost);
        fprintf(stderr, "[qr_blk prof %ld x %ld] us per chunk: last application %.2f  factorization %.2f  publication %.2f\n", (long)m, (long)n,
                pf[0] / 100.0 / G, pf[1] / 100.0 / G, pf[2] / 100.0 / G);
    }
#endif
    rlhip_ws_release(c, mark);
    c->path_count[8]++;
    return 1;
}
template int geqrf_blk<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*);
template int geqrf_blk<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*);

// Sign-modified LU without pivoting of the n x n matrix A (what lapack::orhr_col runs on the top block of Q): L (unit lower) and U in
// place, D(i) = -sign of the i-th pivot before its modification.  Returns 1 when done here, 0 when the problem does not fit the kernel.
template <typename T>
int lunp_blk(rlhip_ctx* c, int64_t n, T* A, int64_t lda, T* D) {
    constexpr int NT = 512, RPT = 4;
    if (n < 1 || n > (int64_t)NT * RPT) return 0;
    const int64_t G = (n + 7) / 8;
    if (G > c->num_cu) return 0;
    size_t mark = rlhip_ws_mark(c);
    LbArgs<T> g;
    g.n = n; g.A = A; g.lda = lda; g.D = D;
    g.flag = ws_alloc<unsigned>(c, (size_t)G + 4);
    if (!g.flag) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    {
        const hipError_t me = hipMemsetAsync(g.flag, 0, (size_t)G * sizeof(unsigned), c->stream);
        if (me != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(me); }
    }
    void* kargs[] = {(void*)&g};
    if (hipLaunchCooperativeKernel((const void*)lunp_blk_kernel<T, NT, RPT>, dim3((unsigned)G), dim3(NT), kargs, 0, c->stream) != hipSuccess) {
        (void)hipGetLastError();
        rlhip_ws_release(c, mark);
        return 0;
    }
    rlhip_ws_release(c, mark);
    c->path_count[9]++;
    return 1;
}
template int lunp_blk<double>(rlhip_ctx*, int64_t, double*, int64_t, double*);
template int lunp_blk<float>(rlhip_ctx*, int64_t, float*, int64_t, float*);

}  // namespace rlhip
