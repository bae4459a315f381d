// Sparse sketching operator (SASO) for CQRRPT / CQRRT and the column-permutation kernels.
//
// Replaces RandBLAS::SparseDist / SparseSkOp / sketch_general at RandLAPACK/drivers/rl_cqrrpt.hh:214-222
// (A_hat = S * A, S is d x m with `nnz` nonzeros of value +-1 in every column) and util::col_swap
// (misc/rl_util.hh:151-198 == LAPACK lapmt forward) at rl_cqrrpt.hh:288, rl_bqrrp.hh:369.
//
// RandBLAS is absent from the reference tree, so the operator's random stream is this library's own ("parity
// unpinned"; restated independently in oracle/__init__.py::saso_dense).  Either way the sketch is
// applied as a deterministic GATHER over inverse lists (which input rows feed sketch row r): every output element is
// summed by one thread in a fixed order, bitwise reproducible, no floating-point atomics.  Input rows are cut into
// blocks of d.  Two structures:
//   mode 1, INDEPENDENT COLUMNS (default; the distribution SURVEY 8 a8 describes for RandBLAS::SparseSkOp): column j
//     draws nnz DISTINCT rows by a Fisher-Yates walk over {0..d-1} and nnz iid signs.  Step i of column j uses Philox
//     block ctr + j * NB + i / 2 (NB = ceil(nnz / 2)), words 2 (i % 2) (position: ell = i + ((w * (d - i)) >> 32)) and
//     2 (i % 2) + 1 (sign = low bit); next state = ctr + m * NB.  The inverse lists are built once on the device
//     (count -> per-block scan -> scatter -> per-list sort by source row: a counting sort with a deterministic result).
//   mode 0, BLOCK AFFINE (faster: closed-form inverse map, exactly nnz sources per sketch row and block): inside block
//     t, input row u feeds  r_i(u) = (a_t * u + b_{t,i}) mod d,  gcd(a_t, d) = 1, b_{t,0..nnz-1} pairwise distinct,
//     iid signs.  (a_t, b_t) come from Philox(ctr + t), the signs of input row j from Philox(ctr + T + j); next state =
//     ctr + T + m, T = ceil(m/d).  Every d x d block of S is then a sum of nnz signed permutation matrices.
//
// apply: grid = (column tiles of 8) x (groups of row blocks); a workgroup stages the d x 8 block of A in LDS
// (coalesced reads, A is streamed exactly once), each thread owns up to 8 sketch rows x 8 columns of
// accumulators in registers and gathers its nnz source rows per block from LDS with ds_read_b128; partial
// sketches of the row-block groups are summed in fixed order.  HBM-bound: 8*m*n bytes.
#include <cstdlib>
#include <vector>
#include "rlhip_internal.h"

extern "C" int rlhip_malloc(rlhip_ctx* ctx, void** dev_ptr, size_t bytes);   // the context's caching pool (capi.hip)
extern "C" int rlhip_free(rlhip_ctx* ctx, void* dev_ptr);

namespace {

__device__ inline void philox4x32_10_dev(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ inline void ctr_add_dev(const uint32_t base[4], uint64_t inc, uint32_t out[4]) {
    uint64_t lo = ((uint64_t)base[1] << 32) | base[0], hi = ((uint64_t)base[3] << 32) | base[2];
    uint64_t nlo = lo + inc;
    if (nlo < lo) hi += 1;
    out[0] = (uint32_t)nlo; out[1] = (uint32_t)(nlo >> 32); out[2] = (uint32_t)hi; out[3] = (uint32_t)(hi >> 32);
}
__device__ inline int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }
// modular inverse of a mod d (gcd = 1), extended Euclid
__device__ inline int64_t modinv(int64_t a, int64_t d) {
    int64_t t = 0, nt = 1, r = d, nr = a % d;
    while (nr) { int64_t q = r / nr; int64_t tmp = t - q * nt; t = nt; nt = tmp; tmp = r - q * nr; r = nr; nr = tmp; }
    return t < 0 ? t + d : t;
}

struct SasoState { uint32_t ctr[4]; uint32_t key[2]; };

// one thread per row block t: parameters a_t^{-1} and b_{t,i}
__global__ void saso_params_kernel(int64_t d, int64_t T, int nnz, SasoState st, int64_t* __restrict__ ainv,
                                   int64_t* __restrict__ b, int64_t* __restrict__ afwd) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    uint32_t c[4], r[4];
    ctr_add_dev(st.ctr, (uint64_t)t, c);
    philox4x32_10_dev(c, st.key, r);
    // a: first candidate >= 1 (from r[0]) coprime with d
    int64_t a = 1 + (int64_t)(r[0] % (uint32_t)(d > 1 ? d - 1 : 1));
    while (gcd64(a, d) != 1) { a += 1; if (a >= d) a = 1; }
    ainv[t] = (d > 1) ? modinv(a, d) : 0;
    afwd[t] = (d > 1) ? a : 0;
    // b_i: LCG walk seeded by r[1..3], rejection for distinctness
    uint64_t s = ((uint64_t)r[1] << 32) | r[2];
    for (int i = 0; i < nnz; ++i) {
        for (;;) {
            s = s * 6364136223846793005ull + ((uint64_t)r[3] | 1ull);
            int64_t cand = (int64_t)((s >> 33) % (uint64_t)d);
            bool dup = false;
            for (int l = 0; l < i; ++l) dup |= (b[t * nnz + l] == cand);
            if (!dup) { b[t * nnz + i] = cand; break; }
        }
    }
}

// src[(t*nnz + i)*d + r] = u | (sign << 31); u == 0x7fffffff marks "no source row" (beyond m, last partial block)
__global__ void saso_lists_kernel(int64_t d, int64_t m, int64_t T, int nnz, SasoState st,
                                  const int64_t* __restrict__ ainv, const int64_t* __restrict__ b,
                                  int32_t* __restrict__ src) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = T * nnz * d;
    if (idx >= total) return;
    int64_t r = idx % d, ti = idx / d, i = ti % nnz, t = ti / nnz;
    int64_t u = ((r - b[t * nnz + i] + d) % d) * ainv[t] % d;     // inverse of r = a u + b
    int64_t j = t * d + u;
    if (j >= m) { src[idx] = 0x7fffffff; return; }
    uint32_t c[4], w[4];
    ctr_add_dev(st.ctr, (uint64_t)(T + j), c);
    philox4x32_10_dev(c, st.key, w);
    uint32_t bit = (w[(i >> 5) & 3] >> (i & 31)) & 1u;
    src[idx] = (int32_t)((uint32_t)u | (bit << 31));
}


// ---- mode 1 (independent columns): generation and the inverse lists ----------------------------------------------------------
// rows[j * nnz + i] = r | (sign << 31): the nnz distinct sketch rows of column j (Fisher-Yates over a virtual identity vector:
// only the touched positions are tracked); cnt[(j / d) * d + r] counts the sources of sketch row r inside row block j / d.
// (MAXNZ: capacity of the per-thread table of touched positions -- 8 for the usual few nonzeros per column keeps it in registers; with
// the general 128 the table lives in scratch memory, 1 KiB per thread: 120 us for C3's 1048576 columns of 4)
template <int MAXNZ, typename F>
__device__ __forceinline__ void saso_draw_column(int64_t d, int nnz, const SasoState& st, int64_t j, F&& emit) {
    const int64_t NB = (nnz + 1) / 2;
    int32_t mp[MAXNZ], mv[MAXNZ];        // touched positions beyond the ones already drawn, and what they hold now
    int nm = 0;
    uint32_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < nnz; ++i) {
        if ((i & 1) == 0) {
            uint32_t c[4];
            ctr_add_dev(st.ctr, (uint64_t)(j * NB + (i >> 1)), c);
            philox4x32_10_dev(c, st.key, w);
        }
        const uint32_t wa = w[2 * (i & 1)], wb = w[2 * (i & 1) + 1];
        const int32_t ell = (int32_t)(i + (int64_t)(((uint64_t)wa * (uint64_t)(d - i)) >> 32));
        int32_t a = i, b = ell;          // current contents of positions i and ell
        int ib = -1;
        for (int l = 0; l < nm; ++l) {
            if (mp[l] == i) a = mv[l];
            if (mp[l] == ell) { b = mv[l]; ib = l; }
        }
        if (ell != i) {                  // swap: position i takes b (final), position ell takes a
            if (ib >= 0) mv[ib] = a; else { mp[nm] = ell; mv[nm] = a; ++nm; }
        } else {
            b = a;
        }
        emit(i, (uint32_t)b | ((wb & 1u) << 31));
    }
}

template <int MAXNZ>
__global__ void saso_ind_gen_kernel(int64_t d, int64_t m, int nnz, SasoState st, int32_t* __restrict__ rows, int32_t* __restrict__ cnt) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t t = j / d;
    saso_draw_column<MAXNZ>(d, nnz, st, j, [&](int i, uint32_t e) {
        rows[j * nnz + i] = (int32_t)e;
        atomicAdd(&cnt[t * d + (int64_t)(e & 0x7fffffffu)], 1);
    });
}

// The four steps below (draw + count, scan, scatter, sort) for ONE row block in ONE workgroup, everything in LDS: the counters, the
// cursors and the lists of a block (d rows, <= d * nnz entries) never leave the CU, the atomics are LDS atomics, and rows / ptr / ent /
// ent16 go out once, coalesced.  C3 (1048576 columns of 4, d = 1280): 0.36 ms of global atomics (4 M counted, 4 M returned) in four
// launches -> one launch.  The lists are sorted by source row as before, so the result does not depend on the arrival order.
// LDS: cnt, ptr, cur [d] + rows, ent [d * nnz] words.
constexpr int SIB_THREADS = 256;
__global__ __launch_bounds__(SIB_THREADS) void saso_ind_block_kernel(int64_t d, int64_t m, int nnz, SasoState st, int64_t Tb, int32_t* __restrict__ rows,
                                                                      int32_t* __restrict__ ptr, int32_t* __restrict__ ent, uint16_t* __restrict__ ent16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sib_smem[];
    __shared__ int32_t s_part[SIB_THREADS];
    int32_t* s_cnt = reinterpret_cast<int32_t*>(sib_smem);
    int32_t* s_ptr = s_cnt + d;
    int32_t* s_cur = s_ptr + d;
    int32_t* s_rows = s_cur + d;
    int32_t* s_ent = s_rows + d * nnz;
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t j0 = t * d;
    const int ncol = (int)((m - j0 < d) ? (m - j0) : d);
    const int nd = (int)d;
    for (int r = tid; r < nd; r += SIB_THREADS) s_cnt[r] = 0;
    __syncthreads();
    for (int u = tid; u < ncol; u += SIB_THREADS)
        saso_draw_column<8>(d, nnz, st, j0 + u, [&](int i, uint32_t e) {
            s_rows[u * nnz + i] = (int32_t)e;
            atomicAdd(&s_cnt[e & 0x7fffffffu], 1);
        });
    __syncthreads();
    // exclusive scan of the d counters: a contiguous chunk per thread, the chunk sums scanned across the workgroup
    const int per = (nd + SIB_THREADS - 1) / SIB_THREADS;
    const int r0 = tid * per, r1 = (r0 + per < nd) ? (r0 + per) : nd;
    int32_t sum = 0;
    for (int r = r0; r < r1; ++r) sum += s_cnt[r];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < SIB_THREADS; off <<= 1) {
        const int32_t add = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    int32_t run = s_part[tid] - sum;
    for (int r = r0; r < r1; ++r) { s_ptr[r] = run; s_cur[r] = run; run += s_cnt[r]; }
    __syncthreads();
    for (int u = tid; u < ncol; u += SIB_THREADS)
        for (int i = 0; i < nnz; ++i) {
            const uint32_t e = (uint32_t)s_rows[u * nnz + i];
            const int32_t pos = atomicAdd(&s_cur[e & 0x7fffffffu], 1);
            s_ent[pos] = (int32_t)((uint32_t)u | (e & 0x80000000u));
        }
    __syncthreads();
    for (int r = tid; r < nd; r += SIB_THREADS) {
        const int32_t p0 = s_ptr[r], p1 = p0 + s_cnt[r];
        for (int32_t p = p0 + 1; p < p1; ++p) {
            const int32_t e = s_ent[p];
            int32_t q = p - 1;
            while (q >= p0 && (s_ent[q] & 0x7fffffff) > (e & 0x7fffffff)) { s_ent[q + 1] = s_ent[q]; --q; }
            s_ent[q + 1] = e;
        }
    }
    __syncthreads();
    const int64_t base = j0 * nnz;
    const int tot = ncol * nnz;
    for (int x = tid; x < tot; x += SIB_THREADS) {
        rows[base + x] = s_rows[x];
        const uint32_t e = (uint32_t)s_ent[x];
        ent[base + x] = (int32_t)e;
        if (ent16) ent16[base + x] = (uint16_t)((e & 0x7fffu) | ((e >> 16) & 0x8000u));
    }
    for (int r = tid; r < nd; r += SIB_THREADS) ptr[j0 + r] = (int32_t)base + s_ptr[r];
    if (t == Tb - 1 && tid == 0) ptr[Tb * d] = (int32_t)base + tot;
}

// ptr[t * d + r] = first entry of (block t, sketch row r) in ent[]; block t's entries start at t * d * nnz (every column of a
// block contributes nnz entries to that block), so the scan is local to a block: one workgroup per block.
__global__ __launch_bounds__(256) void saso_ind_scan_kernel(int64_t d, int nnz, const int32_t* __restrict__ cnt, int32_t* __restrict__ ptr,
                                                            int64_t Tb) {
    __shared__ int32_t s_part[256];
    __shared__ int32_t s_run;
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) s_run = (int32_t)(t * d * nnz);
    __syncthreads();
    for (int64_t r0 = 0; r0 < d; r0 += 256) {
        const int64_t r = r0 + tid;
        const int32_t v = (r < d) ? cnt[t * d + r] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {           // Hillis-Steele inclusive scan
            const int32_t add = (tid >= off) ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const int32_t base = s_run;
        if (r < d) ptr[t * d + r] = base + s_part[tid] - v;
        __syncthreads();
        if (tid == 255) s_run = base + s_part[255];
        __syncthreads();
    }
    if (t == Tb - 1 && tid == 0) ptr[Tb * d] = s_run;
}

// ent[ptr[key] + k] = u | (sign << 31) for the sources u of (block, row) = key, in arrival order (sorted by the next kernel)
__global__ void saso_ind_scatter_kernel(int64_t d, int64_t m, int nnz, const int32_t* __restrict__ rows, const int32_t* __restrict__ ptr,
                                        int32_t* __restrict__ cursor, int32_t* __restrict__ ent) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * nnz) return;
    const int64_t j = idx / nnz;
    const int64_t t = j / d, u = j - t * d;
    const uint32_t e = (uint32_t)rows[idx];
    const int64_t key = t * d + (int64_t)(e & 0x7fffffffu);
    const int32_t k = atomicAdd(&cursor[key], 1);
    ent[ptr[key] + k] = (int32_t)((uint32_t)u | (e & 0x80000000u));
}

// every list sorted by source row: the summation order of the gather no longer depends on the arrival order of the scatter
// (ent16: nullptr, or the same lists with 16-bit entries: source row | sign << 15)
__global__ void saso_ind_sort_kernel(int64_t nkeys, const int32_t* __restrict__ ptr, int32_t* __restrict__ ent, uint16_t* __restrict__ ent16) {
    const int64_t key = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= nkeys) return;
    const int32_t p0 = ptr[key], p1 = ptr[key + 1];
    for (int32_t p = p0 + 1; p < p1; ++p) {
        const int32_t e = ent[p];
        int32_t q = p - 1;
        while (q >= p0 && (ent[q] & 0x7fffffff) > (e & 0x7fffffff)) { ent[q + 1] = ent[q]; --q; }
        ent[q + 1] = e;
    }
    if (ent16)
        for (int32_t p = p0; p < p1; ++p) {
            const uint32_t e = (uint32_t)ent[p];
            ent16[p] = (uint16_t)((e & 0x7fffu) | ((e >> 16) & 0x8000u));
        }
}

template <typename T>
__global__ void saso_ind_dense_kernel(int64_t d, int64_t m, int nnz, const int32_t* __restrict__ rows, T* __restrict__ S) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * nnz) return;
    const uint32_t e = (uint32_t)rows[idx];
    S[(int64_t)(e & 0x7fffffffu) + (idx / nnz) * d] = (e & 0x80000000u) ? T(-1) : T(1);
}

// dense copy of S (d x m, column-major) for tests: S[r, j] = +-1
template <typename T>
__global__ void saso_dense_kernel(int64_t d, int64_t m, int64_t Tb, int nnz, const int32_t* __restrict__ src,
                                  T* __restrict__ S) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Tb * nnz * d) return;
    int32_t e = src[idx];
    if ((e & 0x7fffffff) == 0x7fffffff) return;
    int64_t r = idx % d, t = (idx / d) / nnz;
    int64_t j = t * d + (int64_t)(e & 0x7fffffff);
    S[r + j * d] += (e & 0x80000000) ? T(-1) : T(1);    // distinct (r, j) per entry: no race
}

// ---- S * A for a SPARSE A (one column of A = one row of the transpose's CSR).  A column holds few nonzeros, so the product is a
// scatter: entry (j, v) adds +-v to the nnz sketch rows r_i(j).  Floating-point atomics would make the sum order-dependent;
// instead every value is converted to 64-bit fixed point with a per-column power-of-two scale chosen so that the column cannot
// overflow (|q| <= 2^61 / len), accumulated with INTEGER LDS atomics (associative => bitwise reproducible), and converted back.
// Quantisation error per entry: 2^-62 * len * max|v|, below the rounding error of a floating-point sum of the same terms.
template <typename T>
__global__ __launch_bounds__(256) void saso_apply_csr_kernel(int64_t d, int64_t m, int64_t Tb, int nnz, SasoState st,
                                                             const int64_t* __restrict__ afwd, const int64_t* __restrict__ b,
                                                             const int64_t* __restrict__ rowptrT, const int64_t* __restrict__ colidxT,
                                                             const T* __restrict__ valsT, T alpha, T beta, T* __restrict__ B, int64_t ldb, int64_t row0,
                                                             const int32_t* __restrict__ rows /* mode 1: the columns' row lists; nullptr in mode 0 */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem_raw);       // [d]
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x;
    const int64_t p0 = rowptrT[c], p1 = rowptrT[c + 1], len = p1 - p0;
    for (int64_t r = tid; r < d; r += 256) acc[r] = 0ull;
    double mx = 0;
    for (int64_t p = p0 + tid; p < p1; p += 256) mx = fmax(mx, fabs((double)valsT[p]));
    for (int off = 32; off; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) s_red[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    int e = 0;
    if (mx > 0 && len > 0) {
        int ex;
        frexp(mx, &ex);                                    // mx < 2^ex
        int lg = 0;
        while (((int64_t)1 << lg) < len) ++lg;             // len <= 2^lg
        e = 61 - lg - ex;
        for (int64_t p = p0 + tid; p < p1; p += 256) {
            const int64_t j = colidxT[p] + row0;               // global row of the operand (row-sharded operators pass their offset)
            const long long q = llrint(ldexp((double)valsT[p], e));
            if (rows) {
                for (int i = 0; i < nnz; ++i) {
                    const uint32_t e = (uint32_t)rows[j * nnz + i];
                    atomicAdd(&acc[e & 0x7fffffffu], (unsigned long long)((e & 0x80000000u) ? -q : q));
                }
                continue;
            }
            const int64_t t = j / d, u = j - t * d;
            uint32_t ctr[4], w[4];
            ctr_add_dev(st.ctr, (uint64_t)(Tb + j), ctr);
            philox4x32_10_dev(ctr, st.key, w);
            const int64_t au = (afwd[t] * u) % d;
            for (int i = 0; i < nnz; ++i) {
                int64_t r = au + b[t * nnz + i];
                if (r >= d) r -= d;
                const uint32_t bit = (w[(i >> 5) & 3] >> (i & 31)) & 1u;
                atomicAdd(&acc[r], (unsigned long long)(bit ? -q : q));
            }
        }
    }
    __syncthreads();
    (void)m;
    for (int64_t r = tid; r < d; r += 256) {
        const double v = ldexp((double)(long long)acc[r], -e);
        T* dst = B + r + c * ldb;
        *dst = (beta == (T)0) ? (T)(alpha * v) : (T)(alpha * v) + beta * *dst;
    }
}

// CT = columns per workgroup: 4 gives a 40 KiB slab at d = 1280 (fp64), so several workgroups per CU overlap staging and gathering
// (C3: 8 -> 6.3 ms, 4 -> 4.6 ms, 2 -> 6.3 ms); taller sketches take 2 or 1 columns so that the d x CT slab still fits the CU's LDS

// partial[g][r + c*d] = sum over row blocks t in group g of sum_i sign * A[t*d + u_i(r), c]
template <typename T, int CT, int MODE, int NR>
__global__ __launch_bounds__(256, (NR == 5) ? 4 : 1) void saso_apply_kernel(int64_t d, int64_t n, int64_t m, int64_t Tb, int nnz,
                                                         const int32_t* __restrict__ src, const T* __restrict__ A,
                                                         int64_t lda, int64_t t_per_group, int64_t r_base,
                                                         T* __restrict__ partial, int64_t row0, int64_t mloc, int64_t tb0,
                                                         int64_t tb1, const int32_t* __restrict__ ptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // LDS image of the slab: 16-byte elements (CPE columns of one row), NH planes of d elements: plane h holds columns h CPE .. of
    // every row.  A gather lane then reads ONE 16-byte element per plane, and rows u, u' collide only if u = u' mod 16 (the 64 banks
    // hold 16 such elements) -- with the [d][CT] row-major image (32-byte rows for fp64, CT = 4) every row started on one of 8 bank
    // groups: 2-way conflicts for the affine operator's strided rows, ~4-way for independent columns, and 8-way on the staging stores.
    constexpr int EB = (CT * (int)sizeof(T) >= 16) ? 16 : CT * (int)sizeof(T);   // element bytes
    constexpr int CPE = EB / (int)sizeof(T);                                     // columns per element
    constexpr int NH = CT / CPE;                                                 // planes
    struct alignas(EB) Elem { T v[CPE]; };
    Elem* sE = reinterpret_cast<Elem*>(smem_raw);               // [NH][d]
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CT;
    const int64_t g = blockIdx.y;
    // A holds global rows [row0, row0 + mloc) only (row-block sharding); row blocks tb0 .. tb1-1 intersect that window
    const int64_t t0 = tb0 + g * t_per_group, t1 = (t0 + t_per_group < tb1) ? (t0 + t_per_group) : tb1;
    (void)Tb;
    T acc[NR][CT];
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[q][c] = T(0);
    for (int64_t t = t0; t < t1; ++t) {
        __syncthreads();
        // stage A[t*d : t*d+d, c0:c0+CT] (zero padded) -- threads run along rows: coalesced global loads; a thread holds all CT values
        // of its rows and stores them as whole 16-byte elements (see the layout note above): conflict-free
        for (int ub = 0; ub < (int)d; ub += 256 * NR) {           // (the slab holds all d source rows; 2048 per trip)
            T stg[NR][CT];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int u = ub + tid + 256 * i;
                const int64_t j = t * d + u - row0;      // local row
                const bool row_ok = u < (int)d && j >= 0 && j < mloc && t * d + u < m;
#pragma unroll
                for (int c = 0; c < CT; ++c) stg[i][c] = (row_ok && c0 + c < n) ? A[(c0 + c) * lda + j] : T(0);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int u = ub + tid + 256 * i;
                if (u < (int)d) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        Elem e;
#pragma unroll
                        for (int c2 = 0; c2 < CPE; ++c2) e.v[c2] = stg[i][h * CPE + c2];
                        sE[(int64_t)h * d + u] = e;
                    }
                }
            }
        }
        __syncthreads();
        if (MODE == 1) {
            // variable-length lists (block t, row r) = src[ptr[t d + r] .. ptr[t d + r + 1]), sorted by source row.  All list bounds of
            // the thread's NR rows are requested first, then the first eight entries of every list as two 16-byte loads (a list
            // is contiguous; reading past its end is harmless -- the array is padded -- and masked by the length), so that no load
            // waits for another; lists longer than eight entries (2 % of them at nnz = 4) finish in a scalar tail loop.
            int32_t p0[NR], len[NR];
            int4 ea[NR], eb[NR];
#pragma unroll
            for (int q = 0; q < NR; ++q) {                       // (clamped, not branched: written as `if (r < d) load`, hipcc gives every bound its own
                const int64_t r = r_base + tid + 256 * q;        //  branch with an s_waitcnt vmcnt(0) behind it -- NR dependent round trips per block)
                const int64_t rc = (r < d) ? r : d - 1;
                const int32_t a0 = ptr[t * d + rc], a1 = ptr[t * d + rc + 1];
                p0[q] = a0; len[q] = a1 - a0;
            }
            __builtin_amdgcn_sched_barrier(0);                   // all NR bounds in flight before the first one is used
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const bool ok = (r_base + tid + 256 * q) < d;
                p0[q] = ok ? p0[q] : 0; len[q] = ok ? len[q] : 0;
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                int tmp[8];
                __builtin_memcpy(tmp, src + p0[q], 32);          // two unaligned 16-byte loads
                ea[q] = int4{tmp[0], tmp[1], tmp[2], tmp[3]};
                eb[q] = int4{tmp[4], tmp[5], tmp[6], tmp[7]};
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int ev[8] = {ea[q].x, ea[q].y, ea[q].z, ea[q].w, eb[q].x, eb[q].y, eb[q].z, eb[q].w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k < len[q]) {
                        const int32_t e = ev[k];
                        const T sg = (e & 0x80000000) ? T(-1) : T(1);
                        const int64_t ur = (int64_t)(e & 0x7fffffff);
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            const Elem el = sE[(int64_t)h * d + ur];
#pragma unroll
                            for (int c2 = 0; c2 < CPE; ++c2) acc[q][h * CPE + c2] += sg * el.v[c2];
                        }
                    }
                }
                for (int32_t k = 8; k < len[q]; ++k) {
                    const int32_t e = src[p0[q] + k];
                    const T sg = (e & 0x80000000) ? T(-1) : T(1);
                    const int64_t ur = (int64_t)(e & 0x7fffffff);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const Elem el = sE[(int64_t)h * d + ur];
#pragma unroll
                        for (int c2 = 0; c2 < CPE; ++c2) acc[q][h * CPE + c2] += sg * el.v[c2];
                    }
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int64_t r = r_base + tid + 256 * q;
            if (r < d) {
                for (int i = 0; i < nnz; ++i) {
                    const int32_t e = src[(t * nnz + i) * d + r];
                    if ((e & 0x7fffffff) != 0x7fffffff) {
                        const T sg = (e & 0x80000000) ? T(-1) : T(1);
                        const int64_t ur = (int64_t)(e & 0x7fffffff);
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            const Elem el = sE[(int64_t)h * d + ur];
#pragma unroll
                            for (int c2 = 0; c2 < CPE; ++c2) acc[q][h * CPE + c2] += sg * el.v[c2];
                        }
                    }
                }
            }
        }
        }
    }
    T* out = partial + g * d * n;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int64_t r = r_base + tid + 256 * q;
        if (r < d)
#pragma unroll
            for (int c = 0; c < CT; ++c)
                if (c0 + c < n) out[r + (c0 + c) * d] = acc[q][c];
    }
}


// ---- the same product with the slab brought in by LDS-DMA and the index one step ahead (independent-column operators, the default) ----
// What the register-staged kernel above loses (C3, scripts/saso_time.py): staging alone 1.63 ms, staging + gather 2.91 ms --
// and 2.89 ms with every gather address replaced by a conflict-free one.  The gather's cost is not LDS bank conflicts but the index:
// bounds -> entries are two DEPENDENT round trips (L2 / fabric latency each) that start only after the slab's rendezvous, with four
// workgroups per CU to hide them; hoisting them in front of the slab's loads needs 40 more registers than a 128-register wave has.
// Here the slab needs no registers at all: every wave requests its 1 KiB pieces of the NEXT block with global_load_lds into the other of
// two LDS buffers right behind the rendezvous, the lists of the next block (16-bit entries, sixteen per row = two 16-byte loads + one
// dword for an odd start) and the bounds of the block after it are requested behind those, and everything lands under the gather of the
// current block.  ONE wait per block -- s_waitcnt vmcnt(0) at the top, when everything outstanding is one block old -- so no counted
// wait depends on where hipcc puts a load (tri.hip's lesson).  LDS image of a buffer: CT planes of d elements, column after column,
// i.e. byte 16 p of the buffer = 16-byte piece p of the slab: the DMA destination is lane-linear as the instruction demands.
// Lists longer than sixteen entries (1e-6 of them at 4 nonzeros per column) finish in a tail loop.
template <typename T, int NR, int NJ, int DD>
__global__ __launch_bounds__(256, 2) void saso_apply_dma_kernel(int64_t d_rt, int64_t n, const uint16_t* __restrict__ src16, const int32_t* __restrict__ ptr,
                                                               const T* A, int64_t lda, int64_t t_per_group, T* __restrict__ partial, int64_t row0,
                                                               int64_t fb0, int64_t fb1) {
    constexpr int CT = 4;
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void glb_void_t;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int64_t d = DD ? (int64_t)DD : d_rt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t c0 = (int64_t)blockIdx.x * CT, g = blockIdx.y;
    const int64_t t0 = fb0 + g * t_per_group, t1 = (t0 + t_per_group < fb1) ? (t0 + t_per_group) : fb1;
    const int slab_bytes = (int)(CT * d * (int64_t)sizeof(T));
    const int buf_stride = (slab_bytes + 1023) & ~1023;
    const int npieces = slab_bytes / 16, nchunks = (npieces + 63) / 64, ppc = (int)(d * (int64_t)sizeof(T) / 16);
    // per-lane source byte offsets of this wave's pieces, relative to the first element of the block's first column
    unsigned goff[NJ];
    int kdst[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int k = wid + 4 * j;
        if (k >= nchunks) k %= nchunks;                          // a duplicate: the same bytes to the same place
        int pc = k * 64 + lane;
        if (pc > npieces - 1) pc = npieces - 1;                   // lanes past the slab's end land in the buffer's padding
        const int col = pc / ppc, rp = pc - col * ppc;
        goff[j] = (unsigned)(((int64_t)col * lda) * (int64_t)sizeof(T) + (int64_t)rp * 16);
        kdst[j] = k * 1024;
    }
    auto issue_slab = [&](int64_t t, int buf) {
        const char* base = reinterpret_cast<const char*>(A + c0 * lda + (t * d - row0));
        unsigned char* dst = smem_raw + buf * buf_stride;
#pragma unroll
        for (int j = 0; j < NJ; ++j) __builtin_amdgcn_global_load_lds((glb_void_t*)(base + goff[j]), (lds_void_t*)(dst + kdst[j]), 16, 0, 0);
    };
    // index of one block: bounds first (p0, len), then 18 half-words from the even position at or below the list start
    int rr[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) { const int r = tid + 256 * q; rr[q] = (r < (int)d) ? r : (int)d - 1; }
    // (raw list starts and ends: nothing is computed from them here -- an instruction that uses a loaded value makes hipcc wait for the
    // load on the spot, and with it for every older request, the slab's pieces included)
    auto load_bounds = [&](int64_t t, int32_t (&p0)[NR], int32_t (&p1)[NR]) {
#pragma unroll
        for (int q = 0; q < NR; ++q) { p0[q] = ptr[t * d + rr[q]]; p1[q] = ptr[t * d + rr[q] + 1]; }
    };
    auto load_entries = [&](const int32_t (&p0)[NR], int4 (&ea)[NR], int4 (&eb)[NR], int (&ec)[NR]) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int* w = reinterpret_cast<const int*>(src16 + (p0[q] & ~1));
            int tmp[9];
            __builtin_memcpy(tmp, w, 36);
            ea[q] = int4{tmp[0], tmp[1], tmp[2], tmp[3]};
            eb[q] = int4{tmp[4], tmp[5], tmp[6], tmp[7]};
            ec[q] = tmp[8];
        }
    };
    T acc[NR][CT];
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[q][c] = T(0);
    // index registers: (M, E) = bounds + entries of a block, two sets that alternate (a: even steps of a trip, b: odd); X = the bounds in flight
    int32_t p0a[NR], lena[NR], p0b[NR], lenb[NR], p0x[NR], p1x[NR];
    int4 eaa[NR], eba[NR], eab[NR], ebb[NR];
    int eca[NR], ecb[NR];
    auto take_bounds = [&](int32_t (&p0)[NR], int32_t (&len)[NR]) {      // X (landed) -> a set; rows past d get empty lists
#pragma unroll
        for (int q = 0; q < NR; ++q) { p0[q] = p0x[q]; len[q] = (tid + 256 * q < (int)d) ? p1x[q] - p0x[q] : 0; }
    };
    // the gather of one block: entries in batches of GB, every batch branch-free (an entry past the end of a lane's list reads element
    // 0 with weight 0) so that its LDS reads go out back to back; a batch is skipped when NO lane of the wave has that many entries (a
    // scalar branch).  The (row, batch) slots of a thread form ONE static sequence and the reads of a slot are issued in front of the
    // additions of the slot before it: with two waves per SIMD there is nobody else to cover an LDS round trip.
    constexpr int GB = 2, NB = 16 / GB;                             // entries per batch, batches per row
    auto gather = [&](const T* sA, const int32_t (&p0)[NR], const int32_t (&len)[NR], const int4 (&ea)[NR], const int4 (&eb)[NR], const int (&ec)[NR]) {
        T sg[2][GB], v[2][GB][CT];
        bool act[2] = {false, false};
#pragma unroll
        for (int sl = 0; sl <= NR * NB; ++sl) {
            const int q = sl / NB, bt = sl % NB, cur = sl & 1, prv = cur ^ 1;
            if (sl < NR * NB) {
                act[cur] = __builtin_amdgcn_ballot_w64(len[q] > GB * bt) != 0;
                if (act[cur]) {
                    const unsigned w9[9] = {(unsigned)ea[q].x, (unsigned)ea[q].y, (unsigned)ea[q].z, (unsigned)ea[q].w, (unsigned)eb[q].x,
                                            (unsigned)eb[q].y, (unsigned)eb[q].z, (unsigned)eb[q].w, (unsigned)ec[q]};
                    const unsigned sh = (unsigned)(p0[q] & 1) * 16u;
#pragma unroll
                    for (int h4 = 0; h4 < GB; ++h4) {
                        const int k = GB * bt + h4, i = k >> 1;
                        const unsigned pair = __builtin_amdgcn_alignbit(w9[i + 1], w9[i], sh);      // entries 2 i, 2 i + 1 of the list
                        const unsigned e = (k & 1) ? (pair >> 16) : (pair & 0xffffu);
                        const bool on = k < len[q];
                        sg[cur][h4] = on ? ((e & 0x8000u) ? T(-1) : T(1)) : T(0);
                        const T* el = sA + (on ? (e & 0x7fffu) : 0u);
#pragma unroll
                        for (int c = 0; c < CT; ++c) v[cur][h4][c] = el[(int64_t)c * d];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sl > 0 && act[prv]) {
                const int qp = (sl - 1) / NB;
#pragma unroll
                for (int h4 = 0; h4 < GB; ++h4)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[qp][c] += sg[prv][h4] * v[prv][h4][c];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q)
            for (int32_t k = 16; k < len[q]; ++k) {
                const unsigned e = src16[p0[q] + k];
                const T sg1 = (e & 0x8000u) ? T(-1) : T(1);
                const T* el1 = sA + (e & 0x7fffu);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[q][c] += sg1 * el1[(int64_t)c * d];
            }
    };
    // Every prefetch below is UNCONDITIONAL (past the last block the index of the last block is fetched again, and its slab once more into
    // the idle buffer): a load under `if (t + 1 < t1)` merges with the old register value at the join, hipcc puts the copy there, and the copy
    // waits for the load -- and for the slab's pieces in front of it.
    const int64_t tl = t1 - 1;
    auto cl = [&](int64_t t) { return t < tl ? t : tl; };
    if (t0 < t1) {
        issue_slab(t0, 0);
        load_bounds(t0, p0x, p1x);
        take_bounds(p0a, lena);                                     // (waits for the bounds: prologue only)
        load_entries(p0a, eaa, eba, eca);
        load_bounds(cl(t0 + 1), p0x, p1x);
    }
    // one step = one block; two steps per trip so that the index sets alternate without copies of registers still in flight
    for (int64_t t = t0; t < t1; t += 2) {
        // slab t, entries t, bounds t + 1: all requested one block ago.  (The builtin, not inline asm: hipcc's own wait insertion then KNOWS the
        // counter is drained and adds nothing in front of the first use of those registers -- where a conservative wait would also cover
        // the requests issued just below.)
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        asm volatile("" ::: "memory");
        __syncthreads();                                            // slab t visible to all; everybody has left slab t - 1
        issue_slab(cl(t + 1), 1);
        take_bounds(p0b, lenb);
        load_entries(p0b, eab, ebb, ecb);
        load_bounds(cl(t + 2), p0x, p1x);
        gather(reinterpret_cast<const T*>(smem_raw), p0a, lena, eaa, eba, eca);
        if (t + 1 >= t1) break;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        __syncthreads();
        issue_slab(cl(t + 2), 0);
        take_bounds(p0a, lena);
        load_entries(p0a, eaa, eba, eca);
        load_bounds(cl(t + 3), p0x, p1x);
        gather(reinterpret_cast<const T*>(smem_raw + buf_stride), p0b, lenb, eab, ebb, ecb);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // nothing of this workgroup is in flight when it ends
    T* out = partial + g * d * n;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int64_t r = tid + 256 * q;
        if (r < d)
#pragma unroll
            for (int c = 0; c < CT; ++c) out[r + (c0 + c) * d] = acc[q][c];
    }
}

template <typename T>
__global__ void saso_reduce_kernel(int64_t total, int G, const T* __restrict__ partial, T alpha, T beta,
                                   T* __restrict__ out, int64_t d, int64_t ldo) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    T s = 0;
    for (int g = 0; g < G; ++g) s += partial[(int64_t)g * total + idx];
    int64_t r = idx % d, c = idx / d;
    T v = alpha * s;
    if (beta != T(0)) v += beta * out[r + c * ldo];
    out[r + c * ldo] = v;
}

// ---------------------------------------------------------------------------------------------------
// column permutation (forward lapmt): on exit column i holds former column idx[i]-1.
// moves[] = cycle-ordered list built by one thread; each thread of the apply kernel owns one row and walks
// the list, so the whole matrix is read once and written once, in place, with lanes along rows.
__global__ void perm_moves_kernel(int64_t n, const int64_t* __restrict__ idx, int64_t* __restrict__ moves,
                                  int64_t* __restrict__ nmoves, unsigned char* __restrict__ seen) {
    if (threadIdx.x || blockIdx.x) return;
    for (int64_t i = 0; i < n; ++i) seen[i] = 0;
    int64_t w = 0;
    bool bad = false;
    for (int64_t i = 0; i < n && !bad; ++i) {
        if (seen[i]) continue;
        seen[i] = 1;
        int64_t s = idx[i] - 1;
        if (s < 0 || s >= n) { bad = true; break; }
        if (s == i) continue;                 // fixed point
        // cycle: i <- s <- idx[s]-1 <- ... ; encode as (start marker, dst, src ...): -(i+1) opens a cycle
        moves[w++] = -(i + 1);
        while (s != i) {
            if (s < 0 || s >= n || seen[s]) { bad = true; break; }     // out of range or a repeated index: idx is not a permutation
            moves[w++] = s;                   // the column walked last receives column s
            seen[s] = 1;
            s = idx[s] - 1;
        }
    }
    // a vector that is not a permutation of 1..n leaves the matrix untouched (the host path for n >= 4096 returns -7 for it; this
    // device path is asynchronous, so the refusal shows as a no-op instead of an endless walk)
    *nmoves = bad ? 0 : w;
}

template <typename T>
__global__ __launch_bounds__(256) void perm_apply_kernel(int64_t m, T* __restrict__ A, int64_t lda,
                                                         const int64_t* __restrict__ moves,
                                                         const int64_t* __restrict__ nmoves, const int64_t* __restrict__ chunk) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    // chunk != nullptr: blockIdx.y takes the moves [chunk[y], chunk[y + 1]) -- whole cycles, cut by the host: cycles touch disjoint columns, so
    // the chunks run side by side (a 2048-row sketch is 8 workgroups of rows: with one walk over ~4000 moves each the launch was a latency chain)
    if (chunk) moves += chunk[blockIdx.y];
    const int64_t nm = chunk ? chunk[blockIdx.y + 1] - chunk[blockIdx.y] : *nmoves;
    T* row = A + r;
    T saved = 0;
    int64_t cur = 0;
    bool open = false;
    // Sixteen moves at a time: their SOURCE elements are loaded first, then stored in order.  A column of a cycle is read at its own step
    // and overwritten at the NEXT one (the start column: read when the cycle opens, overwritten by its first move), and cycles are disjoint,
    // so a load may always run ahead of the stores in front of it.  One load -> one store at a time (each behind the previous store: the
    // compiler cannot prove the columns distinct) the walk was latency-bound: 1.39 ms for ~4100 moved columns of a 65536-row matrix, 1.5 TB/s.
    constexpr int W = 16;
    for (int64_t q0 = 0; q0 < nm; q0 += W) {
        int64_t mv[W];
        T val[W];
#pragma unroll
        for (int u = 0; u < W; ++u) mv[u] = (q0 + u < nm) ? moves[q0 + u] : (int64_t)0;      // (padding: column 0, loaded and ignored)
#pragma unroll
        for (int u = 0; u < W; ++u) val[u] = row[(mv[u] < 0 ? -mv[u] - 1 : mv[u]) * lda];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (q0 + u >= nm) break;
            if (mv[u] < 0) {                   // new cycle: close the previous one first
                if (open) row[cur * lda] = saved;
                saved = val[u];
                cur = -mv[u] - 1;
                open = true;
            } else {
                row[cur * lda] = val[u];
                cur = mv[u];
            }
        }
    }
    if (open) row[cur * lda] = saved;
}

// integer vector overload (rl_util.hh:174-198): permutes the first k entries by a permutation of 1..k
__global__ void vec_gather_i64_kernel(int64_t k, const int64_t* __restrict__ in, const int64_t* __restrict__ idx,
                                      int64_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[i] = in[idx[i] - 1];
}

}  // namespace

namespace rlhip {

// opaque device-side operator
struct SasoOp {
    int64_t d, m, T;
    int nnz;
    int mode;           // 1 independent columns (default), 0 block affine
    int32_t* src;       // mode 0: T * nnz * d inverse table; mode 1: m * nnz list entries (ent)
    int64_t* ainv;      // T                                   (mode 0)
    int64_t* b;         // T * nnz                             (mode 0)
    int64_t* afwd;      // T   (the forward multiplier a_t; the sparse-operand path scatters)   (mode 0)
    int32_t* rows;      // m * nnz: the columns' own row lists (mode 1; forward map for the sparse-operand path and the dense copy)
    int32_t* ptr;       // T * d + 1 list starts               (mode 1)
    uint16_t* src16;    // mode 1, d <= 32768: the entries of src as 16 bits (source row | sign << 15), same positions (the DMA apply kernel)
    SasoState st;
};

static int saso_default_mode(const rlhip_ctx* c) { return c->opt[RLHIP_OPT_SASO_MODE] == 0 ? 0 : 1; }   // 0: block affine (RLHIP_OPT_SASO_MODE)

// the operator's index arrays come from the context's caching pool (rlhip_malloc): no hipMalloc / hipFree in steady state
#define RLHIP_SASO_ALLOC(field, bytes)                                                    \
    do {                                                                                  \
        void* _p = nullptr;                                                               \
        const int _rc = rlhip_malloc(c, &_p, (bytes));                                    \
        if (_rc) { saso_destroy(c, op); return _rc; }                                     \
        field = reinterpret_cast<decltype(field)>(_p);                                    \
    } while (0)
int saso_destroy(rlhip_ctx* c, SasoOp* op);

int saso_build(rlhip_ctx* c, int64_t d, int64_t m, int nnz, int mode, const uint32_t ctr[4], const uint32_t key[2],
               uint32_t next_ctr[4], SasoOp** out) {
    if (d <= 0 || m < 0 || nnz <= 0 || nnz > d || nnz > 128 || d >= ((int64_t)1 << 31)) return -2;
    if (mode < 0) mode = saso_default_mode(c);
    if (mode > 1) return -5;
    SasoOp* op = new SasoOp();
    op->d = d; op->m = m; op->nnz = nnz; op->T = (m + d - 1) / d; op->mode = mode;
    op->src = nullptr; op->ainv = nullptr; op->b = nullptr; op->afwd = nullptr; op->rows = nullptr; op->ptr = nullptr; op->src16 = nullptr;
    const int64_t T = op->T > 0 ? op->T : 1;
    SasoState st;
    for (int i = 0; i < 4; ++i) st.ctr[i] = ctr[i];
    st.key[0] = key[0]; st.key[1] = key[1];
    op->st = st;
    uint64_t inc;
    if (mode == 0) {
        RLHIP_SASO_ALLOC(op->src, sizeof(int32_t) * (size_t)(T * nnz * d));
        RLHIP_SASO_ALLOC(op->ainv, sizeof(int64_t) * (size_t)T);
        RLHIP_SASO_ALLOC(op->b, sizeof(int64_t) * (size_t)(T * nnz));
        RLHIP_SASO_ALLOC(op->afwd, sizeof(int64_t) * (size_t)T);
        if (op->T > 0) {
            hipLaunchKernelGGL(saso_params_kernel, dim3((unsigned)((op->T + 63) / 64)), dim3(64), 0, c->stream, d, op->T, nnz,
                               st, op->ainv, op->b, op->afwd);
            int64_t total = op->T * nnz * d;
            hipLaunchKernelGGL(saso_lists_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d, m, op->T,
                               nnz, st, op->ainv, op->b, op->src);
            RLHIP_LAUNCH_CHECK();
        }
        inc = (uint64_t)op->T + (uint64_t)m;
    } else {
        if (m * nnz >= ((int64_t)1 << 31)) { delete op; return -3; }      // 32-bit list positions
        const size_t nent = (size_t)(m * nnz > 0 ? m * nnz : 1), nkeys = (size_t)(T * d);
        RLHIP_SASO_ALLOC(op->rows, sizeof(int32_t) * nent);
        RLHIP_SASO_ALLOC(op->src, sizeof(int32_t) * (nent + 8));        // + 8: the apply kernel reads eight entries from any list start
        RLHIP_CHECK(hipMemsetAsync(op->src + nent, 0, sizeof(int32_t) * 8, c->stream));
        RLHIP_SASO_ALLOC(op->ptr, sizeof(int32_t) * (nkeys + 1));
        if (d <= 32768) {                                               // + 20: the DMA apply kernel reads 18 entries from the even position at or below any list start
            RLHIP_SASO_ALLOC(op->src16, sizeof(uint16_t) * (nent + 20));
            RLHIP_CHECK(hipMemsetAsync(op->src16 + nent, 0, sizeof(uint16_t) * 20, c->stream));
        }
        const size_t sib_lds = sizeof(int32_t) * (size_t)d * (size_t)(3 + 2 * nnz);
        if (op->T > 0 && nnz <= 8 && sib_lds <= 144 * 1024) {           // one workgroup per row block, lists built in LDS
            RLHIP_FUNC_LDS(c, saso_ind_block_kernel, 144 * 1024);
            hipLaunchKernelGGL(saso_ind_block_kernel, dim3((unsigned)op->T), dim3(SIB_THREADS), sib_lds, c->stream, d, m, nnz, st, op->T, op->rows, op->ptr,
                               op->src, op->src16);
            RLHIP_LAUNCH_CHECK();
        } else if (op->T > 0) {
            size_t mark = rlhip_ws_mark(c);
            int32_t* cnt = ws_alloc<int32_t>(c, nkeys);
            int32_t* cursor = ws_alloc<int32_t>(c, nkeys);
            if (!cnt || !cursor) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
            RLHIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * nkeys, c->stream));
            RLHIP_CHECK(hipMemsetAsync(cursor, 0, sizeof(int32_t) * nkeys, c->stream));
            if (nnz <= 8) hipLaunchKernelGGL(saso_ind_gen_kernel<8>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, d, m, nnz, st, op->rows, cnt);
            else hipLaunchKernelGGL(saso_ind_gen_kernel<128>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, d, m, nnz, st, op->rows, cnt);
            hipLaunchKernelGGL(saso_ind_scan_kernel, dim3((unsigned)op->T), dim3(256), 0, c->stream, d, nnz, cnt, op->ptr, op->T);
            hipLaunchKernelGGL(saso_ind_scatter_kernel, dim3((unsigned)((m * nnz + 255) / 256)), dim3(256), 0, c->stream, d, m, nnz, op->rows,
                               op->ptr, cursor, op->src);
            hipLaunchKernelGGL(saso_ind_sort_kernel, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, c->stream, (int64_t)nkeys, op->ptr, op->src, op->src16);
            RLHIP_LAUNCH_CHECK();
            rlhip_ws_release(c, mark);
        }
        inc = (uint64_t)m * (uint64_t)((nnz + 1) / 2);
    }
    if (next_ctr) {
        uint64_t lo = ((uint64_t)ctr[1] << 32) | ctr[0], hi = ((uint64_t)ctr[3] << 32) | ctr[2];
        uint64_t nlo = lo + inc;
        if (nlo < lo) hi += 1;
        next_ctr[0] = (uint32_t)nlo; next_ctr[1] = (uint32_t)(nlo >> 32);
        next_ctr[2] = (uint32_t)hi; next_ctr[3] = (uint32_t)(hi >> 32);
    }
    *out = op;
    return 0;
}

int saso_destroy(rlhip_ctx* c, SasoOp* op) {
    if (!op) return 0;
    // back to the context's pool: stream-ordered reuse, no device synchronisation (with hipFree the destructor of a sketching operator
    // made the host wait for the apply it had just enqueued -- and the device then idled while the host prepared the next launch:
    // 0.6 ms between the sketch and its pivoted QR in CQRRPT's timeline)
    rlhip_free(c, op->src); rlhip_free(c, op->ainv); rlhip_free(c, op->b); rlhip_free(c, op->afwd); rlhip_free(c, op->rows); rlhip_free(c, op->ptr); rlhip_free(c, op->src16);
    delete op;
    return 0;
}

template <typename T>
int saso_dense(rlhip_ctx* c, const SasoOp* op, T* S /* d x m, zeroed here */) {
    RLHIP_CHECK(hipMemsetAsync(S, 0, sizeof(T) * (size_t)(op->d * op->m), c->stream));
    if (op->mode == 1) {
        if (op->m > 0) {
            hipLaunchKernelGGL(saso_ind_dense_kernel<T>, dim3((unsigned)((op->m * op->nnz + 255) / 256)), dim3(256), 0, c->stream, op->d, op->m, op->nnz,
                               op->rows, S);
            RLHIP_LAUNCH_CHECK();
        }
        return 0;
    }
    int64_t total = op->T * op->nnz * op->d;
    if (total > 0) {
        hipLaunchKernelGGL(saso_dense_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, op->d,
                           op->m, op->T, op->nnz, op->src, S);
        RLHIP_LAUNCH_CHECK();
    }
    return 0;
}

// B (d x n, ldb) = alpha * S[:, row0 : row0 + mloc] * A_loc (mloc x n, lda) + beta * B : the contribution of one row shard
// (the whole product when row0 = 0, mloc = m)
template <typename T>
int saso_apply_rows(rlhip_ctx* c, const SasoOp* op, int64_t n, T alpha, const T* A, int64_t lda, int64_t row0, int64_t mloc, T beta,
                    T* B, int64_t ldb) {
    const int64_t d = op->d, m = op->m;
    if (n <= 0) return 0;
    if (row0 < 0 || mloc < 0 || row0 + mloc > m) return -6;
    if (lda < (mloc > 1 ? mloc : 1)) return -6;
    if (ldb < d) return -9;
    const int64_t tb0 = (mloc > 0) ? row0 / d : 0, tb1 = (mloc > 0) ? (row0 + mloc + d - 1) / d : 0, nTb = tb1 - tb0;
    // columns per workgroup: the d x CT slab has to fit the CU's LDS (4 columns up to d = 5120 fp64 / 10240 fp32, then 2, then 1:
    // d <= 20480 fp64 / 40960 fp32; taller sketches are refused with -2, documented in rlhip.h)
    const size_t lds_cap = 160 * 1024;
    const int CT = (sizeof(T) * (size_t)d * 4 <= lds_cap) ? 4 : (sizeof(T) * (size_t)d * 2 <= lds_cap) ? 2 : 1;
    const size_t smem = sizeof(T) * (size_t)d * CT;
    if (smem > lds_cap) return -2;
    const int64_t ctiles = (n + CT - 1) / CT;
    int64_t G = (1024 + ctiles - 1) / ctiles;                       // ~4 workgroups per CU in flight
    if (G > nTb) G = nTb;
    if (G < 1) G = 1;
    const int64_t tpg = (nTb + G - 1) / G;
    G = (nTb + tpg - 1) / tpg;
    if (G < 1) G = 1;
    // LDS-DMA route (saso_apply_dma_kernel): independent-column operator with 16-bit lists, whole 4-column slabs, 16-byte aligned
    // columns and blocks, d <= 1280 rows (five per thread: with eight the index sets of two blocks no longer fit the registers); its blocks are the ones that lie entirely inside
    // this shard's rows [row0, row0 + mloc) -- a ragged first / last block goes through the register-staged kernel into its own partial group
    const int64_t fb0 = (row0 + d - 1) / d, fb1_raw = (row0 + mloc) / d, fb1 = fb1_raw > fb0 ? fb1_raw : fb0;
    const int64_t nfb = fb1 - fb0;
    const size_t slab2 = 2 * (((size_t)4 * d * sizeof(T) + 1023) & ~(size_t)1023);
    const bool dma = op->mode == 1 && op->src16 && CT == 4 && n % 4 == 0 && d <= 1280 && (d * sizeof(T)) % 16 == 0 && (lda * sizeof(T)) % 16 == 0 &&
                     ((uintptr_t)A % 16) == 0 && (((fb0 * d - row0) * (int64_t)sizeof(T)) % 16) == 0 && slab2 <= lds_cap &&
                     (int64_t)sizeof(T) * (4 * lda + d) < ((int64_t)1 << 32) && nfb >= 8;
    const int head = dma ? (int)(fb0 - tb0) : 0, tail = dma ? (int)(tb1 - fb1) : 0;      // 0 or 1 ragged blocks each
    if (dma) {
        G = (512 + ctiles - 1) / ctiles;                             // two workgroups per CU in flight
        if (G > nfb) G = nfb;
        if (G < 1) G = 1;
    }
    const int64_t tpg_d = dma ? (nfb + G - 1) / G : tpg;
    if (dma) G = (nfb + tpg_d - 1) / tpg_d;
    const int64_t Gtot = G + head + tail;
    size_t mark = rlhip_ws_mark(c);
    T* partial = ws_alloc<T>(c, (size_t)Gtot * d * n);
    if (!partial) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    if (nTb == 0) RLHIP_CHECK(hipMemsetAsync(partial, 0, sizeof(T) * (size_t)(d * n), c->stream));
    // sketch rows per thread and pass: 5 (d <= 1280, the CQRRPT sketches of the benchmark configurations: 40 accumulator registers
    // instead of 64) or 8
    auto launch = [&](auto kern, int nr, T* part, int64_t groups, int64_t per_group, int64_t b0, int64_t b1) -> int {
        RLHIP_FUNC_LDS_DYN(c, kern, lds_cap);       // per kernel ADDRESS: the twelve instantiations share this lambda's statics
        for (int64_t r_base = 0; r_base < d; r_base += 256 * nr)
            hipLaunchKernelGGL(kern, dim3((unsigned)ctiles, (unsigned)groups), dim3(256), smem, c->stream, d, n, m, op->T, op->nnz, op->src, A, lda, per_group, r_base,
                               part, row0, mloc, b0, b1, op->ptr);
        return 0;
    };
    if (nTb > 0) {
        int rc = 0;
        const bool small = d <= 1280;
#define RLHIP_SASO_LAUNCH(CTV, MODEV, PART, GR, PG, B0, B1) (small ? launch(saso_apply_kernel<T, CTV, MODEV, 5>, 5, PART, GR, PG, B0, B1) : launch(saso_apply_kernel<T, CTV, MODEV, 8>, 8, PART, GR, PG, B0, B1))
        if (dma) {
            const int nchunks = (int)((4 * d * sizeof(T) / 16 + 63) / 64), nj = (nchunks + 3) / 4;
            auto launch_dma = [&](auto kern) -> int {
                RLHIP_FUNC_LDS_DYN(c, kern, lds_cap);
                hipLaunchKernelGGL(kern, dim3((unsigned)ctiles, (unsigned)G), dim3(256), slab2, c->stream, d, n, (const uint16_t*)op->src16, (const int32_t*)op->ptr, A, lda,
                                   tpg_d, partial, row0, fb0, fb1);
                return 0;
            };
            if (d == 1280 && sizeof(T) == 8) rc = launch_dma(saso_apply_dma_kernel<T, 5, 10, 1280>);
            else if (nj <= 5) rc = launch_dma(saso_apply_dma_kernel<T, 5, 5, 0>);
            else rc = launch_dma(saso_apply_dma_kernel<T, 5, 10, 0>);
            c->path_count[14]++;
            if (!rc && head) rc = RLHIP_SASO_LAUNCH(4, 1, partial + (size_t)G * d * n, 1, 1, tb0, fb0);
            if (!rc && tail) rc = RLHIP_SASO_LAUNCH(4, 1, partial + (size_t)(G + head) * d * n, 1, 1, fb1, tb1);
        } else if (op->mode == 1) rc = (CT == 4) ? RLHIP_SASO_LAUNCH(4, 1, partial, G, tpg, tb0, tb1) : (CT == 2) ? RLHIP_SASO_LAUNCH(2, 1, partial, G, tpg, tb0, tb1) : RLHIP_SASO_LAUNCH(1, 1, partial, G, tpg, tb0, tb1);
        else rc = (CT == 4) ? RLHIP_SASO_LAUNCH(4, 0, partial, G, tpg, tb0, tb1) : (CT == 2) ? RLHIP_SASO_LAUNCH(2, 0, partial, G, tpg, tb0, tb1) : RLHIP_SASO_LAUNCH(1, 0, partial, G, tpg, tb0, tb1);
#undef RLHIP_SASO_LAUNCH
        if (rc) { rlhip_ws_release(c, mark); return rc; }
    }
    RLHIP_LAUNCH_CHECK();
    const int64_t total = d * n;
    hipLaunchKernelGGL(saso_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, total, (int)Gtot,
                       partial, alpha, beta, B, d, ldb);
    RLHIP_LAUNCH_CHECK();
    rlhip_ws_release(c, mark);
    return 0;
}

template <typename T>
int saso_apply(rlhip_ctx* c, const SasoOp* op, int64_t n, T alpha, const T* A, int64_t lda, T beta, T* B, int64_t ldb) {
    return saso_apply_rows<T>(c, op, n, alpha, A, lda, 0, op->m, beta, B, ldb);
}

// B (d x n, ldb) = alpha * S * A + beta * B for a sparse A (m x n) given by the CSR of its transpose
template <typename T>
int saso_apply_csr(rlhip_ctx* c, const SasoOp* op, int64_t n, T alpha, const int64_t* rowptrT, const int64_t* colidxT, const T* valsT, T beta,
                   T* B, int64_t ldb, int64_t row0) {
    if (row0 < 0 || row0 > op->m) return -6;
    if (n <= 0) return 0;
    if (ldb < op->d) return -9;
    const size_t smem = sizeof(unsigned long long) * (size_t)op->d;
    if (smem > 150 * 1024) return -2;            // d up to 19200
    RLHIP_FUNC_LDS(c, saso_apply_csr_kernel<T>, 150 * 1024);
    hipLaunchKernelGGL(saso_apply_csr_kernel<T>, dim3((unsigned)n), dim3(256), smem, c->stream, op->d, op->m, op->T, op->nnz, op->st, op->afwd,
                       op->b, rowptrT, colidxT, valsT, alpha, beta, B, ldb, row0, op->rows);
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template int saso_apply_csr<double>(rlhip_ctx*, const SasoOp*, int64_t, double, const int64_t*, const int64_t*, const double*, double, double*, int64_t, int64_t);
template int saso_apply_csr<float>(rlhip_ctx*, const SasoOp*, int64_t, float, const int64_t*, const int64_t*, const float*, float, float*, int64_t, int64_t);

// in-place forward column permutation; idx is a DEVICE array of n 1-based indices (left untouched)
// Small matrices (sketch-sized blocks: the finished rows of R_sk in CQRRPT's split QRCP, 512 x 512): the permutation as a parallel gather into
// scratch and a copy back, validity checked on the way (a repeated or out-of-range entry raises the flag and the copy back does nothing --
// the matrix is left as it was, like the cycle walk below).  The serial cycle walk costs 164 us at n = 512 and its apply kernel 241 us, on
// the critical path between the two halves of that solve; this is two launches of a few microseconds.
template <typename T>
__global__ __launch_bounds__(256) void colperm_gather_kernel(int64_t m, int64_t n, const T* __restrict__ A, int64_t lda, const int64_t* __restrict__ idx,
                                                             T* __restrict__ tmp, unsigned* __restrict__ seen, int* __restrict__ bad) {
    const int64_t cidx = blockIdx.y;
    const int64_t sidx = idx[cidx] - 1;
    if (sidx < 0 || sidx >= n) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicExch(bad, 1); return; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned bit = 1u << (sidx & 31);
        if (atomicOr(seen + (sidx >> 5), bit) & bit) atomicExch(bad, 1);
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) tmp[i + cidx * m] = A[i + sidx * lda];
}
template <typename T>
__global__ __launch_bounds__(256) void colperm_store_kernel(int64_t m, const T* __restrict__ tmp, T* __restrict__ A, int64_t lda, const int* __restrict__ bad) {
    if (*bad) return;
    const int64_t cidx = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) A[i + cidx * lda] = tmp[i + cidx * m];
}

template <typename T>
int col_swap(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, T* A, int64_t lda, const int64_t* idx_dev) {
    if (k > n) return -3;                       // reference throws (rl_util.hh:159-160)
    if (m <= 0 || n <= 0) return 0;
    if (n < 4096 && (size_t)m * (size_t)n * sizeof(T) <= ((size_t)32 << 20)) {
        const size_t mk = rlhip_ws_mark(c);
        T* tmp = ws_alloc<T>(c, (size_t)m * n);
        unsigned* seen = ws_alloc<unsigned>(c, (size_t)n / 32 + 2);
        if (tmp && seen) {
            int* bad = (int*)(seen + n / 32 + 1);
            hipError_t he = hipMemsetAsync(seen, 0, ((size_t)n / 32 + 2) * sizeof(unsigned), c->stream);
            if (he == hipSuccess) {
                const dim3 grid((unsigned)std::min<int64_t>((m + 255) / 256, 64), (unsigned)n);
                hipLaunchKernelGGL(colperm_gather_kernel<T>, grid, dim3(256), 0, c->stream, m, n, A, lda, idx_dev, tmp, seen, bad);
                hipLaunchKernelGGL(colperm_store_kernel<T>, grid, dim3(256), 0, c->stream, m, tmp, A, lda, bad);
                he = hipGetLastError();
            }
            rlhip_ws_release(c, mk);
            return he == hipSuccess ? 0 : RLHIP_ERR_HIP(he);
        }
        rlhip_ws_release(c, mk);                // no scratch: the in-place cycle walk
    }
    size_t mark = rlhip_ws_mark(c);
    int64_t* moves = ws_alloc<int64_t>(c, (size_t)(2 * n + 2));
    int64_t* nmoves = ws_alloc<int64_t>(c, 1);
    unsigned char* seen = ws_alloc<unsigned char>(c, (size_t)n);
    if (!moves || !nmoves || !seen) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    std::vector<int64_t> h_chunk;
    int64_t* chunk_dev = nullptr;
    if (n >= 4096) {
        // The cycle decomposition is a serial pointer chase (3-6 ms for one device thread at n = 32768, more than the data movement
        // it steers): for long index vectors it runs on the host instead -- 8n bytes down, the move list up, one stream sync.
        std::vector<int64_t> h_idx((size_t)n), h_moves((size_t)(2 * n + 2));
        std::vector<unsigned char> h_seen((size_t)n, 0);
        RLHIP_CHECK(hipMemcpyAsync(h_idx.data(), idx_dev, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        int64_t w = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (h_seen[(size_t)i]) continue;
            h_seen[(size_t)i] = 1;
            int64_t s_ = h_idx[(size_t)i] - 1;
            if (s_ < 0 || s_ >= n) { rlhip_ws_release(c, mark); return -7; }
            if (s_ == i) continue;
            h_moves[(size_t)w++] = -(i + 1);
            int64_t j = i;
            while (s_ != i) {
                h_moves[(size_t)w++] = s_;
                h_seen[(size_t)s_] = 1;
                j = s_;
                s_ = h_idx[(size_t)j] - 1;
                if (s_ < 0 || s_ >= n || (s_ != i && h_seen[(size_t)s_])) { rlhip_ws_release(c, mark); return -7; }   // not a permutation
            }
        }
        // chunks of whole cycles for the apply kernel's second grid dimension: ~1024 workgroups in all, at least 64 moves per chunk
        {
            const int64_t gx = (m + 255) / 256;
            int64_t want = gx >= 1024 ? 1 : (1024 + gx - 1) / gx;
            if (want > 256) want = 256;
            const int64_t per = std::max<int64_t>(64, (w + want - 1) / std::max<int64_t>(want, 1));
            h_chunk.clear(); h_chunk.push_back(0);
            for (int64_t q = 1; q < w; ++q)
                if (h_moves[(size_t)q] < 0 && q - h_chunk.back() >= per) h_chunk.push_back(q);
            h_chunk.push_back(w);
        }
        h_moves[(size_t)(2 * n + 1)] = w;                       // nmoves travels in the same copy when adjacent
        RLHIP_CHECK(hipMemcpyAsync(moves, h_moves.data(), sizeof(int64_t) * (size_t)std::max<int64_t>(w, 1), hipMemcpyHostToDevice, c->stream));
        RLHIP_CHECK(hipMemcpyAsync(nmoves, &h_moves[(size_t)(2 * n + 1)], sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        if (h_chunk.size() > 2) {
            chunk_dev = ws_alloc<int64_t>(c, h_chunk.size());
            if (chunk_dev) RLHIP_CHECK(hipMemcpyAsync(chunk_dev, h_chunk.data(), sizeof(int64_t) * h_chunk.size(), hipMemcpyHostToDevice, c->stream));
        }
        RLHIP_CHECK(rlhip_stream_sync(c));           // the host vectors die at the end of this scope
    } else
        hipLaunchKernelGGL(perm_moves_kernel, dim3(1), dim3(1), 0, c->stream, n, idx_dev, moves, nmoves, seen);
    hipLaunchKernelGGL(perm_apply_kernel<T>, dim3((unsigned)((m + 255) / 256), (unsigned)(chunk_dev ? h_chunk.size() - 1 : 1)), dim3(256), 0, c->stream, m, A, lda,
                       moves, nmoves, (const int64_t*)chunk_dev);
    RLHIP_LAUNCH_CHECK();
    rlhip_ws_release(c, mark);
    return 0;
}

int col_swap_i64(rlhip_ctx* c, int64_t n, int64_t k, int64_t* A, const int64_t* idx_dev) {
    if (k > n) return -3;
    if (k <= 0) return 0;
    size_t mark = rlhip_ws_mark(c);
    int64_t* tmp = ws_alloc<int64_t>(c, (size_t)k);
    if (!tmp) return RLHIP_ERR_HIP(hipErrorOutOfMemory);
    hipLaunchKernelGGL(vec_gather_i64_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, k, A, idx_dev, tmp);
    RLHIP_CHECK(hipMemcpyAsync(A, tmp, sizeof(int64_t) * (size_t)k, hipMemcpyDeviceToDevice, c->stream));
    rlhip_ws_release(c, mark);
    return 0;
}

template int saso_dense<double>(rlhip_ctx*, const SasoOp*, double*);
template int saso_dense<float>(rlhip_ctx*, const SasoOp*, float*);
template int saso_apply_rows<double>(rlhip_ctx*, const SasoOp*, int64_t, double, const double*, int64_t, int64_t, int64_t, double, double*, int64_t);
template int saso_apply_rows<float>(rlhip_ctx*, const SasoOp*, int64_t, float, const float*, int64_t, int64_t, int64_t, float, float*, int64_t);
template int saso_apply<double>(rlhip_ctx*, const SasoOp*, int64_t, double, const double*, int64_t, double, double*, int64_t);
template int saso_apply<float>(rlhip_ctx*, const SasoOp*, int64_t, float, const float*, int64_t, float, float*, int64_t);
template int col_swap<double>(rlhip_ctx*, int64_t, int64_t, int64_t, double*, int64_t, const int64_t*);
template int col_swap<float>(rlhip_ctx*, int64_t, int64_t, int64_t, float*, int64_t, const int64_t*);

}  // namespace rlhip
