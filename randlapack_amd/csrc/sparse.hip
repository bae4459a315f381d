// Sparse operator kernels behind linops::SparseLinOp (reference: RandLAPACK/linops/rl_sparse_linop.hh:125-330, which
// forwards to RandBLAS left_spmm/right_spmm -- RandBLAS is an absent dependency; the contract restated here is
// "C = alpha * op(A_sparse) * B + beta * C").  Device design:
//
//   * the operator is held as CSR together with the CSR of its transpose, so that both A*X and A^T*X are a gather over the rows
//     of one of the two structures (no atomics; the summation order inside a row is the storage order -> bit-reproducible);
//   * dense operands are ROW-major inside the kernel: one wave owns one CSR row, its 64 lanes own 64*NC consecutive columns of
//     the output row, and every nonzero (col, v) contributes the contiguous row B[col, :] -- each load is a full 512-byte
//     coalesced segment instead of 64 scattered 8-byte reads.  A column-major caller goes through the LDS-tiled transpose
//     (svd.hip) on the way in / out; that costs 2 extra passes over the dense operand, the gather costs nnz(row)/... passes.
//   * (col, v) pairs are fetched 64 at a time (coalesced) and broadcast with v_readlane.
//
// HBM-bound: algorithmic bytes = nnz * ncols * sizeof(T) (gather) + rows * ncols * sizeof(T) (store).
#include "rlhip_internal.h"
#include <cstdlib>
#include "../../include/rlhip.h"

namespace rlhip {

__device__ __forceinline__ int64_t bcast_i64(int64_t x, int t) {
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), t);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), t);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ double bcast_val(double x, int t) {
    return __longlong_as_double(bcast_i64(__double_as_longlong(x), t));
}
__device__ __forceinline__ float bcast_val(float x, int t) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), t));
}

// C (nrows x nc, row-major, ldc) = alpha * A (CSR, nrows x k) * B (k x nc, row-major, ldb) + beta * C
template <typename T, int NC>
__global__ __launch_bounds__(256) void csr_spmm_rm_kernel(int64_t nrows, int64_t nc, const int64_t* __restrict__ rowptr,
                                                          const int64_t* __restrict__ colidx, const T* __restrict__ vals, T alpha,
                                                          const T* __restrict__ B, int64_t ldb, T beta, T* __restrict__ C, int64_t ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t jb = (int64_t)blockIdx.y * (64 * NC);
    int64_t jq[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int64_t j = jb + lane + 64 * q;
        jq[q] = j < nc ? j : nc - 1;   // clamp (no per-element branch around the loads); masked at the store
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < nrows; row += (int64_t)gridDim.x * 4) {
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        T acc[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[q] = (T)0;
        for (int64_t p = p0; p < p1; p += 64) {
            const int64_t pc = p + lane;
            const bool in = pc < p1;
            const int64_t pcl = in ? pc : p1 - 1;
            int64_t ci = colidx[pcl];
            T v = in ? vals[pcl] : (T)0;
            const int cnt = (int)((p1 - p) < 64 ? (p1 - p) : 64);
            int t = 0;
            for (; t + 1 < cnt; t += 2) {   // two independent rows of B in flight
                const int64_t c0 = bcast_i64(ci, t), c1 = bcast_i64(ci, t + 1);
                const T v0 = bcast_val(v, t), v1 = bcast_val(v, t + 1);
                const T* b0 = B + c0 * ldb;
                const T* b1 = B + c1 * ldb;
                T x0[NC], x1[NC];
#pragma unroll
                for (int q = 0; q < NC; ++q) { x0[q] = b0[jq[q]]; x1[q] = b1[jq[q]]; }
#pragma unroll
                for (int q = 0; q < NC; ++q) { acc[q] += v0 * x0[q]; acc[q] += v1 * x1[q]; }
            }
            if (t < cnt) {
                const int64_t c0 = bcast_i64(ci, t);
                const T v0 = bcast_val(v, t);
                const T* b0 = B + c0 * ldb;
#pragma unroll
                for (int q = 0; q < NC; ++q) acc[q] += v0 * b0[jq[q]];
            }
        }
        T* crow = C + row * ldc;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int64_t j = jb + lane + 64 * q;
            if (j < nc) crow[j] = (beta == (T)0) ? alpha * acc[q] : alpha * acc[q] + beta * crow[j];
        }
    }
}

// Narrow right-hand sides (nc <= 32: ABRIK's Krylov blocks, rl_abrik.hh:311): a wavefront per CSR row would leave half (nc = 32) or three
// quarters (nc = 16) of its lanes clamped and masked.  Here W = 32 or 16 lanes own a row and a wavefront takes 64 / W rows at once; every
// lane group walks its own row (the (column, value) pair of an entry is ONE address for the whole group: the memory pipeline broadcasts
// it), two entries in flight, and every nonzero still pulls one contiguous row segment of the row-major operand.  Same summation order
// as the wide kernel (entry by entry along the row): bitwise the same products.
template <typename T, int W>
__global__ __launch_bounds__(256) void csr_spmm_rm_narrow_kernel(int64_t nrows, int64_t nc, const int64_t* __restrict__ rowptr,
                                                                 const int64_t* __restrict__ colidx, const T* __restrict__ vals, T alpha,
                                                                 const T* __restrict__ B, int64_t ldb, T beta, T* __restrict__ C, int64_t ldc) {
    constexpr int RPW = 64 / W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % W, sr = lane / W;
    const int64_t j = sub < nc ? sub : nc - 1;          // clamp; masked at the store
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; row0 < nrows; row0 += (int64_t)gridDim.x * 4 * RPW) {
        const int64_t row = row0 + sr;
        const bool valid = row < nrows;
        const int64_t rl = valid ? row : nrows - 1;
        const int64_t p0 = rowptr[rl], p1 = valid ? rowptr[rl + 1] : p0;
        T acc = (T)0;
        int64_t p = p0;
        for (; p + 1 < p1; p += 2) {
            const int64_t c0 = colidx[p], c1 = colidx[p + 1];
            const T v0 = vals[p], v1 = vals[p + 1];
            const T x0 = B[c0 * ldb + j], x1 = B[c1 * ldb + j];
            acc += v0 * x0;
            acc += v1 * x1;
        }
        if (p < p1) acc += vals[p] * B[colidx[p] * ldb + j];
        if (valid && sub < nc) {
            T* cr = C + row * ldc + sub;
            *cr = (beta == (T)0) ? alpha * acc : alpha * acc + beta * *cr;
        }
    }
}

// The same product for a COLUMN-major C (the layout every caller above the C entry uses): a workgroup takes 64 consecutive rows, its four
// wavefronts walk them 64 / W at a time exactly as above, and the 64 x nc block of results crosses LDS once so that every column leaves
// as one 512-byte run -- the separate transpose pass of C (a read and a write of the whole block) is gone.  Same sums, same order.
template <typename T, int W>
__global__ __launch_bounds__(256) void csr_spmm_cmout_narrow_kernel(int64_t nrows, int64_t nc, const int64_t* __restrict__ rowptr,
                                                                    const int64_t* __restrict__ colidx, const T* __restrict__ vals, T alpha,
                                                                    const T* __restrict__ B, int64_t ldb, T beta, T* __restrict__ C, int64_t ldc) {
    constexpr int RPW = 64 / W, RB = 64;
    __shared__ T tile[W][RB + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % W, sr = lane / W;
    const int64_t j = sub < nc ? sub : nc - 1;
    for (int64_t blk = blockIdx.x; blk * RB < nrows; blk += gridDim.x) {
        const int64_t r0 = blk * RB;
#pragma unroll 2
        for (int pass = 0; pass < RB / (4 * RPW); ++pass) {
            const int rl = (pass * 4 + wave) * RPW + sr;
            const int64_t row = r0 + rl;
            const bool valid = row < nrows;
            const int64_t rr = valid ? row : nrows - 1;
            const int64_t p0 = rowptr[rr], p1 = valid ? rowptr[rr + 1] : p0;
            T acc = (T)0;
            int64_t p = p0;
            for (; p + 3 < p1; p += 4) {                 // four entries in flight: the chain is index -> operand row, twice the depth of the kernel above
                const int64_t c0 = colidx[p], c1 = colidx[p + 1], c2 = colidx[p + 2], c3 = colidx[p + 3];
                const T v0 = vals[p], v1 = vals[p + 1], v2 = vals[p + 2], v3 = vals[p + 3];
                const T x0 = B[c0 * ldb + j], x1 = B[c1 * ldb + j], x2 = B[c2 * ldb + j], x3 = B[c3 * ldb + j];
                acc += v0 * x0;
                acc += v1 * x1;
                acc += v2 * x2;
                acc += v3 * x3;
            }
            for (; p + 1 < p1; p += 2) {
                const int64_t c0 = colidx[p], c1 = colidx[p + 1];
                const T v0 = vals[p], v1 = vals[p + 1];
                const T x0 = B[c0 * ldb + j], x1 = B[c1 * ldb + j];
                acc += v0 * x0;
                acc += v1 * x1;
            }
            if (p < p1) acc += vals[p] * B[colidx[p] * ldb + j];
            tile[sub][rl] = acc;
        }
        __syncthreads();
        const int rl = threadIdx.x & 63;
        if (r0 + rl < nrows) {
            for (int col = threadIdx.x >> 6; col < nc; col += 4) {
                T* cp = C + (r0 + rl) + (int64_t)col * ldc;
                const T a = alpha * tile[col][rl];
                *cp = (beta == (T)0) ? a : a + beta * *cp;
            }
        }
        __syncthreads();
    }
}

template <typename T>
int csr_spmm_rowmajor(rlhip_ctx* c, int64_t nrows, int64_t nc, const int64_t* rowptr, const int64_t* colidx, const T* vals, T alpha,
                      const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {
    if (nrows <= 0 || nc <= 0) return 0;
    const int64_t gx = std::min<int64_t>((nrows + 3) / 4, 256 * 64);
    if (nc <= 32) {
        const int rpw = nc <= 16 ? 4 : 2;
        dim3 grid((unsigned)std::min<int64_t>((nrows + 4 * rpw - 1) / (4 * rpw), 256 * 64), 1);
        if (nc <= 16) hipLaunchKernelGGL((csr_spmm_rm_narrow_kernel<T, 16>), grid, dim3(256), 0, c->stream, nrows, nc, rowptr, colidx, vals, alpha, B, ldb, beta, C, ldc);
        else hipLaunchKernelGGL((csr_spmm_rm_narrow_kernel<T, 32>), grid, dim3(256), 0, c->stream, nrows, nc, rowptr, colidx, vals, alpha, B, ldb, beta, C, ldc);
        RLHIP_LAUNCH_CHECK();
        return 0;
    }
    if (nc > 128) {
        dim3 grid((unsigned)gx, (unsigned)((nc + 255) / 256));
        hipLaunchKernelGGL((csr_spmm_rm_kernel<T, 4>), grid, dim3(256), 0, c->stream, nrows, nc, rowptr, colidx, vals, alpha, B, ldb, beta, C, ldc);
    } else if (nc > 64) {
        dim3 grid((unsigned)gx, 1);
        hipLaunchKernelGGL((csr_spmm_rm_kernel<T, 2>), grid, dim3(256), 0, c->stream, nrows, nc, rowptr, colidx, vals, alpha, B, ldb, beta, C, ldc);
    } else {
        dim3 grid((unsigned)gx, 1);
        hipLaunchKernelGGL((csr_spmm_rm_kernel<T, 1>), grid, dim3(256), 0, c->stream, nrows, nc, rowptr, colidx, vals, alpha, B, ldb, beta, C, ldc);
    }
    RLHIP_LAUNCH_CHECK();
    return 0;
}

// C (nrows x nc) = alpha * A_csr * B (k x nc) + beta * C with column-major B, C: transposes through scratch.
template <typename T>
int csr_spmm(rlhip_ctx* c, int layout_rowmajor, int64_t nrows, int64_t k, int64_t nc, const int64_t* rowptr, const int64_t* colidx,
             const T* vals, T alpha, const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {
    if (nrows <= 0 || nc <= 0) return 0;
    if (layout_rowmajor) return csr_spmm_rowmajor<T>(c, nrows, nc, rowptr, colidx, vals, alpha, B, ldb, beta, C, ldc);
    const size_t mark = rlhip_ws_mark(c);
    if (nc <= 32) {
        T* Bt = ws_alloc<T>(c, (size_t)std::max<int64_t>(k, 1) * nc);
        if (!Bt) { rlhip_ws_release(c, mark); return -3; }
        int rc = 0;
        if (k > 0) rc = transpose<T>(c, k, nc, B, ldb, Bt, nc, 0);
        if (!rc) {
            dim3 grid((unsigned)std::min<int64_t>((nrows + 63) / 64, 256 * 64), 1);
            if (nc <= 16) hipLaunchKernelGGL((csr_spmm_cmout_narrow_kernel<T, 16>), grid, dim3(256), 0, c->stream, nrows, nc, rowptr, colidx, vals, alpha, Bt, nc, beta, C, ldc);
            else hipLaunchKernelGGL((csr_spmm_cmout_narrow_kernel<T, 32>), grid, dim3(256), 0, c->stream, nrows, nc, rowptr, colidx, vals, alpha, Bt, nc, beta, C, ldc);
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) rc = RLHIP_ERR_HIP(le);
        }
        rlhip_ws_release(c, mark);
        return rc;
    }
    T* Bt = ws_alloc<T>(c, (size_t)std::max<int64_t>(k, 1) * nc);
    T* Ct = ws_alloc<T>(c, (size_t)nrows * nc);
    if (!Bt || !Ct) { rlhip_ws_release(c, mark); return -3; }
    int rc = 0;
    if (k > 0) rc = transpose<T>(c, k, nc, B, ldb, Bt, nc, 0);          // Bt is nc x k column-major == k x nc row-major
    if (!rc && beta != (T)0) rc = transpose<T>(c, nrows, nc, C, ldc, Ct, nc, 0);
    if (!rc) rc = csr_spmm_rowmajor<T>(c, nrows, nc, rowptr, colidx, vals, alpha, Bt, nc, beta, Ct, nc);
    if (!rc) rc = transpose<T>(c, nc, nrows, Ct, nc, C, ldc, 0);
    rlhip_ws_release(c, mark);
    return rc;
}

// columns [c0, c0 + b) of the sparse matrix as a dense column-major block: out (m x b, ldo) from rows c0.. of the TRANSPOSE's CSR
template <typename T>
__global__ __launch_bounds__(256) void csr_densify_kernel(int64_t m, const int64_t* __restrict__ rowptrT, const int64_t* __restrict__ colidxT,
                                                          const T* __restrict__ valsT, int64_t c0, T* __restrict__ out, int64_t ldo) {
    const int64_t c = blockIdx.x;
    T* col = out + c * ldo;
    for (int64_t i = threadIdx.x; i < m; i += 256) col[i] = (T)0;
    __syncthreads();
    const int64_t p0 = rowptrT[c0 + c], p1 = rowptrT[c0 + c + 1];
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) atomicAdd(&col[colidxT[p]], valsT[p]);   // duplicates (if any) sum
}

template <typename T>
int csr_densify_cols(rlhip_ctx* c, int64_t m, const int64_t* rowptrT, const int64_t* colidxT, const T* valsT, int64_t c0, int64_t b, T* out,
                     int64_t ldo) {
    if (m <= 0 || b <= 0) return 0;
    hipLaunchKernelGGL(csr_densify_kernel<T>, dim3((unsigned)b), dim3(256), 0, c->stream, m, rowptrT, colidxT, valsT, c0, out, ldo);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

// ---- CSR of the transpose (k x m) of an m x k CSR matrix: a STABLE counting sort of the entries by column, entirely on the device.
// Stability (entries of one transposed row keep the source order = ascending source row) makes the summation order of A^T X
// independent of scheduling, so results are bit-reproducible.  Three phases over NB contiguous chunks of the entry list:
//   count    cnt[b][j] = entries of chunk b in column j (integer atomics: order-free)
//   scan     cnt[b][j] <- entries of column j in chunks < b; rowptrT = exclusive scan of the column totals
//   scatter  one WAVE per chunk walks its entries 64 at a time in order; a lane's slot is
//            rowptrT[col] + cnt[b][col] + (number of lower lanes of the tile with the same column)
__global__ __launch_bounds__(256) void ct_count_kernel(int64_t nnz, int64_t per, int64_t k, const int64_t* __restrict__ colidx,
                                                       int* __restrict__ cnt, int* __restrict__ bad) {
    const int64_t b = blockIdx.x;
    const int64_t p1 = (b + 1) * per < nnz ? (b + 1) * per : nnz;
    for (int64_t p = b * per + threadIdx.x; p < p1; p += 256) {
        const int64_t c = colidx[p];
        if (c < 0 || c >= k) { *bad = 1; continue; }
        atomicAdd(&cnt[b * k + c], 1);
    }
}
__global__ void ct_chunk_scan_kernel(int64_t nb, int64_t k, int* __restrict__ cnt, int64_t* __restrict__ total) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    int run = 0;
    int64_t b = 0;
    for (; b + 8 <= nb; b += 8) {                      // eight loads in flight per thread (one at a time the walk over 333 chunks of a
        int t[8];                                      // 200000-column operator took 209 us for 533 MB)
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = cnt[(b + u) * k + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) { cnt[(b + u) * k + j] = run; run += t[u]; }
    }
    for (; b < nb; ++b) {
        const int t = cnt[b * k + j];
        cnt[b * k + j] = run;
        run += t;
    }
    total[j] = run;
}
// rowptrT[0..k] = exclusive scan of total[0..k) in three coalesced launches (a one-workgroup kernel that walked a strided slice per thread
// took 344 us for 200000 columns, a third of the transpose ABRIK builds for its A^T X products): block sums of 1024 totals -> their exclusive scan (one workgroup) -> block scans + offsets
__device__ __forceinline__ int64_t ct_block_scan_incl(int64_t v, int64_t* part, int t) {     // inclusive scan over the 1024 threads of a workgroup
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t a = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += a;
        __syncthreads();
    }
    return part[t];
}
__global__ __launch_bounds__(1024) void ct_blocksum_kernel(int64_t k, const int64_t* __restrict__ total, int64_t* __restrict__ bsum,
                                                           unsigned long long* __restrict__ maxlen) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * 1024 + t;
    const int64_t v = j < k ? total[j] : 0;
    const int64_t incl = ct_block_scan_incl(v, part, t);
    if (t == 1023) bsum[blockIdx.x] = incl;
    if (maxlen) {                                       // longest transposed row (decides between the two scatter routes)
        __syncthreads();
        part[t] = v;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) {
            if (t < off && part[t + off] > part[t]) part[t] = part[t + off];
            __syncthreads();
        }
        if (t == 0) atomicMax(maxlen, (unsigned long long)part[0]);
    }
}
__global__ __launch_bounds__(1024) void ct_blockscan_kernel(int64_t nblk, int64_t* __restrict__ bsum) {     // in place: exclusive scan; bsum[nblk] = grand total
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblk; b0 += 1024) {
        const int64_t b = b0 + t;
        const int64_t v = b < nblk ? bsum[b] : 0;
        const int64_t incl = ct_block_scan_incl(v, part, t);
        const int64_t base = carry;
        if (b < nblk) bsum[b] = base + incl - v;
        __syncthreads();
        if (t == 1023) carry = base + incl;
        __syncthreads();
    }
    if (t == 0) bsum[nblk] = carry;
}
__global__ __launch_bounds__(1024) void ct_rowptr3_kernel(int64_t k, const int64_t* __restrict__ total, const int64_t* __restrict__ bsum, int64_t nblk,
                                                          int64_t* __restrict__ rowptrT) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * 1024 + t;
    const int64_t v = j < k ? total[j] : 0;
    const int64_t incl = ct_block_scan_incl(v, part, t);
    if (j < k) rowptrT[j] = bsum[blockIdx.x] + incl - v;
    if (blockIdx.x == 0 && t == 0) rowptrT[k] = bsum[nblk];
}
// rowid[p] = source row of entry p (one thread per row; the scatter used to find it by a 17-step binary search over rowptr per entry)
__global__ __launch_bounds__(256) void ct_rowid_kernel(int64_t m, const int64_t* __restrict__ rowptr, int64_t* __restrict__ rowid) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const int64_t p1 = rowptr[r + 1];
    for (int64_t p = rowptr[r]; p < p1; ++p) rowid[p] = r;
}
// ---- short transposed rows (the usual case: no column of A holds more than CT_SORT_MAX entries): histogram -> scan -> scatter of the
// entry NUMBERS through one returning atomic per entry -> every transposed row sorted by entry number by its own thread -> rows and values
// gathered.  Sorted entry numbers = ascending source rows = exactly the order of the stable counting sort above (bitwise the same
// transpose), without its nb x k counter table: for ABRIK's 200000 x 200000 operator with 2e6 nonzeros that table was 266 MB to clear,
// fill and scan (0.5 ms of the 0.55 ms transpose).
constexpr int64_t CT_SORT_MAX = 512;
__global__ __launch_bounds__(256) void ct_hist_kernel(int64_t nnz, int64_t k, const int64_t* __restrict__ colidx, unsigned long long* __restrict__ total,
                                                      int* __restrict__ bad) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * 256) {
        const int64_t c = colidx[p];
        if (c < 0 || c >= k) { *bad = 1; continue; }
        atomicAdd(&total[c], 1ull);
    }
}
__global__ __launch_bounds__(256) void ct_scatter_key_kernel(int64_t nnz, const int64_t* __restrict__ colidx, const int64_t* __restrict__ rowptrT,
                                                             int* __restrict__ cursor, int64_t* __restrict__ key) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * 256) {
        const int64_t c = colidx[p];
        key[rowptrT[c] + atomicAdd(&cursor[c], 1)] = p;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void ct_sortfill_kernel(int64_t k, const int64_t* __restrict__ rowptrT, int64_t* __restrict__ colidxT,
                                                          const int64_t* __restrict__ rowid, const T* __restrict__ vals, T* __restrict__ valsT) {
    __shared__ int64_t slab[256 * 17];                      // rows of <= 16 entries are sorted in a private LDS strip (stride 17: no bank conflicts)
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= k) return;
    const int64_t p0 = rowptrT[c], p1 = rowptrT[c + 1];
    if (p1 - p0 <= 16) {
        int64_t* my = slab + threadIdx.x * 17;
        const int len = (int)(p1 - p0);
        for (int q = 0; q < len; ++q) my[q] = colidxT[p0 + q];
        for (int q = 1; q < len; ++q) {
            const int64_t e = my[q];
            int r = q - 1;
            while (r >= 0 && my[r] > e) { my[r + 1] = my[r]; --r; }
            my[r + 1] = e;
        }
        for (int q = 0; q < len; ++q) {
            const int64_t p = my[q];
            valsT[p0 + q] = vals[p];
            colidxT[p0 + q] = rowid[p];
        }
        return;
    }
    for (int64_t q = p0 + 1; q < p1; ++q) {                 // insertion sort of the entry numbers (colidxT holds them until the loop below)
        const int64_t e = colidxT[q];
        int64_t r = q - 1;
        while (r >= p0 && colidxT[r] > e) { colidxT[r + 1] = colidxT[r]; --r; }
        colidxT[r + 1] = e;
    }
    for (int64_t q = p0; q < p1; ++q) {
        const int64_t p = colidxT[q];
        valsT[q] = vals[p];
        colidxT[q] = rowid[p];
    }
}

template <typename T>
__global__ __launch_bounds__(64) void ct_scatter_kernel(int64_t m, int64_t nnz, int64_t per, int64_t k, const int64_t* __restrict__ rowptr,
                                                        const int64_t* __restrict__ colidx, const T* __restrict__ vals,
                                                        const int64_t* __restrict__ rowptrT, int* __restrict__ cnt,
                                                        int64_t* __restrict__ colidxT, T* __restrict__ valsT, const int64_t* __restrict__ rowid) {
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t p0 = b * per, p1 = (b + 1) * per < nnz ? (b + 1) * per : nnz;
    int* mycnt = cnt + b * k;
    for (int64_t pt = p0; pt < p1; pt += 64) {
        const int64_t p = pt + lane;
        const bool valid = p < p1;
        const int c = valid ? (int)colidx[p] : -1 - lane;          // invalid lanes get distinct sentinels
        int rank = 0, same = 0;
        for (int l = 0; l < 64; ++l) {
            const int cl = __builtin_amdgcn_readlane(c, l);
            const int eq = (cl == c) ? 1 : 0;
            rank += (l < lane) ? eq : 0;
            same += eq;
        }
        if (valid) {
            // per-chunk running offsets live in global memory; this wave is their only writer, agent-scope atomics keep the
            // read-after-write of the next tile coherent (the vector L1 is not)
            const int before = __hip_atomic_load(&mycnt[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int64_t dst = rowptrT[c] + before + rank;
            colidxT[dst] = rowid[p];                                   // source row of entry p (ct_rowid_kernel)
            valsT[dst] = vals[p];
            if (rank == same - 1) __hip_atomic_store(&mycnt[c], before + same, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename T>
int csr_transpose(rlhip_ctx* c, int64_t m, int64_t k, const int64_t* rowptr, const int64_t* colidx, const T* vals, int64_t* rowptrT,
                  int64_t* colidxT, T* valsT) {
    if (k >= ((int64_t)1 << 31)) return -2;
    int64_t nnz = 0;
    if (m > 0) {
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 50, rowptr + m, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        nnz = c->h_mail[50];
    }
    if (nnz < 0 || nnz >= ((int64_t)1 << 31)) return -2;      // per-column counters are 32-bit
    if (k == 0) { RLHIP_CHECK(hipMemsetAsync(rowptrT, 0, sizeof(int64_t), c->stream)); return nnz == 0 ? 0 : -2; }
    const int64_t nblk = (k + 1023) / 1024;
    int* d_bad = (int*)(c->d_mail + 51);
    unsigned long long* d_maxlen = (unsigned long long*)(c->d_mail + 52);
    const size_t mark = rlhip_ws_mark(c);
    int64_t* total = ws_alloc<int64_t>(c, (size_t)k);
    int64_t* bsum = ws_alloc<int64_t>(c, (size_t)nblk + 1);
    int64_t* rowid = ws_alloc<int64_t>(c, (size_t)(nnz > 0 ? nnz : 1));
    int* cursor = ws_alloc<int>(c, (size_t)k);
    if (!total || !bsum || !rowid || !cursor) { rlhip_ws_release(c, mark); return -3; }
    int rc = 0;
    do {
        // ---- column totals, row pointers of the transpose, the longest transposed row (one host read for it and the index check)
        if (hipMemsetAsync(total, 0, sizeof(int64_t) * (size_t)k, c->stream) != hipSuccess || hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)k, c->stream) != hipSuccess ||
            hipMemsetAsync(c->d_mail + 51, 0, 2 * sizeof(int64_t), c->stream) != hipSuccess) { rc = -1; break; }
        const unsigned gh = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (nnz + 255) / 256));
        if (nnz > 0) hipLaunchKernelGGL(ct_hist_kernel, dim3(gh), dim3(256), 0, c->stream, nnz, k, colidx, (unsigned long long*)total, d_bad);
        hipLaunchKernelGGL(ct_blocksum_kernel, dim3((unsigned)nblk), dim3(1024), 0, c->stream, k, (const int64_t*)total, bsum, d_maxlen);
        hipLaunchKernelGGL(ct_blockscan_kernel, dim3(1), dim3(1024), 0, c->stream, nblk, bsum);
        hipLaunchKernelGGL(ct_rowptr3_kernel, dim3((unsigned)nblk), dim3(1024), 0, c->stream, k, (const int64_t*)total, (const int64_t*)bsum, nblk, rowptrT);
        if (nnz > 0 && m == 1) {
            // one source row (SparseLinOp::from_coo sorts a COO list as a 1 x rows matrix): every entry's source row is 0 -- one memset instead
            // of ONE thread writing nnz words
            if (hipMemsetAsync(rowid, 0, (size_t)nnz * sizeof(int64_t), c->stream) != hipSuccess) { rc = -1; break; }
        } else
        if (nnz > 0 && m > 0) hipLaunchKernelGGL(ct_rowid_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, m, rowptr, rowid);
        if (hipMemcpyAsync(c->h_mail + 51, c->d_mail + 51, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess || rlhip_stream_sync(c) != hipSuccess) { rc = -1; break; }
        if (*(int*)(c->h_mail + 51)) { rc = -2; break; }                                   // a column index outside [0, k)
        if (nnz == 0) break;
        if (c->h_mail[52] <= CT_SORT_MAX) {
            hipLaunchKernelGGL(ct_scatter_key_kernel, dim3(gh), dim3(256), 0, c->stream, nnz, colidx, (const int64_t*)rowptrT, cursor, colidxT);
            hipLaunchKernelGGL(ct_sortfill_kernel<T>, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, k, (const int64_t*)rowptrT, colidxT, (const int64_t*)rowid, vals, valsT);
            if (hipGetLastError() != hipSuccess) rc = -1;
            break;
        }
        // ---- a long transposed row somewhere: the stable counting sort over chunks of the entry list
        // chunks: enough waves to fill the chip, bounded so that the nb x k counter table stays under 256 MiB
        int64_t nb = std::min<int64_t>(4096, std::max<int64_t>(1, ((int64_t)1 << 26) / k));
        nb = std::max<int64_t>(1, std::min<int64_t>(nb, (nnz + 63) / 64));
        const int64_t per = std::max<int64_t>(64, ((nnz + nb - 1) / nb + 63) / 64 * 64);
        nb = std::max<int64_t>(1, (nnz + per - 1) / per);
        if (per >= ((int64_t)1 << 31)) { rc = -2; break; }
        int* cnt = ws_alloc<int>(c, (size_t)nb * k);
        if (!cnt) { rc = -3; break; }
        if (hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)nb * k, c->stream) != hipSuccess) { rc = -1; break; }
        hipLaunchKernelGGL(ct_count_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, nnz, per, k, colidx, cnt, d_bad);
        hipLaunchKernelGGL(ct_chunk_scan_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, nb, k, cnt, total);   // (total: the same values again)
        hipLaunchKernelGGL(ct_scatter_kernel<T>, dim3((unsigned)nb), dim3(64), 0, c->stream, m, nnz, per, k, rowptr, colidx, vals, rowptrT, cnt, colidxT, valsT, (const int64_t*)rowid);
        if (hipGetLastError() != hipSuccess) rc = -1;
    } while (0);
    rlhip_ws_release(c, mark);
    return rc;
}

#define INST(T)                                                                                                                         \
    template int csr_spmm<T>(rlhip_ctx*, int, int64_t, int64_t, int64_t, const int64_t*, const int64_t*, const T*, T, const T*, int64_t, \
                             T, T*, int64_t);                                                                                           \
    template int csr_densify_cols<T>(rlhip_ctx*, int64_t, const int64_t*, const int64_t*, const T*, int64_t, int64_t, T*, int64_t);      \
    template int csr_transpose<T>(rlhip_ctx*, int64_t, int64_t, const int64_t*, const int64_t*, const T*, int64_t*, int64_t*, T*);
INST(double)
INST(float)
#undef INST

}  // namespace rlhip

#define CAPI(T, SUF)                                                                                                                    \
    extern "C" int rlhip_csr_spmm_##SUF(rlhip_ctx* c, char layout, int64_t m, int64_t n, int64_t k, T alpha, const int64_t* rowptr,    \
                                        const int64_t* colidx, const T* vals, const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {     \
        if (!c || m < 0 || n < 0 || k < 0 || (layout != 'C' && layout != 'R')) return -2;                                               \
        if (layout == 'C' ? (ldb < (k > 1 ? k : 1) || ldc < (m > 1 ? m : 1)) : (ldb < (n > 1 ? n : 1) || ldc < (n > 1 ? n : 1))) return -2; \
        return rlhip::csr_spmm<T>(c, layout == 'R', m, k, n, rowptr, colidx, vals, alpha, B, ldb, beta, C, ldc);                        \
    }                                                                                                                                   \
    extern "C" int rlhip_csr_densify_cols_##SUF(rlhip_ctx* c, int64_t m, const int64_t* rowptrT, const int64_t* colidxT, const T* valsT, \
                                                int64_t c0, int64_t b, T* out, int64_t ldo) {                                           \
        if (!c || m < 0 || b < 0 || c0 < 0 || ldo < (m > 1 ? m : 1)) return -2;                                                         \
        return rlhip::csr_densify_cols<T>(c, m, rowptrT, colidxT, valsT, c0, b, out, ldo);                                              \
    }                                                                                                                                   \
    extern "C" int rlhip_csr_transpose_##SUF(rlhip_ctx* c, int64_t m, int64_t k, const int64_t* rowptr, const int64_t* colidx,          \
                                             const T* vals, int64_t* rowptrT, int64_t* colidxT, T* valsT) {                             \
        if (!c || m < 0 || k < 0) return -2;                                                                                            \
        return rlhip::csr_transpose<T>(c, m, k, rowptr, colidx, vals, rowptrT, colidxT, valsT);                                         \
    }
CAPI(double, f64)
CAPI(float, f32)
