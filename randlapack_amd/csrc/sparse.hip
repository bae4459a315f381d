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

template <typename T>
int csr_spmm_rowmajor(rlhip_ctx* c, int64_t nrows, int64_t nc, const int64_t* rowptr, const int64_t* colidx, const T* vals, T alpha,
                      const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {
    if (nrows <= 0 || nc <= 0) return 0;
    const int64_t gx = std::min<int64_t>((nrows + 3) / 4, 256 * 64);
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

// CSR of the transpose (k x m) of an m x k CSR matrix.  Construction-time, not on the timed path: staged through the host with a
// stable counting sort so that the entry order inside every transposed row (= ascending source row) is deterministic.
template <typename T>
int csr_transpose(rlhip_ctx* c, int64_t m, int64_t k, const int64_t* rowptr, const int64_t* colidx, const T* vals, int64_t* rowptrT,
                  int64_t* colidxT, T* valsT) {
    int64_t* h_rp = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m + 1));
    if (!h_rp) return -3;
    RLHIP_CHECK(hipMemcpyAsync(h_rp, rowptr, sizeof(int64_t) * (size_t)(m + 1), hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(hipStreamSynchronize(c->stream));
    const int64_t nnz = h_rp[m];
    int64_t* h_ci = (int64_t*)malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(nnz, 1));
    T* h_v = (T*)malloc(sizeof(T) * (size_t)std::max<int64_t>(nnz, 1));
    int64_t* h_rpt = (int64_t*)calloc((size_t)(k + 2), sizeof(int64_t));
    int64_t* h_cit = (int64_t*)malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(nnz, 1));
    T* h_vt = (T*)malloc(sizeof(T) * (size_t)std::max<int64_t>(nnz, 1));
    int rc = 0;
    if (!h_ci || !h_v || !h_rpt || !h_cit || !h_vt) rc = -3;
    if (!rc && nnz > 0) {
        if (hipMemcpyAsync(h_ci, colidx, sizeof(int64_t) * (size_t)nnz, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(h_v, vals, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = -1;
    }
    if (!rc) {
        for (int64_t p = 0; p < nnz; ++p) {
            if (h_ci[p] < 0 || h_ci[p] >= k) { rc = -2; break; }
            h_rpt[h_ci[p] + 2]++;
        }
    }
    if (!rc) {
        for (int64_t j = 0; j < k; ++j) h_rpt[j + 2] += h_rpt[j + 1];   // h_rpt[j+1] = start of row j while filling
        for (int64_t i = 0; i < m; ++i)
            for (int64_t p = h_rp[i]; p < h_rp[i + 1]; ++p) {
                const int64_t dst = h_rpt[h_ci[p] + 1]++;
                h_cit[dst] = i;
                h_vt[dst] = h_v[p];
            }
        if (hipMemcpyAsync(rowptrT, h_rpt, sizeof(int64_t) * (size_t)(k + 1), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = -1;
        if (!rc && nnz > 0 &&
            (hipMemcpyAsync(colidxT, h_cit, sizeof(int64_t) * (size_t)nnz, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
             hipMemcpyAsync(valsT, h_vt, sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice, c->stream) != hipSuccess))
            rc = -1;
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = -1;
    }
    free(h_rp); free(h_ci); free(h_v); free(h_rpt); free(h_cit); free(h_vt);
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
