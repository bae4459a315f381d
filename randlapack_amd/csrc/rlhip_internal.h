// Internal (non-ABI) declarations shared by the HIP translation units of librlhip.so.
// Everything here is MI355X / gfx950 only: 64-lane wavefronts, MFMA, 160 KiB LDS.
#pragma once
#include <mutex>
#include <utility>
#include <vector>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include "rlhip.h"

#define RLHIP_ERR_HIP(e) (-1000 - (int)(e))

#define RLHIP_CHECK(expr)                                                     \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) {                                               \
            fprintf(stderr, "[rlhip] %s:%d: %s -> %s\n", __FILE__, __LINE__,  \
                    #expr, hipGetErrorString(_e));                            \
            return RLHIP_ERR_HIP(_e);                                         \
        }                                                                     \
    } while (0)

#define RLHIP_LAUNCH_CHECK() RLHIP_CHECK(hipGetLastError())

// Raise a kernel's dynamic-LDS limit once PER DEVICE (kernel function attributes are per device: a process-wide flag would leave
// the second device of a process at the 64 KiB default).  `func` must be the kernel's address; the flag array lives at the call site.
#define RLHIP_FUNC_LDS(c, func, bytes)                                                                                     \
    do {                                                                                                                   \
        static bool _rlhip_lds_done[64] = {};                                                                              \
        const int _d = (c)->device & 63;                                                                                   \
        if (!_rlhip_lds_done[_d]) {                                                                                        \
            RLHIP_CHECK(hipSetDevice((c)->device));                                                                        \
            RLHIP_CHECK(hipFuncSetAttribute((const void*)(func), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            _rlhip_lds_done[_d] = true;                                                                                    \
        }                                                                                                                  \
    } while (0)

// The same for call sites where the kernel is a RUN-TIME value (a generic lambda over several instantiations shares one static per
// function-pointer TYPE, so the macro above would raise the limit of the first variant only): one flag per (device, kernel address).
#define RLHIP_FUNC_LDS_DYN(c, func, bytes)                                                                                 \
    do {                                                                                                                   \
        static std::mutex _rlhip_lds_mu;                                                                                   \
        static std::vector<std::pair<int, const void*>> _rlhip_lds_seen;                                                   \
        std::lock_guard<std::mutex> _rlhip_lds_lock(_rlhip_lds_mu);                                                        \
        const std::pair<int, const void*> _key((c)->device, (const void*)(func));                                          \
        bool _found = false;                                                                                               \
        for (auto const& _e : _rlhip_lds_seen) _found = _found || (_e == _key);                                            \
        if (!_found) {                                                                                                     \
            RLHIP_CHECK(hipSetDevice((c)->device));                                                                        \
            RLHIP_CHECK(hipFuncSetAttribute(_key.second, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));       \
            _rlhip_lds_seen.push_back(_key);                                                                               \
        }                                                                                                                  \
    } while (0)

// Execution context: one HIP stream + a growable device scratch arena + a small
// pinned host mailbox for info codes / scalars coming back from the device.
struct rlhip_ctx {
    int device = 0;
    int num_cu = 256;            // compute units of THIS context's device (persistent kernels size their grids with it)
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // scratch arena: stack-disciplined bump allocator over a short list of device segments.  Marks are virtual
    // offsets (segment k starts where segment k-1's full size ends); when a request does not fit, a new, larger
    // segment is appended (hipMalloc once); when the stack returns to empty the segments are merged into one so
    // that steady-state calls never allocate.
    struct Seg { char* base; size_t size; };
    Seg segs[32];
    int nsegs = 0;
    int cur_seg = 0;          // segment currently bumped
    size_t cur_used = 0;      // bytes used inside segs[cur_seg]
    size_t ws_highwater = 0;  // largest virtual offset ever reached
    // pinned mailbox
    int64_t* h_mail = nullptr;   // 64 x int64 host-pinned
    int64_t* d_mail = nullptr;   // 64 x int64 device
    void* xchg = nullptr;        // exchange words of the persistent panel kernels (see rlhip_xchg_buffer)
    size_t xchg_bytes = 0;
    void* xloc = nullptr;        // ordinary (cached) twin of the exchange buffer: same-XCD hand-overs of the persistent Jacobi launch (rlhip_xloc_buffer)
    size_t xloc_bytes = 0;
    // timing of the most recent GEMM-family launch set (bench.py roofline leg)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_flag = nullptr;   // marks a flag read-back INSIDE a stream of launches: the host waits for the flags only, the device runs on (tri.hip)
    // ||A||_F fused into a product and not yet collected (rlhip_gemm_norma_f64 with a null result pointer): 0 nothing, 1 the sum of squares
    // is on its way to h_mail[40] behind the stream, 2 norma_value holds the norm
    int norma_state = 0;
    double norma_value = 0;
    int norma_reduced = 0;           // 1: the sum over the row shards is on its way to h_mail[41] as well (rode on the Gram matrix's all-reduce, tri.hip::cholqrq)
    unsigned long norma_epoch = 0;   // sync_epoch when the deferred copy was enqueued
    unsigned long sync_epoch = 0;    // completed host waits on the stream (rlhip_stream_sync): anything enqueued before the last one has landed
    // != 0: products take the tiled kernel whose workgroups come and go, not the persistent stream-K kernel that holds every CU for its whole
    // duration (set around a product that is meant to share the device with another stream: house.hip::gemqrt_lt_tail)
    int avoid_persistent = 0;
    // > 0: columns per workgroup of the tag-exchange pivoted QR (default 4; a caller that overlaps the factorization with another kernel packs
    // the columns into fewer workgroups so that it occupies fewer CUs)
    int qrcp_cols_per_wg = 0;
    int64_t opt[RLHIP_OPT_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // rlhip_set_option; -1 = default
    rlhip_ctx* side_ctx = nullptr;   // cached side context (rlhip_side_of): created on first use, destroyed with this one
    hipStream_t side = nullptr;  // second stream, created on first use (rlhip_dvfs_burn: load beside the main stream's latency-bound kernels)
    // row-sharding communicator (comm.hip), nullptr = single GPU
    void* comm = nullptr;
    // caching pool behind rlhip_malloc/rlhip_free (outputs the drivers allocate for the caller: Q, BT, U, S, V).
    // Blocks are recycled by exact size; reuse is ordered by the context's stream, so freeing does not synchronise.
    struct PoolBlk { void* p; size_t bytes; bool in_use; unsigned long stamp; };
    PoolBlk pool[256];
    int npool = 0;
    size_t pool_idle_bytes = 0, pool_cap_bytes = 0;
    unsigned long pool_clock = 0;
    // diagnostics: how often each specialised kernel path was taken (rlhip_path_count; tests assert the path under test ran)
    //   0 stream-K f64 GEMM, 1 stream-K f32 GEMM, 2 fused trsm block kernel, 3 substitution trsm sub-block, 4 fused out-of-place trsm
    //   (rlhip_trsm_gather), 5 sketch-preconditioned Cholesky-QR panel inside geqrf -- the list in include/rlhip.h is the contract
    int64_t path_count[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

// every host wait of the library goes through here: the epoch lets deferred read-backs know that an earlier wait already covered them
static inline hipError_t rlhip_stream_sync(rlhip_ctx* c) {
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) ++c->sync_epoch;
    return e;
}

// scratch arena helpers (capi.hip)
void* rlhip_ws_alloc(rlhip_ctx* c, size_t bytes);           // 256-B aligned, never fails softly (nullptr on OOM)
size_t rlhip_ws_mark(rlhip_ctx* c);
void rlhip_ws_release(rlhip_ctx* c, size_t mark);

template <typename T>
static inline T* ws_alloc(rlhip_ctx* c, size_t n) { return (T*)rlhip_ws_alloc(c, n * sizeof(T)); }
// device buffer for the tagged-word exchanges between workgroups (QRCP / LU panel kernels); grows, lives with the context.
// RLHIP_XCHG = 0: ordinary device memory, 1: fine-grained, 2: uncached (default)
void* rlhip_xchg_buffer(rlhip_ctx* c, size_t bytes);
void* rlhip_xloc_buffer(rlhip_ctx* c, size_t bytes);

// ---- typed internal entry points (implemented in the .hip files; the extern "C" ABI wraps them) ----
namespace rlhip {

enum Op : int { NoTrans = 0, Trans = 1 };
enum Uplo : int { Upper = 0, Lower = 1 };
enum Diag : int { NonUnit = 0, Unit = 1 };
enum Dist : int { Gaussian = 0, UniformPM1 = 1 };

template <typename T>
int gemm(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A,
         int64_t lda, const T* B, int64_t ldb, T beta, T* C, int64_t ldc);

// C(upper or lower) = alpha * op(A)^T-style Gram + beta*C, LAPACK syrk semantics (only `uplo` part referenced/written)
template <typename T>
int syrk(rlhip_ctx* c, int uplo, int trans, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
         T beta, T* C, int64_t ldc);

template <typename T>
int potrf_upper(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int* info_host);

// B <- alpha * B * inv(op(A)),  A upper triangular n x n (Side::Right, Uplo::Upper, NoTrans)
template <typename T>
int trsm_right_upper_oop(rlhip_ctx* c, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, const T* Bsrc, int64_t ldsrc,
                         const int64_t* perm_dev, T* B, int64_t ldb);
template <typename T>
int trsm_right_upper_oop_range(rlhip_ctx* c, int diag, int64_t m, int64_t nsrc, T alpha, const T* A, int64_t lda, const T* Bsrc, int64_t ldsrc,
                               const int64_t* perm_dev, T* B, int64_t ldb, int64_t col0, int64_t col1);
template <typename T>
int trsm_right_upper(rlhip_ctx* c, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda,
                     T* B, int64_t ldb);

// B <- alpha * B * A,  A upper triangular n x n (Side::Right, Uplo::Upper, NoTrans); B is m x n
template <typename T>
int trmm_right_upper(rlhip_ctx* c, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda,
                     T* B, int64_t ldb);

template <typename T>
int trmm_left_upper(rlhip_ctx* c, int trans, int diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, T* B, int64_t ldb);

template <typename T>
int fill_dense(rlhip_ctx* c, int dist, int64_t rows, int64_t cols, T* buf, const uint32_t ctr[4],
               const uint32_t key[2], uint32_t next_ctr[4]);

template <typename T>
int lange_fro(rlhip_ctx* c, int64_t m, int64_t n, const T* A, int64_t lda, T* result_host);

template <typename T>
int lacpy(rlhip_ctx* c, int uplo /*0 upper,1 lower,2 general*/, int64_t m, int64_t n, const T* A,
          int64_t lda, T* B, int64_t ldb);

template <typename T>
int laset(rlhip_ctx* c, int uplo /*0 upper,1 lower,2 general*/, int64_t m, int64_t n, T offdiag, T diag,
          T* A, int64_t lda);

// one-sided Jacobi SVD of a tall m x n (m >= n) matrix: A = U diag(S) VT.
// On exit A holds U (m x n), S descending, VT n x n (ld ldvt).
template <typename T>
int gesvdj(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* S, T* VT, int64_t ldvt,
           int* sweeps_host);


// A[i,i] += alpha
template <typename T>
int add_diag(rlhip_ctx* c, int64_t n, T alpha, T* A, int64_t lda);

// AT (n x m, ld ldat) = A^T (A m x n); upper_only: only entries i <= j of A are moved
template <typename T>
int transpose(rlhip_ctx* c, int64_t m, int64_t n, const T* A, int64_t lda, T* AT, int64_t ldat, int upper_only);

// thin SVD of a tall matrix with separate outputs (LAPACK gesdd 'S' contract); A is destroyed
template <typename T>
int gesdd_tall(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* S, T* U, int64_t ldu, T* VT,
               int64_t ldvt, int* sweeps_host);

}  // namespace rlhip
