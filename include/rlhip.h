/* rlhip.h -- C ABI of librlhip.so: the MI355X (gfx950) device layer underneath the RandLAPACK
 * sketch-and-factor path.
 *
 * This is the drop-in boundary.  The reference has no FFI on this path: its
 * drivers are C++ templates that bottom out in blas:: / lapack:: / RandBLAS:: free functions
 * (RandLAPACK/rl_blaspp.hh:5-9, RandLAPACK/rl_lapackpp.hh:7-9).  Each entry point below names the
 * reference call site(s) it replaces.  The C++ classes in include/RandLAPACK_amd/ (same names and call
 * signatures as the reference's driver/comp objects) are written purely against this header.
 *
 * Conventions
 *   - every matrix pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - column-major storage, int64_t dimensions and leading dimensions, 1-based int64_t pivots;
 *   - op / uplo / diag / side flags are the LAPACK characters ('N','T','U','L','R', ...);
 *   - return value: 0 success; >0 LAPACK-style info (e.g. potrf failing minor); <0 = -(index of the bad
 *     argument) or -1000-hipError_t for a runtime failure.  Nothing aborts the process.
 *   - suffix _f64 / _f32 = the arithmetic type; results of the f64 entry points are what the parity
 *     tests compare against the CPU oracle.
 *   - all work is enqueued on the context's HIP stream; entry points that return a value to the host
 *     (info codes, norms, ranks) synchronise that stream, the others are asynchronous.
 */
#ifndef RLHIP_H
#define RLHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rlhip_ctx rlhip_ctx;

/* ---- context, memory, stream (replaces blas::Queue / lapack::Queue, device_malloc, device_free,
 *      device_copy_matrix: RandLAPACK/drivers/rl_bqrrp_gpu.hh:232-233, rl_cqrrpt_gpu.hh:267-297) ---- */
/* own_stream != 0: create a private non-blocking stream (hip_stream ignored);
 * own_stream == 0: enqueue on hip_stream (a hipStream_t; NULL = the device's default stream, which is what
 * PyTorch-ROCm uses unless told otherwise). */
int rlhip_create(rlhip_ctx** ctx, int device, void* hip_stream, int own_stream);
/* a second context on the parent's device: own high-priority stream, own scratch arena -- for work that runs beside the parent's stream
 * (BQRRP's look-ahead); destroyed with rlhip_destroy.  rlhip_order_after: what `waiter` enqueues from now on starts after what `signaler`
 * has enqueued so far (an event, no host wait). */
int rlhip_create_side(rlhip_ctx* parent, rlhip_ctx** out);
int rlhip_order_after(rlhip_ctx* waiter, rlhip_ctx* signaler);
int rlhip_destroy(rlhip_ctx* ctx);
/* ---- per-context options: the library's supported switches (value -1 = the default; a side context inherits its parent's at creation).
 *      Everything else about kernel selection is decided from the shapes alone -- there are no environment switches for kernels; the
 *      environment only carries diagnostics (RLHIP_SK_CLOCK, RLHIP_JACOBI_CLOCK, RLHIP_SK_TUNE, RLHIP_GESDD_TRACE, RLHIP_POOL_TRACE) and
 *      RLHIP_COMM_SINGLE_RANK_NCCL (build a real one-rank RCCL communicator).
 *      The first group selects between two routes of this library that deliver the same result (tests compare them, bitwise where
 *      stated); the RLHIP_OPT_DRV_* group holds defaults that the rlhip_drv_* entry points (rlhip_drivers.h) hand to the C++ objects
 *      they construct -- C++ callers set the members of the same name on their own objects. */
enum rlhip_option {
    RLHIP_OPT_CHOLQRQ_ONE_STREAM = 0,   /* 1: rlhip_cholqrq_* serves tall 256-aligned inputs as one stream of kernels; 0: returns 1 (caller runs syrk, potrf, trsm) */
    RLHIP_OPT_GESDD_GRAM = 1,           /* 1: device SVD of a well-conditioned tall factor, 32 < k <= 256, by Jacobi on its Gram matrix; 0: classic route always */
    RLHIP_OPT_JACOBI_PERSIST = 2,       /* 1: all Jacobi sweeps in one cooperative launch (blocks handed over through the XCD's L2 when the workers share one); 2: the same, hand-over through uncached memory always; 3: the workers alone through an ordinary (non-cooperative) launch -- the route rocprofv3 --pmc can profile, idle device only; 0: one launch per round */
    RLHIP_OPT_TRSM_XASM = 3,            /* fused solve: X loads issued from inline asm (1) or plain loads (0); default = what scripts/check_trsm_asm.py proved for this build */
    RLHIP_OPT_SASO_MODE = 4,            /* rlhip_saso_create: 1 independent columns (RandBLAS's short-axis SASO), 0 block-affine family */
    RLHIP_OPT_HQRRP_TALL_PANEL = 5,     /* hqrrp: 1 pivots of a tall panel from the QRCP of its R factor; 0: one pivoted sweep (the reference's order) */
    RLHIP_OPT_DRV_BQRRP_LOOKAHEAD_MIN_ELEMS = 6,   /* BQRRP::lookahead_min_elems (default 2.5e8; 0 = whenever the shapes allow, a huge value = never) */
    RLHIP_OPT_DRV_BQRRP_CHOLQR_FALLBACK = 7,       /* BQRRP::cholqr_fallback (default 1; 0 = the reference's behaviour on a Cholesky breakdown) */
    RLHIP_OPT_DRV_CQRRPT_FOLD_PIVOTING = 8,        /* CQRRPT::fold_pivoting (default 1) */
    RLHIP_OPT_DRV_CQRRPT_SPLIT_QRCP = 9,           /* CQRRPT::split_qrcp (default 1) */
    RLHIP_OPT_DRV_SPARSE_SKETCH_DENSIFY = 10,      /* linops::SparseLinOp::force_densified_sketch (default 0) */
    RLHIP_OPT_JACOBI_CLOCK_HOLDERS = 11,           /* persistent Jacobi: 1 (default) the CUs its workers leave idle run FMA-burning holder workgroups so that DVFS keeps the clock up for the next GEMM (never on a context that shares the device with another stream); 0: the workers alone */
    RLHIP_OPT_COUNT = 12
};
int rlhip_set_option(rlhip_ctx* ctx, int option, int64_t value);     /* -1 (bad context / option) or 0 */
int64_t rlhip_get_option(rlhip_ctx* ctx, int option);                /* the stored value (-1 = default), INT64_MIN for a bad option */
int rlhip_sync(rlhip_ctx* ctx);
void* rlhip_stream(rlhip_ctx* ctx);
int rlhip_malloc(rlhip_ctx* ctx, void** dev_ptr, size_t bytes);
int rlhip_free(rlhip_ctx* ctx, void* dev_ptr);
/* rlhip_malloc/rlhip_free recycle blocks by exact size (stream-ordered, no synchronisation on free; at most 1/8 of
 * the device memory idles in the pool).  rlhip_trim returns every idle block to the driver. */
int rlhip_trim(rlhip_ctx* ctx);
/* page-locked HOST memory that the device kernels can address directly (slowly, over the host link): lets code written for the
 * reference's host buffers -- std::fill on an array that is then handed to a driver, test/comps/test_qb.cc:154 -- run unchanged
 * against this library for small problems.  Production data belongs in HBM (rlhip_malloc). */
int rlhip_malloc_host(rlhip_ctx* ctx, void** host_ptr, size_t bytes);
int rlhip_free_host(rlhip_ctx* ctx, void* host_ptr);
int rlhip_memcpy_h2d(rlhip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int rlhip_memcpy_d2h(rlhip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int rlhip_memcpy_d2d(rlhip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int rlhip_memset(rlhip_ctx* ctx, void* dst_dev, int byte, size_t bytes);
/* reserve the scratch arena up front (split-K slabs, trsm panels ...) so no hipMalloc happens later */
int rlhip_reserve_workspace(rlhip_ctx* ctx, size_t bytes);
size_t rlhip_workspace_highwater(rlhip_ctx* ctx);
const char* rlhip_version(void);

/* stream-ordered scratch arena for driver temporaries (Omega, Gram matrices, split-K slabs ...): a bump
 * allocator that replaces the reference's new[]/delete[] of work buffers inside call() (e.g. rl_rs.hh:130,
 * rl_rf.hh:116, rl_orth.hh:76).  Take a mark, allocate, release back to the mark; memory handed out is
 * only valid for work enqueued on the context's stream before the release. */
size_t rlhip_scratch_mark(rlhip_ctx* ctx);
int rlhip_scratch_alloc(rlhip_ctx* ctx, void** dev_ptr, size_t bytes);
int rlhip_scratch_release(rlhip_ctx* ctx, size_t mark);

/* event timing on the context's stream (bench.py roofline leg) */
int rlhip_timer_start(rlhip_ctx* ctx);
int rlhip_timer_stop_ms(rlhip_ctx* ctx, float* ms_host);

/* ---- counter-based RNG: RandBLAS::RNGState + fill_dense (rl_rs.hh:134-139, rl_bqrrp.hh:310-311,
 *      rl_hqrrp.hh:929-930, rl_abrik.hh:298-299).  dist: 0 = N(0,1), 1 = U(-1,1).
 *      ctr[4]/key[2] are host arrays; next_ctr_host receives the advanced counter (may be NULL). ---- */
int rlhip_philox4x32_10(rlhip_ctx* ctx, int64_t nblocks, uint32_t* out_dev, const uint32_t ctr_host[4],
                        const uint32_t key_host[2]);
int rlhip_fill_dense_f64(rlhip_ctx* ctx, int dist, int64_t rows, int64_t cols, double* buf,
                         const uint32_t ctr_host[4], const uint32_t key_host[2], uint32_t next_ctr_host[4]);
int rlhip_fill_dense_f32(rlhip_ctx* ctx, int dist, int64_t rows, int64_t cols, float* buf,
                         const uint32_t ctr_host[4], const uint32_t key_host[2], uint32_t next_ctr_host[4]);
/* rows [row0, row0 + loc_rows) of the glob_rows x cols operator rlhip_fill_dense_* generates (same stream positions):
 * a row shard regenerates its slice of the global m x k sketch (odd power-pass counts, rl_rs.hh:137-139, under
 * row-block sharding); the device counterpart of RandBLAS::fill_dense_unpacked's offsets.  next_ctr advances by the
 * GLOBAL size, so every rank's state stays identical to the single-device run. */
int rlhip_fill_dense_rows_f64(rlhip_ctx* ctx, int dist, int64_t glob_rows, int64_t cols, int64_t row0, int64_t loc_rows,
                              double* buf, int64_t ld, const uint32_t ctr[4], const uint32_t key[2], uint32_t next_ctr[4]);
int rlhip_fill_dense_rows_f32(rlhip_ctx* ctx, int dist, int64_t glob_rows, int64_t cols, int64_t row0, int64_t loc_rows,
                              float* buf, int64_t ld, const uint32_t ctr[4], const uint32_t key[2], uint32_t next_ctr[4]);

/* ---- BLAS-3 on MFMA.  blas::gemm (rl_rs.hh:142,153,165; rl_rf.hh:123; rl_qb.hh:210-218,260;
 *      rl_rsvd.hh:148), blas::syrk (rl_orth.hh:78; rl_cqrrpt.hh:310; rl_bqrrp.hh:460),
 *      blas::trsm Side::Right/Uplo::Upper/NoTrans (rl_orth.hh:95; rl_cqrrpt.hh:302,338; rl_bqrrp.hh:457,464),
 *      blas::trmm Side::Right/Uplo::Upper/NoTrans (rl_cqrrpt.hh:345; rl_bqrrp.hh:497). ---- */
int rlhip_gemm_f64(rlhip_ctx* ctx, char transa, char transb, int64_t m, int64_t n, int64_t k, double alpha,
                   const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                   int64_t ldc);
int rlhip_gemm_f32(rlhip_ctx* ctx, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                   const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                   int64_t ldc);
/* gemm + ||A||_F in ONE pass over A (QB needs both: rl_qb.hh:168 then rl_rf.hh:123 read A twice in the
 * reference).  A is (m x k) for transa 'N', (k x m) for 'T'.  *fused_host = 1 when the norm came out of the GEMM
 * kernel itself (stream-K path), 0 when a separate lange pass was needed.  Synchronises the stream -- unless norm_a_host is NULL:
 * the norm is then DEFERRED (when it came out of the product, its sum of squares follows the stream into a pinned mailbox) and
 * rlhip_norma_collect_f64 returns it later, after the caller's next synchronisation (rl_qb.hh:168,221: ||A||_F and ||B_i||_F with one
 * host round trip).  One norm can be pending per context. */
int rlhip_gemm_norma_f64(rlhip_ctx* ctx, char transa, char transb, int64_t m, int64_t n, int64_t k, double alpha,
                         const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                         int64_t ldc, double* norm_a_host, int* fused_host);
/* the norm a rlhip_gemm_norma_f64 call with a NULL result pointer left pending (-4 when there is none).  over_ranks != 0 on a row-sharded
 * context: the norm of the WHOLE matrix (root of the sum over the ranks' sums of squares) -- the sum rides on the Gram matrix's all-reduce
 * when a rlhip_cholqrq call came in between (the QB sequence), otherwise it takes a scalar all-reduce of its own here. */
int rlhip_norma_collect_f64(rlhip_ctx* ctx, int over_ranks, double* norm_a_host);
/* uplo must be 'U' (the only form the path uses); the strictly lower triangle of C is not touched. */
int rlhip_syrk_f64(rlhip_ctx* ctx, char uplo, char trans, int64_t n, int64_t k, double alpha, const double* A,
                   int64_t lda, double beta, double* C, int64_t ldc);
int rlhip_syrk_f32(rlhip_ctx* ctx, char uplo, char trans, int64_t n, int64_t k, float alpha, const float* A,
                   int64_t lda, float beta, float* C, int64_t ldc);
/* B <- alpha * B * inv(A), A n x n upper triangular, B m x n (side 'R', uplo 'U', trans 'N' only) */
int rlhip_trsm_f64(rlhip_ctx* ctx, char side, char uplo, char trans, char diag, int64_t m, int64_t n,
                   double alpha, const double* A, int64_t lda, double* B, int64_t ldb);
int rlhip_trsm_f32(rlhip_ctx* ctx, char side, char uplo, char trans, char diag, int64_t m, int64_t n,
                   float alpha, const float* A, int64_t lda, float* B, int64_t ldb);
/* Out-of-place right-side solve with a column gather:  B <- alpha * (Bsrc * P) * inv(A), where column c of Bsrc * P is column
 * jpvt_dev[c] - 1 of Bsrc (LAPACK-style 1-based pivot vector in device memory; NULL: P = I).  Bsrc (m x n, ldsrc) is not modified and
 * must not overlap B unless it IS B with jpvt_dev == NULL (plain rlhip_trsm).  This is CQRRPT's "util::col_swap(A, J) followed by
 * blas::trsm(A, R_sk)" (rl_cqrrpt.hh:288-300) as ONE pass over A, and its second solve (rl_cqrrpt.hh:329) written straight into A.
 * Returns -7 and writes nothing when jpvt_dev is not a permutation of 1..n (the same refusal as rlhip_col_swap). */
int rlhip_trsm_gather_f64(rlhip_ctx* ctx, char diag, int64_t m, int64_t n, double alpha, const double* A, int64_t lda,
                          const double* Bsrc, int64_t ldsrc, const int64_t* jpvt_dev, double* B, int64_t ldb);
int rlhip_trsm_gather_f32(rlhip_ctx* ctx, char diag, int64_t m, int64_t n, float alpha, const float* A, int64_t lda,
                          const float* Bsrc, int64_t ldsrc, const int64_t* jpvt_dev, float* B, int64_t ldb);
/* Columns [col0, col1) (multiples of 256) of the same solve, given that B[:, 0 : col0) already holds the solution's leading columns; only the
 * leading col1 x col1 part of A and jpvt[0 : col1) are read, and jpvt maps into nsrc >= col1 source columns (a prefix of a pivot vector).
 * Pieces of a split solve are bitwise the whole solve.  Returns 0, 1 (not taken: outside the fused kernel's domain, nothing written --
 * the caller takes rlhip_trsm_gather_* for the whole matrix) or an error (< 0). */
int rlhip_trsm_gather_range_f64(rlhip_ctx* ctx, char diag, int64_t m, int64_t nsrc, double alpha, const double* A, int64_t lda,
                                const double* Bsrc, int64_t ldsrc, const int64_t* jpvt, double* B, int64_t ldb, int64_t col0, int64_t col1);
int rlhip_trsm_gather_range_f32(rlhip_ctx* ctx, char diag, int64_t m, int64_t nsrc, float alpha, const float* A, int64_t lda,
                                const float* Bsrc, int64_t ldsrc, const int64_t* jpvt, float* B, int64_t ldb, int64_t col0, int64_t col1);
/* side 'R': B <- alpha * B * A, A n x n upper triangular, B m x n (trans 'N' only);
 * side 'L': B <- alpha * op(A) * B, A m x m upper triangular (trans 'N' or 'T').  uplo 'U' only. */
int rlhip_trmm_f64(rlhip_ctx* ctx, char side, char uplo, char trans, char diag, int64_t m, int64_t n,
                   double alpha, const double* A, int64_t lda, double* B, int64_t ldb);
int rlhip_trmm_f32(rlhip_ctx* ctx, char side, char uplo, char trans, char diag, int64_t m, int64_t n,
                   float alpha, const float* A, int64_t lda, float* B, int64_t ldb);

/* ---- LAPACK-level pieces.  lapack::potrf Upper (rl_orth.hh:81; rl_cqrrpt.hh:311; rl_bqrrp.hh:462):
 *      returns info (0, or the 1-based order of the first non-positive leading minor). ---- */
int rlhip_potrf_f64(rlhip_ctx* ctx, char uplo, int64_t n, double* A, int64_t lda);
int rlhip_potrf_f32(rlhip_ctx* ctx, char uplo, int64_t n, float* A, int64_t lda);
/* CholQRQ::call (rl_orth.hh:69-98: syrk :78, potrf :81, trsm :95) as ONE stream of kernels with ONE host read: R (k x k, ld k) receives the
 * Cholesky factor of A^T A, A (m x k) is overwritten by A R^-1, *info_host = potrf's info (!= 0: A is untouched, as the reference returns
 * before its trsm).  reduce_gram != 0: row-sharded A, the Gram matrix is all-reduced before it is factored.  Returns 0 when done, 1 when
 * this shape is not served (the caller issues syrk / potrf / trsm itself), < 0 on error. */
int rlhip_cholqrq_f64(rlhip_ctx* ctx, int64_t m, int64_t k, double* A, int64_t lda, double* R, int reduce_gram, int* info_host);
int rlhip_cholqrq_f32(rlhip_ctx* ctx, int64_t m, int64_t k, float* A, int64_t lda, float* R, int reduce_gram, int* info_host);
/* lapack::lange(Norm::Fro) (rl_qb.hh:168,221) */
int rlhip_lange_fro_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t lda, double* result_host);
int rlhip_lange_fro_f32(rlhip_ctx* ctx, int64_t m, int64_t n, const float* A, int64_t lda, float* result_host);
/* lapack::lacpy / lapack::laset; uplo 'U','L' or 'G' (rl_qb.hh:171; rl_cqrrpt.hh:281; rl_util.hh:102-131) */
int rlhip_lacpy_f64(rlhip_ctx* ctx, char uplo, int64_t m, int64_t n, const double* A, int64_t lda, double* B,
                    int64_t ldb);
int rlhip_lacpy_f32(rlhip_ctx* ctx, char uplo, int64_t m, int64_t n, const float* A, int64_t lda, float* B,
                    int64_t ldb);
int rlhip_laset_f64(rlhip_ctx* ctx, char uplo, int64_t m, int64_t n, double offdiag, double diag, double* A,
                    int64_t lda);
int rlhip_laset_f32(rlhip_ctx* ctx, char uplo, int64_t m, int64_t n, float offdiag, float diag, float* A,
                    int64_t lda);
/* thin SVD of a tall m x n matrix (m >= n), replacing lapack::gesdd(Job::SomeVec) at rl_rsvd.hh:146:
 * on exit A holds U (m x n), S[n] descending, VT (n x n, ld ldvt).  One-sided Jacobi, entirely on device.
 * returns the number of sweeps used in *sweeps_host (may be NULL); info > 0 = not converged. */
int rlhip_gesvdj_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* S, double* VT,
                     int64_t ldvt, int* sweeps_host);
int rlhip_gesvdj_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* S, float* VT,
                     int64_t ldvt, int* sweeps_host);

/* A[i,i] += alpha for i < n (the "A_gram[i*k+i] -= 1" loop of util::orthogonality_check, rl_util.hh:480-482) */
int rlhip_add_diag_f64(rlhip_ctx* ctx, int64_t n, double alpha, double* A, int64_t lda);
int rlhip_add_diag_f32(rlhip_ctx* ctx, int64_t n, float alpha, float* A, int64_t lda);
/* lapack::gesdd(Job::SomeVec) contract for a tall matrix (m >= n): A (destroyed) = U diag(S) VT with
 * U m x n (ldu), VT n x n (ldvt), S descending.  Cholesky-QR2 preconditioning + Jacobi on R^T; falls back
 * to rlhip_gesvdj on A when the Cholesky steps cannot be trusted.  (rl_rsvd.hh:146) */
int rlhip_gesdd_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* S, double* U, int64_t ldu,
                    double* VT, int64_t ldvt, int* sweeps_host);
int rlhip_gesdd_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* S, float* U, int64_t ldu,
                    float* VT, int64_t ldvt, int* sweeps_host);
/* util::transposition (misc/rl_util.hh:315-334): AT (n x m, ld ldat) = A^T; upper_only != 0 moves only i <= j */
int rlhip_transpose_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t lda, double* AT,
                        int64_t ldat, int upper_only);
int rlhip_transpose_f32(rlhip_ctx* ctx, int64_t m, int64_t n, const float* A, int64_t lda, float* AT, int64_t ldat,
                        int upper_only);

/* ---- sparse sketching operator (SASO): RandBLAS::SparseDist(d, m, nnz) + SparseSkOp(DS, state) +
 *      sketch_general(ColMajor, NoTrans, NoTrans, d, n, m, alpha, S, 0, 0, A, lda, beta, B, ldb)
 *      (rl_cqrrpt.hh:214-222, rl_cqrrt.hh:174-182).  S is d x m with nnz nonzeros (+-1) per column.
 *      next_ctr_host receives `S.next_state`.  Two structures (sketch.hip header; restated in oracle/__init__.py::saso_dense):
 *      mode 1 = INDEPENDENT COLUMNS, every column draws its nnz distinct rows by its own Fisher-Yates walk (the distribution of
 *      RandBLAS::SparseSkOp) -- the default; mode 0 = BLOCK AFFINE (one affine row map per block of d columns: faster,
 *      every d x d block is a sum of nnz signed permutations).  rlhip_saso_create takes the default (environment RLHIP_SASO_MODE =
 *      affine switches it), rlhip_saso_create_mode the given one.  Limits: nnz <= min(d, 128); the dense-operand apply stages a
 *      d-row slab in LDS: d <= 20480 (fp64) / 40960 (fp32), -2 beyond. ---- */
typedef struct rlhip_saso rlhip_saso;
int rlhip_saso_create(rlhip_ctx* ctx, int64_t d, int64_t m, int nnz, const uint32_t ctr_host[4],
                      const uint32_t key_host[2], uint32_t next_ctr_host[4], rlhip_saso** S);
int rlhip_saso_create_mode(rlhip_ctx* ctx, int64_t d, int64_t m, int nnz, int mode, const uint32_t ctr_host[4],
                           const uint32_t key_host[2], uint32_t next_ctr_host[4], rlhip_saso** S);
int rlhip_saso_destroy(rlhip_ctx* ctx, rlhip_saso* S);
int rlhip_saso_apply_f64(rlhip_ctx* ctx, const rlhip_saso* S, int64_t n, double alpha, const double* A, int64_t lda,
                         double beta, double* B, int64_t ldb);
int rlhip_saso_apply_f32(rlhip_ctx* ctx, const rlhip_saso* S, int64_t n, float alpha, const float* A, int64_t lda,
                         float beta, float* B, int64_t ldb);
/* contribution of ONE ROW SHARD: B = alpha * S[:, row0 : row0+mloc] * A_loc (mloc x n, lda) + beta * B.  Summed over the
 * shards (all-reduce) this is S * A; S was created for the GLOBAL row count. */
int rlhip_saso_apply_rows_f64(rlhip_ctx* ctx, const rlhip_saso* S, int64_t n, double alpha, const double* A_loc, int64_t lda,
                              int64_t row0, int64_t mloc, double beta, double* B, int64_t ldb);
int rlhip_saso_apply_rows_f32(rlhip_ctx* ctx, const rlhip_saso* S, int64_t n, float alpha, const float* A_loc, int64_t lda,
                              int64_t row0, int64_t mloc, float beta, float* B, int64_t ldb);
/* B (d x n, ldb) = alpha * S * A + beta * B for a SPARSE A (m x n) given by the CSR of its transpose (n rows; colidxT = source
 * row).  Scatter with 64-bit fixed-point integer LDS atomics: bitwise reproducible (sketch.hip).  d <= 19200.
 * row0: global index of the operand's first row (0 unless the operator is a row shard of the matrix S was built for). */
int rlhip_saso_apply_csr_f64(rlhip_ctx* ctx, const rlhip_saso* S, int64_t n, double alpha, const int64_t* rowptrT, const int64_t* colidxT,
                             const double* valsT, double beta, double* B, int64_t ldb, int64_t row0);
int rlhip_saso_apply_csr_f32(rlhip_ctx* ctx, const rlhip_saso* S, int64_t n, float alpha, const int64_t* rowptrT, const int64_t* colidxT,
                             const float* valsT, float beta, float* B, int64_t ldb, int64_t row0);
int rlhip_saso_dense_f64(rlhip_ctx* ctx, const rlhip_saso* S, double* dense_d_by_m);   /* tests / debugging */
int rlhip_saso_dense_f32(rlhip_ctx* ctx, const rlhip_saso* S, float* dense_d_by_m);
/* util::col_swap (misc/rl_util.hh:151-164 == lapmt forward): on exit column i holds former column idx[i]-1.
 * idx: DEVICE array of n 1-based indices, left untouched.  k > n is an error (reference throws).  In place,
 * one read + one write of the matrix, lanes along rows. */
int rlhip_col_swap_f64(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, double* A, int64_t lda, const int64_t* idx);
int rlhip_col_swap_f32(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, float* A, int64_t lda, const int64_t* idx);
/* integer-vector overload (rl_util.hh:174-198): first k entries of A permuted by a permutation idx of 1..k */
int rlhip_col_swap_i64(rlhip_ctx* ctx, int64_t n, int64_t k, int64_t* A, const int64_t* idx);
/* lapack::geqp3 (rl_cqrrpt.hh:247): column-pivoted Householder QR, LAPACK output format (R above, reflectors
 * below the diagonal, tau, 1-based jpvt); all arrays on the device.  jpvt entries on entry are ignored. */
int rlhip_geqp3_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, int64_t* jpvt, double* tau);
int rlhip_geqp3_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, int64_t* jpvt, float* tau);
/* diag(A)[0..n) to the host (rank tests read R's diagonal: rl_cqrrpt.hh:267-272,319-331, rl_bqrrp.hh:421-427) */
int rlhip_get_diag_f64(rlhip_ctx* ctx, int64_t n, const double* A, int64_t lda, double* diag_host);
int rlhip_get_diag_f32(rlhip_ctx* ctx, int64_t n, const float* A, int64_t lda, float* diag_host);

/* lapack::getrf (rl_bqrrp.hh:343, rl_orth.hh:219): row-pivoted LU of an m x n (m up to ~1e5, tall) device matrix;
 * ipiv: DEVICE int64, 1-based, min(m,n) entries.  Returns LAPACK info (first exactly-zero pivot) or 0. */
int rlhip_getrf_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, int64_t* ipiv);
/* the same factorization when only the PIVOTS are wanted (BQRRP's LU-based qrcp_wide reads J_buffer_lu and discards the factors,
 * rl_bqrrp.hh:343-352): ipiv is identical; A is left as scratch (U is valid, L misses the interchanges of later panels). */
int rlhip_getrf_piv_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, int64_t* ipiv);
int rlhip_getrf_piv_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, int64_t* ipiv);
int rlhip_getrf_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, int64_t* ipiv);
/* lapack::geqrf (rl_orth.hh:157, rl_bqrrp.hh:356,515): Householder QR, reflectors below the diagonal, tau: DEVICE.
 * lapack::ungqr(m, n, k = n) (rl_orth.hh:162): overwrite the reflectors with the first n columns of Q (k must equal n).
 * lapack::laswp(n, A, lda, k1, k2, ipiv, 1) (rl_orth.hh:226): forward row interchanges, ipiv DEVICE int64 1-based. */
/* geqrf FOLLOWED BY ungqr(m, n, n) in one call, for the call sites that want the explicit orthonormal factor (rl_abrik.hh:333-342, :420-444,
 * :552-570; HQRQ, rl_orth.hh:157-162): A <- the first n columns of the Householder Q, R (n x n, ld ldr) <- the triangle geqrf leaves
 * (zero below the diagonal).  Tall well-conditioned panels only (Cholesky-QR twice + the sign vector of the Householder reconstruction,
 * csrc/house.hip::geqrf_q): returns 0 when done, 1 when NOT taken -- the caller then runs rlhip_geqrf + rlhip_ungqr on A, which is still its
 * input to rounding -- < 0 on error.  Same Q and R as the two calls up to rounding. */
int rlhip_geqrf_q_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* R, int64_t ldr);
int rlhip_geqrf_q_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* R, int64_t ldr);
int rlhip_geqrf_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* tau);
int rlhip_geqrf_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* tau);
int rlhip_ungqr_f64(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, double* A, int64_t lda, const double* tau);
int rlhip_ungqr_f32(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, float* A, int64_t lda, const float* tau);
int rlhip_laswp_f64(rlhip_ctx* ctx, int64_t n, double* A, int64_t lda, int64_t k1, int64_t k2, const int64_t* ipiv);
int rlhip_laswp_f32(rlhip_ctx* ctx, int64_t n, float* A, int64_t lda, int64_t k1, int64_t k2, const int64_t* ipiv);
/* J <- iota(1..cols); for i < min(sd, cols): swap(J[ipiv[i]-1], J[i])   (rl_bqrrp.hh:345-350; CUDA twin
 * LUQRCP_piv_process_gpu_global, rl_cuda_kernels.cuh:203-220).  Integer-exact. */
int rlhip_luqrcp_piv(rlhip_ctx* ctx, int64_t sd, int64_t cols, const int64_t* ipiv, int64_t* J);

/* ---- Householder reconstruction / block reflectors (BQRRP, HQRRP) ----
 * lapack::orhr_col(m, n, nb, A, lda, T, ldt, D) (rl_bqrrp.hh:480; util::rl_orhr_col rl_util.hh:339-379; CUDA twin
 * rl_cuda_kernels.cuh:772-803): A holds orthonormal columns on entry, the unit-lower-trapezoidal V on exit; T (nb x n)
 * the compact-WY factors; D (n) the sign vector. */
int rlhip_orhr_col_f64(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t nb, double* A, int64_t lda, double* T, int64_t ldt,
                       double* D);
int rlhip_orhr_col_f32(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t nb, float* A, int64_t lda, float* T, int64_t ldt,
                       float* D);
/* lapack::gemqrt(Side::Left, Op::Trans, m, n, k, nb, V, ldv, T, ldt, C, ldc) (rl_bqrrp.hh:543) = the block-Householder
 * apply; with rlhip_larft_* it also serves lapack::ormqr (rl_bqrrp.hh:545, cusolver_ormqr rl_bqrrp_gpu.hh:734). */
int rlhip_gemqrt_f64(rlhip_ctx* ctx, char side, char trans, int64_t m, int64_t n, int64_t k, int64_t nb, const double* V,
                     int64_t ldv, const double* T, int64_t ldt, double* C, int64_t ldc);
int rlhip_gemqrt_f32(rlhip_ctx* ctx, char side, char trans, int64_t m, int64_t n, int64_t k, int64_t nb, const float* V,
                     int64_t ldv, const float* T, int64_t ldt, float* C, int64_t ldc);
/* The left / transposed apply of ONE compact-WY block (k reflectors, nb >= k) in two calls, cut where the first k rows of C -- BQRRP's block
 * row R12 (rl_bqrrp.hh:547) -- are final: head computes W2 = T^T V^T C (k x n, ld k, the caller's buffer) and updates rows 0..k-1 of C;
 * tail updates rows k..m-1 (C2 -= V2 W2).  head + tail == rlhip_gemqrt_*('L', 'T', ..., nb = k) bit for bit. */
int rlhip_gemqrt_head_f64(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* V, int64_t ldv, const double* T, int64_t ldt, double* C, int64_t ldc, double* W2);
int rlhip_gemqrt_head_f32(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, const float* V, int64_t ldv, const float* T, int64_t ldt, float* C, int64_t ldc, float* W2);
int rlhip_gemqrt_tail_f64(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* V, int64_t ldv, const double* W2, double* C, int64_t ldc);
int rlhip_gemqrt_tail_f32(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t k, const float* V, int64_t ldv, const float* W2, float* C, int64_t ldc);
/* (side = 'R', trans = 'N', nb >= k): C (m x n) <- C (I - V T V^T), V n x k -- lapack::larfb(Right, NoTrans, Forward, Columnwise) as
 * used by HQRRP to carry the sketching matrix along (NoFLA_Apply_Q_WY_rnfc_blk_var4, rl_hqrrp.hh:178-206).
 * rlhip_qrp_partial_*: pivoted Householder QR of the first `steps` columns only, HQRRP's norm down-date form
 * (NoFLA_QRPmod_WY_unb_var4 with pivoting = 1, num_stages = steps; rl_hqrrp.hh:516-770); jpvt (device, n, 1-based on exit)
 * is the complete permutation produced by the swaps. */
int rlhip_qrp_partial_f64(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t steps, double* A, int64_t lda, int64_t* jpvt, double* tau);
int rlhip_qrp_partial_f32(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t steps, float* A, int64_t lda, int64_t* jpvt, float* tau);
/* The first `steps` steps of rlhip_geqp3_* itself (LAPACK's norm down-date; lapack::geqp3 as called by rl_cqrrpt.hh:247): rows 0 .. steps-1
 * of R final for all columns, the trailing block updated by every reflector so far, jpvt the permutation so far.  geqp3 of the trailing
 * (m - steps) x (n - steps) block, with its pivots applied to the columns of the finished rows and composed into jpvt, completes it. */
int rlhip_geqp3_steps_f64(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t steps, double* A, int64_t lda, int64_t* jpvt, double* tau);
int rlhip_geqp3_steps_f32(rlhip_ctx* ctx, int64_t m, int64_t n, int64_t steps, float* A, int64_t lda, int64_t* jpvt, float* tau);
/* the context's own cached side context (see rlhip_create_side): created on first use, owned by and destroyed with `parent` -- do NOT
 * rlhip_destroy it */
int rlhip_side_of(rlhip_ctx* parent, rlhip_ctx** out);
/* columns per workgroup of the tag-exchange pivoted QR on this context (0 = default): fewer, fuller workgroups for a factorization that runs
 * beside another stream's kernel */
int rlhip_set_qrcp_cols(rlhip_ctx* ctx, int cols);
/* rows [toff, toff + tcnt) of the unit-lower-triangular V1 held implicitly in Vtop (br x br), written explicitly (0 above, 1 on the
 * diagonal) into out (tcnt x br): a rank's own rows of a reflector block under row sharding (BQRRP). */
int rlhip_vrows_explicit_f64(rlhip_ctx* ctx, int64_t br, int64_t toff, int64_t tcnt, const double* Vtop, int64_t ldv, double* out, int64_t ldo);
int rlhip_vrows_explicit_f32(rlhip_ctx* ctx, int64_t br, int64_t toff, int64_t tcnt, const float* Vtop, int64_t ldv, float* out, int64_t ldo);
/* lapack::larft(Forward, Columnwise): T (k x k) from V (m x k) and tau (k) */
int rlhip_larft_f64(rlhip_ctx* ctx, int64_t m, int64_t k, const double* V, int64_t ldv, const double* tau, double* T, int64_t ldt);
int rlhip_larft_f32(rlhip_ctx* ctx, int64_t m, int64_t k, const float* V, int64_t ldv, const float* tau, float* T, int64_t ldt);
/* R(j,i) *= D(j), j <= i  (rl_bqrrp.hh:485-487; R_cholqr_signs_gpu rl_cuda_kernels.cuh:222-235) */
int rlhip_row_sign_f64(rlhip_ctx* ctx, int64_t n, double* R, int64_t ldr, const double* D);
int rlhip_row_sign_f32(rlhip_ctx* ctx, int64_t n, float* R, int64_t ldr, const float* D);
/* tau(i) = T(i % nb, i)  (rl_bqrrp.hh:490-491) */
int rlhip_tau_from_t_f64(rlhip_ctx* ctx, int64_t k, int64_t nb, const double* T, int64_t ldt, double* tau);
int rlhip_tau_from_t_f32(rlhip_ctx* ctx, int64_t k, int64_t nb, const float* T, int64_t ldt, float* tau);
/* *any_host = 1 iff some |x[i]| > thr, i < n: the zero-block test of rl_bqrrp.hh:373-379 with the CPU semantics (the
 * reference's CUDA all_of only honours its first block) */
int rlhip_any_abs_gt_f64(rlhip_ctx* ctx, int64_t n, const double* x, double thr, int* any_host);
int rlhip_any_abs_gt_f32(rlhip_ctx* ctx, int64_t n, const float* x, float thr, int* any_host);

/* ---- test-matrix generator pieces (RandLAPACK/testing/rl_gen.hh; host-side logic in include/RandLAPACK_amd/rl_gen.hh) ----
 * scal_cols: A[:, j] *= s[j] (s DEVICE, n entries) -- the S factor of gen_singvec (rl_gen.hh:79).
 * scal_rows_idx: A[idx[r], :] *= alpha for the cnt listed rows (idx DEVICE, each row once) -- gen_spiked_mat rl_gen.hh:286-291.
 * gen_kahan: the Kahan matrix of gen_kahan_mat (rl_gen.hh:408-434), m x n leading block. */
int rlhip_scal_cols_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, const double* s_dev);
int rlhip_scal_cols_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, const float* s_dev);
int rlhip_scal_rows_idx_f64(rlhip_ctx* ctx, int64_t cnt, const int64_t* idx_dev, int64_t n, double* A, int64_t lda, double alpha);
int rlhip_scal_rows_idx_f32(rlhip_ctx* ctx, int64_t cnt, const int64_t* idx_dev, int64_t n, float* A, int64_t lda, float alpha);
int rlhip_gen_kahan_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double theta, double perturb);
int rlhip_gen_kahan_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float theta, float perturb);

/* symmetrize: F (n x n, ldf) = the symmetric matrix whose `uplo` triangle is stored in A; the other triangle of A is not read
 * (linops::ExplicitSymLinOp, RandLAPACK/linops/rl_sym_linops.hh:55-98).  axpby: y = alpha * x + beta * y on n-vectors (beta == 0
 * overwrites without reading y) -- the scal / axpy / copy steps of REVD2's power_error_est (drivers/rl_revd2.hh:34-63). */
int rlhip_symmetrize_f64(rlhip_ctx* ctx, char uplo, int64_t n, const double* A, int64_t lda, double* F, int64_t ldf);
int rlhip_symmetrize_f32(rlhip_ctx* ctx, char uplo, int64_t n, const float* A, int64_t lda, float* F, int64_t ldf);
int rlhip_axpby_f64(rlhip_ctx* ctx, int64_t n, double alpha, const double* x, double beta, double* y);
int rlhip_axpby_f32(rlhip_ctx* ctx, int64_t n, float alpha, const float* x, float beta, float* y);

/* ---- sparse linear operator kernels (linops::SparseLinOp; reference RandLAPACK/linops/rl_sparse_linop.hh:125-330 forwards to
 *      RandBLAS left_spmm/right_spmm).  CSR with int64 indices, all arrays DEVICE pointers.
 *      csr_spmm: C (m x n) = alpha * A (m x k, CSR) * B (k x n) + beta * C; layout 'C' (column-major B, C) or 'R' (row-major).
 *      csr_transpose: CSR of A^T (construction time; staged through the host, deterministic entry order).
 *      csr_densify_cols: out (m x b, column-major) = A[:, c0 : c0 + b], read from the CSR of A^T. ---- */
int rlhip_csr_spmm_f64(rlhip_ctx* ctx, char layout, int64_t m, int64_t n, int64_t k, double alpha, const int64_t* rowptr,
                       const int64_t* colidx, const double* vals, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
int rlhip_csr_spmm_f32(rlhip_ctx* ctx, char layout, int64_t m, int64_t n, int64_t k, float alpha, const int64_t* rowptr,
                       const int64_t* colidx, const float* vals, const float* B, int64_t ldb, float beta, float* C, int64_t ldc);
int rlhip_csr_transpose_f64(rlhip_ctx* ctx, int64_t m, int64_t k, const int64_t* rowptr, const int64_t* colidx, const double* vals,
                            int64_t* rowptrT, int64_t* colidxT, double* valsT);
int rlhip_csr_transpose_f32(rlhip_ctx* ctx, int64_t m, int64_t k, const int64_t* rowptr, const int64_t* colidx, const float* vals,
                            int64_t* rowptrT, int64_t* colidxT, float* valsT);
int rlhip_csr_densify_cols_f64(rlhip_ctx* ctx, int64_t m, const int64_t* rowptrT, const int64_t* colidxT, const double* valsT,
                               int64_t c0, int64_t b, double* out, int64_t ldo);
int rlhip_csr_densify_cols_f32(rlhip_ctx* ctx, int64_t m, const int64_t* rowptrT, const int64_t* colidxT, const float* valsT,
                               int64_t c0, int64_t b, float* out, int64_t ldo);

/* ---- row-block sharding across the GPUs of a node (new design; the reference has no
 *      distributed code).  One process per GPU.  Sum all-reduces run on the context's stream through RCCL
 *      (bound at run time), or through a host-installed hook.  With no communicator every call is a no-op,
 *      so single-GPU callers never notice. ---- */
typedef int (*rlhip_allreduce_hook)(void* user, void* dev_buf, int64_t count, int is_f64);
/* 1 when RCCL can be bound in this process, 0 otherwise.  ncclCommInitRank is collective, so ranks agree on this (MIN over the
 * ranks) BEFORE any of them calls rlhip_comm_init; a process that already maps an RCCL (PyTorch's) gets that copy, never a second one. */
int rlhip_comm_can_load(void);
const char* rlhip_comm_rccl_origin(void);   /* where the bound RCCL came from (diagnostics) */
/* ncclGetVersion of the bound RCCL (e.g. 22203; 0: not bound); how this context's collectives travel: 1 its own RCCL communicator, 2 the
 * caller's hook, 0 none (one rank) -- bench.py prints both next to rlhip_comm_size so that a scaling line proves what it ran on */
int rlhip_comm_rccl_version(void);
int rlhip_comm_kind(rlhip_ctx* ctx);
int rlhip_comm_unique_id(unsigned char id_out[128]);                 /* rank 0: ncclGetUniqueId */
int rlhip_comm_init(rlhip_ctx* ctx, int nranks, int rank, const unsigned char id[128]);
int rlhip_comm_set_hook(rlhip_ctx* ctx, rlhip_allreduce_hook hook, void* user, int nranks, int rank);
int rlhip_comm_destroy(rlhip_ctx* ctx);
int rlhip_comm_size(rlhip_ctx* ctx);
int rlhip_comm_rank(rlhip_ctx* ctx);
int rlhip_allreduce_sum_f64(rlhip_ctx* ctx, double* buf, int64_t count);
int rlhip_allreduce_sum_f32(rlhip_ctx* ctx, float* buf, int64_t count);
int rlhip_allreduce_sum_host_f64(rlhip_ctx* ctx, double* x_host, int64_t n);   /* n <= 16 scalars */

/* ---- diagnostics ---- */
/* number of times this context took a specialised kernel path: 0 persistent stream-K fp64 GEMM (gemm_sk.hip), 1 its fp32 twin,
 * 2 fused MFMA trsm block kernel (tri.hip), 3 row-per-lane substitution trsm sub-block, 4 fused out-of-place trsm (rlhip_trsm_gather),
 * 5 sketch-preconditioned Cholesky-QR panel inside geqrf (house.hip), 6 persistent one-launch Jacobi sweeps (jacobi.hip),
 * 7 column-at-a-time LU panel of a matrix taller than the resident register kernels hold (lu.hip), 8 register-resident block-pipelined
 * Householder QR of a sketch-sized matrix (qr_blk.hip), 9 its sign-modified LU twin inside orhr_col, 10 the Gram route of the device SVD
 * (svd.hip::gesdd_tall_gram: Jacobi on A^T A, one host read), 11 the one-stream Cholesky-QR (rlhip_cholqrq),
 * 12 block iterations of BQRRP whose sketch down-date and next QRCP ran on the side queue (the look-ahead of rl_bqrrp.hh; noted by the C++
 * layer through rlhip_path_note), 13 CQRRPT calls that took the split order (rl_cqrrpt.hh), 14 sparse-sign sketches applied by the LDS-DMA kernel
 * (sketch.hip::saso_apply_dma_kernel), 15 persistent Jacobi launches whose workers sat on one XCD and handed their blocks over through its L2
 * (jacobi.hip; the other launches use the uncached hand-over: same bits, slower).  -1 for an unknown index.  Tests use it to
 * assert that the kernel / route under test is the one that ran. */
int64_t rlhip_path_count(rlhip_ctx* ctx, int which);
/* the host layers above this ABI (include/RandLAPACK_amd/) report their own route decisions into the same counters */
int rlhip_path_note(rlhip_ctx* ctx, int which, int64_t delta);
/* Profiler phase markers (the reference brackets every phase of BQRRP_GPU with NVTX ranges, drivers/rl_bqrrp_gpu.hh:11,335-403; these are the
 * ROCm equivalent, roctxRangePushA / roctxRangePop of librocprofiler-sdk-roctx, bound at run time).  Active when the roctx library is already
 * in the process (rocprofv3 --marker-trace) or RLHIP_ROCTX=1; otherwise both calls return 0 and do nothing.  Host-side ranges: they bracket
 * the ENQUEUE of a phase; with the drivers' timing switch on (every lap drains the stream) they bracket its execution too. */
int rlhip_range_push(const char* name);
int rlhip_range_pop(void);
/* Scope switch for the products this context launches: on != 0 -> no kernel of this context may hold every CU until it is done (the
 * persistent stream-K GEMMs give way to the tiled ones, whose workgroups retire continuously), so that work enqueued on a side queue
 * (rlhip_side_of) finds CUs beside it.  Returns the previous setting (0 / 1), -1 on a bad context. */
int rlhip_avoid_persistent(rlhip_ctx* ctx, int on);
/* pure-MFMA issue-rate microbenchmark; returns achieved TFLOP/s of v_mfma_{f64,f32}_16x16x4 in *tflops_host */
int rlhip_mfma_peak(rlhip_ctx* ctx, int is_f64, int iters, double* tflops_host);
/* diagnostic: keep `blocks` workgroups busy for `usec` microseconds (mode 0 sleeping, 1 fp64 FMA chain, 2 fp64 MFMA stream), on the context's
 * stream (side = 0) or on a second stream beside it (side = 1); scripts/dvfs_probe.py maps the part's clock response to load with it */
int rlhip_dvfs_burn(rlhip_ctx* ctx, int blocks, int mode, int usec, int side);
/* streaming-read bandwidth microbenchmark over `bytes` of device memory (GB/s) */
int rlhip_hbm_read_peak(rlhip_ctx* ctx, const void* buf, size_t bytes, double* gbps_host);

#ifdef __cplusplus
}
#endif
#endif /* RLHIP_H */
