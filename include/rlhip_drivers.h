/* rlhip_drivers.h -- C entry points of the driver/comp objects (include/RandLAPACK_amd/), for bindings that
 * cannot instantiate C++ templates (ctypes in tests/bench, or a C caller).  Each function builds the same
 * object graph a RandLAPACK user builds (Stabilization -> RS -> RF -> QB -> RSVD) and invokes
 * call().  All matrices are DEVICE pointers, column-major.
 *
 * state[6] = RNGState: counter[0..3], key[0..1]; advanced in place exactly as the reference threads it
 * (`state = fill_dense(D, buf, state)`, comps/rl_rs.hh:135).
 * stab kinds: 0 = CholQRQ, 1 = HQRQ, 2 = PLUL (comps/rl_orth.hh:26-65,101-141,167-207).
 * Return codes are the reference's (RS 0/1, RF 0/1/2, QB 0/2/3/4/5/6, RSVD 0); -100 = RandLAPACK::Error
 * (bad argument), -101 = device/runtime failure; rlhip_last_error() gives the message.
 */
#ifndef RLHIP_DRIVERS_H
#define RLHIP_DRIVERS_H
#include "rlhip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* rlhip_last_error(void);

/* Stabilization<double>::call(m, k, A)                                   comps/rl_orth.hh:69-98 */
int rlhip_drv_stab_f64(rlhip_ctx* ctx, int kind, int cond_check, int64_t m, int64_t k, double* A, int* chol_fail);
/* RS<double>::call -> Omega (n x k, caller allocated)                    comps/rl_rs.hh:117-178 */
int rlhip_drv_rs_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t k, int64_t p, int64_t q,
                     int stab_kind, double* Omega, uint32_t state[6]);
/* RF<double>::call -> Q (m x k, caller allocated)                        comps/rl_rf.hh:107-137 */
int rlhip_drv_rf_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t k, int64_t p, int64_t q,
                     int rs_stab, int orth_kind, double* Q, uint32_t state[6]);
/* QB<double>::call; *Q (m x k) and *BT (n x k) are allocated by the callee, free with rlhip_free.
 *                                                                         comps/rl_qb.hh:134-268 */
int rlhip_drv_qb_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t* k, int64_t b_sz, double tol,
                     int64_t p, int64_t q, int rs_stab, int rf_orth, int qb_orth, int orth_check, double** Q,
                     double** BT, uint32_t state[6]);
/* RSVD<double>::call; *U (m x k), *S (k), *V (n x k) allocated by the callee, free with rlhip_free.
 * *qb_ret receives QB's return code.                                      drivers/rl_rsvd.hh:114-154 */
int rlhip_drv_rsvd_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t* k, int64_t b_sz, double tol,
                       int64_t p, int64_t q, int rs_stab, int rf_orth, int qb_orth, int orth_check, double** U,
                       double** S, double** V, uint32_t state[6], int* qb_ret);

/* CQRRPT<double>::call; qrcp {0 hqrrp, 1 bqrrp, 2 geqp3} in the reference's enum order (rl_cqrrpt.hh:41), -1 = default (geqp3); add 16 for
 * CQRRPT::orthogonalization = true (the trailing n - rank columns of A become an orthonormal completion, R is left as R_chol; :347-367).  A (m x n, lda) -> Q; R (n x n, ldr); J (n, device int64).  If A_hat_in
 * is non-NULL it is used as the d x n sketch (ld d) instead of generating a SASO (parity tests share one sketch
 * between this path and the oracle, like test/drivers/test_bqrrp_gpu.cu:91-110); if A_hat_out is non-NULL the
 * sketch that was factored is copied there BEFORE geqp3.  *rank_out = CQRRPT::rank; times_us[8] may be NULL.
 *                                                                         drivers/rl_cqrrpt.hh:147-391 */
int rlhip_drv_cqrrpt_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* R, int64_t ldr,
                         int64_t* J, double d_factor, int64_t nnz, double eps, uint32_t state[6],
                         const double* A_hat_in, double* A_hat_out, int64_t* rank_out, long* times_us, int qrcp);

/* hqrrp (drivers/rl_hqrrp.hh:812): GEQP3-format Householder QR with randomized pivoting.  qr_type 0 Householder panel
 * (pivoted inside the panel when panel_pivoting != 0), 1 geqrf, 2 CholQR (2 needs panel_pivoting == 0, as in the
 * reference's dispatch :587-591).  G_out (may be NULL): receives the (nb_alg+pp) x m Uniform(-1,1) sketching matrix.
 * Returns 0, or 1 if a CholQR panel broke down. */
int rlhip_drv_hqrrp_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, int64_t* jpvt, double* tau, int64_t nb_alg,
                        int64_t pp, int64_t panel_pivoting, int64_t qr_type, uint32_t state[6], double* G_out);
/* the same call with the reference's `T** timing` argument armed (rl_hqrrp.hh:815, :1144-1164): times27 receives the 27 entries the
 * reference's HQRRP_runtime_breakdown benchmark prints, in microseconds (layout: include/RandLAPACK_amd/rl_hqrrp.hh) */
int rlhip_drv_hqrrp_timed_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, int64_t* jpvt, double* tau, int64_t nb_alg,
                              int64_t pp, int64_t panel_pivoting, int64_t qr_type, uint32_t state[6], double times27[27]);

/* BQRRP<double>::call.  qrcp_wide {0 luqr, 1 geqp3}, qr_tall {0 geqrt, 1 cholqr, 2 geqrf}, apply_trans_q {0 ormqr, 1 gemqrt}
 * follow the reference's enum order (rl_bqrrp.hh:45-49); -1 keeps the object's default = the reference's {luqr, geqrf, ormqr}
 * (rl_bqrrp.hh:74-76).  The fastest triple on the device is {0, 1, 1} = {luqr, cholqr, gemqrt} (BQRRP::use_fast_subroutines()).
 * DIVERGENCE from the reference, qr_tall = cholqr only: when the panel's Cholesky factorization breaks down, or diag(R_chol) is
 * graded beyond eps^(1/4) (the preconditioned panel is not well conditioned, so Cholesky QR cannot deliver an orthonormal Q), the
 * panel is re-factored with Householder reflectors (BQRRP::cholqr_fallback, default on; environment RLHIP_BQRRP_CHOLQR_FALLBACK=0
 * restores the reference's behaviour, which carries on with the half-factored Gram matrix, rl_bqrrp.hh:461).  Pivots, rank and
 * well-conditioned panels are unaffected.
 * Row-sharded context (rlhip_comm_*): m and A are this rank's rows, tau needs min(global rows, n) entries; every qr_tall runs sharded
 * (cholqr: one b x b Gram all-reduce per panel; geqrf / geqrt: TSQR, the ranks' b x b triangles stacked by one all-reduce);
 * qr_tall + 16 (16 + 3 for the object's default qr_tall) selects the BLOCK-CYCLIC layout (global row blocks of b_sz rows dealt round-robin,
 * block g on rank g % P, stacked in increasing order in A) instead of one contiguous row block per rank.  A (m x n, lda) -> GEQP3
 * format, tau (min(m,n)), J (n) all on the device.  A_sk_in / A_sk_out: shared-sketch hooks as for CQRRPT (d x n,
 * ld d, d = (int64)(d_factor * b_sz)).  times_us[9] may be NULL.              drivers/rl_bqrrp.hh:155-665 */
int rlhip_drv_bqrrp_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double d_factor, int64_t b_sz,
                        int64_t internal_nb, double tol, double* tau, int64_t* J, uint32_t state[6],
                        const double* A_sk_in, double* A_sk_out, int64_t* rank_out, long* times_us, int qrcp_wide, int qr_tall,
                        int apply_trans_q);

/* BQRRP_GPU<T>::call (drivers/rl_bqrrp_gpu.hh:120-129, body :152-934): the reference's device class -- the d x n sketch A_sk (ld d,
 * DEVICE, overwritten) is an INPUT, d >= b_sz is free (not tied to a d_factor).  qr_tall in the reference's enum order
 * (BQRRPGPUSubroutines::QRTall, :44-46): 0 cholqr, 1 geqrf, -1 = the object's default (geqrf, :84).  tol <= 0 keeps eps.
 * A (m x n, lda) -> GEQP3 format, tau (n), J (n), all DEVICE.  times15 (may be NULL) receives BQRRP_GPU::times, the 15 entries of
 * rl_bqrrp_gpu.hh:829-834 in microseconds: {preallocation, qrcp_main, copy_A_sk, qrcp_piv, copy_A, piv_A, copy_J, updating_J,
 * preconditioning, qr_tall, q_reconstruction, apply_transq, sample_update, rest, total}. */
int rlhip_drv_bqrrp_gpu_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* A_sk, int64_t d, int64_t b_sz, int qr_tall,
                            double tol, double* tau, int64_t* J, int64_t* rank_out, long* times15);
int rlhip_drv_bqrrp_gpu_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* A_sk, int64_t d, int64_t b_sz, int qr_tall,
                            float tol, float* tau, int64_t* J, int64_t* rank_out, long* times15);
/* CQRRPT_GPU<T>::call (drivers/rl_cqrrpt_gpu.hh:117-127, body :149-390): HOST matrices in and out as in the reference (A m x n lda -> Q,
 * R n x n ldr, J n); every stage runs on the device, the class owns the two transfers.  nnz <= 0 keeps the object's default (2);
 * no_hqrrp: 1 geqp3 (default, :75), 0 hqrrp, -1 default.  A_hat_out_host (may be NULL): HOST buffer receiving the d x n sketch that
 * was factored (d = (int64)(d_factor * n)) -- the shared-sketch hook of the parity tests.  times8 (may be NULL): CQRRPT_GPU::times
 * {saso, qrcp, rank_reveal, cholqr, a_mod_piv, a_mod_trsm, rest, total} in microseconds (:371).  Returns 0 or 1. */
int rlhip_drv_cqrrpt_gpu_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A_host, int64_t lda, double* R_host, int64_t ldr, int64_t* J_host,
                             double d_factor, int64_t nnz, double eps, int no_hqrrp, uint32_t state[6], double* A_hat_out_host,
                             int64_t* rank_out, long* times8);
int rlhip_drv_cqrrpt_gpu_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A_host, int64_t lda, float* R_host, int64_t ldr, int64_t* J_host,
                             float d_factor, int64_t nnz, float eps, int no_hqrrp, uint32_t state[6], float* A_hat_out_host,
                             int64_t* rank_out, long* times8);

/* ABRIK<double>::call(m, n, A, lda, k, U, V, Sigma, state), the dense-pointer overload (drivers/rl_abrik.hh:122-143).  A (m x n, lda) is not
 * modified.  *U (m x triplets), *Sigma (triplets), *V (n x triplets) are allocated by the callee (rlhip_malloc), freed by the caller.
 * max_krylov_iters <= 0 keeps the object's default (INT_MAX). */
int rlhip_drv_abrik_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t lda, int64_t k, double tol, int64_t max_krylov_iters,
                        double** U, double** Sigma, double** V, uint32_t state[6], int64_t* triplets, int64_t* iters, double* norm_R_end,
                        int qr_exp /* 0 geqrf_ungqr, 1 cqrrt, -1 default (geqrf_ungqr); both run on a row-sharded context */);

/* CQRRT<double>::call (drivers/rl_cqrrt.hh:124): unpivoted CQRRPT.  A (m x n, lda) -> Q; upper triangle of R (n x n, ldr).
 * A_hat_in / A_hat_out: shared-sketch hooks as for CQRRPT (d x n, ld d, d = (int64)(d_factor * n)).  Returns 0 or 1. */
int rlhip_drv_cqrrt_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* R, int64_t ldr, double d_factor, int64_t nnz,
                        double eps, uint32_t state[6], const double* A_hat_in, double* A_hat_out);

/* fp32 instantiations (same contracts; tol / eps / d_factor are float) */
int rlhip_drv_stab_f32(rlhip_ctx* ctx, int kind, int cond_check, int64_t m, int64_t k, float* A, int* chol_fail);
int rlhip_drv_rsvd_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t* k, int64_t b_sz, float tol,
                       int64_t p, int64_t q, int rs_stab, int rf_orth, int qb_orth, int orth_check, float** U,
                       float** S, float** V, uint32_t state[6], int* qb_ret);
int rlhip_drv_cqrrpt_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* R, int64_t ldr,
                         int64_t* J, float d_factor, int64_t nnz, float eps, uint32_t state[6],
                         const float* A_hat_in, float* A_hat_out, int64_t* rank_out, long* times_us, int qrcp);
int rlhip_drv_hqrrp_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, int64_t* jpvt, float* tau, int64_t nb_alg,
                        int64_t pp, int64_t panel_pivoting, int64_t qr_type, uint32_t state[6], float* G_out);
int rlhip_drv_bqrrp_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float d_factor, int64_t b_sz,
                        int64_t internal_nb, float tol, float* tau, int64_t* J, uint32_t state[6],
                        const float* A_sk_in, float* A_sk_out, int64_t* rank_out, long* times_us, int qrcp_wide, int qr_tall,
                        int apply_trans_q);

/* ---- drivers over abstract linear operators (include/RandLAPACK_amd/rl_linops.hh, rl_qr_linops.hh).
 * An operator is described by plain arrays: kind 0 = dense column-major (dense, ld); kind 1 = CSR with int64 indices
 * (rowptr, colidx, vals; nnz entries); kind 2 = CSC (RandBLAS::sparse_data::CSCMatrix: the `rowptr` field carries colptr (cols + 1), the `colidx`
 * field the ROW indices); kind 3 = COO (COOMatrix: `rowptr` carries the nnz row indices, `colidx` the nnz column indices; any order, duplicates
 * are summed).  All arrays are DEVICE pointers.  `right` == NULL: the operator is `left`;
 * otherwise it is the implicit product left * right (linops::CompositeOperator, rl_composite_linop.hh:43). */
typedef struct rlhip_linop_desc {
    int kind;
    int64_t rows, cols;
    const void* dense;
    int64_t ld;
    int64_t nnz;
    const int64_t* rowptr;
    const int64_t* colidx;
    const void* vals;
} rlhip_linop_desc;

/* alg: 0 CholQR_linops (rl_cholqr_linops.hh:60), 1 sCholQR3_linops (rl_scholqr3_linops.hh:182), 2 sCholQR3_linops_basic (:600),
 *      3 CQRRT_linops (rl_cqrrt_linops.hh:144).  R (n x n, ldr) receives the upper-triangular factor.  block_size as the classes'
 *      member.  Q_out != NULL turns test_mode on and receives a library-owned copy of Q (m x n; free with rlhip_free).
 *      d_factor, nnz, use_dense_sketch, state, A_hat_in/out (d x n sketch injection / export): CQRRT_linops only.
 *      Returns the class's return value (0 ok, >0 Cholesky breakdown index), negative on argument / device errors. */
int rlhip_drv_qr_linops_f64(rlhip_ctx* ctx, int alg, const rlhip_linop_desc* left, const rlhip_linop_desc* right, double* R, int64_t ldr,
                            int64_t block_size, double** Q_out, double d_factor, int64_t nnz, int use_dense_sketch, uint32_t state[6],
                            const double* A_hat_in, double* A_hat_out);
int rlhip_drv_qr_linops_f32(rlhip_ctx* ctx, int alg, const rlhip_linop_desc* left, const rlhip_linop_desc* right, float* R, int64_t ldr,
                            int64_t block_size, float** Q_out, float d_factor, int64_t nnz, int use_dense_sketch, uint32_t state[6],
                            const float* A_hat_in, float* A_hat_out);
/* ABRIK<double>::call on an operator given by descriptor(s) (rl_abrik.hh:166; sparse / composite operators as in
 * benchmark/bench_ABRIK/ABRIK_speed_comparisons_sparse.cc).  Outputs as rlhip_drv_abrik_f64. */
int rlhip_drv_abrik_linop_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, const rlhip_linop_desc* right /* must be NULL */, int64_t k, double tol,
                              int64_t max_krylov_iters, double** U, double** Sigma, double** V, uint32_t state[6], int64_t* triplets,
                              int64_t* iters, double* norm_R_end, int qr_exp);
/* The same call with ABRIK's subroutine timers armed (ABRIK(verbose, time_subroutines = true, tol), rl_abrik.hh:64; the 13 entries of
 * ABRIK::times, rl_abrik.hh:733-734, in microseconds: allocation, get_factors, ungqr, reorth, qr, gemm_A, main_loop, sketching, r_cpy,
 * s_cpy, norm, rest, total).  Every lap drains the stream, so a timed call is slower than an untimed one. */
int rlhip_drv_abrik_linop_timed_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, int64_t k, double tol, int64_t max_krylov_iters, double** U,
                                    double** Sigma, double** V, uint32_t state[6], int64_t* triplets, int64_t* iters, double* norm_R_end,
                                    int qr_exp, long times[13]);
/* C (m x n, ldc) = alpha * op(A) * B + beta * C for an operator given by descriptor(s): the raw operator call, for tests and for
 * callers that only want the SpMM / composite product.  side 'L' or 'R', trans 'N' or 'T', column-major B and C. */
int rlhip_linop_apply_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, const rlhip_linop_desc* right, char side, char trans, int64_t m,
                          int64_t n, int64_t k, double alpha, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);

/* The same product with a BLOCK VIEW of the operator (rl_dense_linop.hh:295-330, rl_sparse_linop.hh:393-465, rl_composite_linop.hh:505-530):
 * how 0 = row_block(view[0], view[2]), 1 = col_block(view[1], view[3]), 2 = submatrix(view[0], view[1], view[2], view[3]);
 * view = {row_start, col_start, row_count, col_count}; m, n, k describe the product with the VIEW. */
int rlhip_linop_apply_view_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, const rlhip_linop_desc* right, int how, const int64_t view[4], char side,
                               char trans, int64_t m, int64_t n, int64_t k, double alpha, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
/* linops::RegExplicitSymLinOp (rl_sym_linops.hh:134-233): C (dim x n) = alpha * (A + mu_i I) * B + beta * C with A given by its UPPER triangle
 * (dim x dim, lda, DEVICE; the strictly lower triangle is never read); regs_host[num_ops] on the HOST; eval_includes_reg as
 * set_eval_includes_reg; with num_ops > 1 column i takes regs[i] and n must equal num_ops. */
int rlhip_regsym_apply_f64(rlhip_ctx* ctx, int64_t dim, const double* A, int64_t lda, const double* regs_host, int64_t num_ops, int eval_includes_reg,
                           int64_t n, double alpha, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);

/* gen::mat_gen (RandLAPACK/testing/rl_gen.hh:712-772) in HBM.  type: 0 polynomial, 1 exponential, 2 gaussian, 3 step, 4 spiked,
 * 5 adverserial, 6 bad_cholqr, 7 kahan (the enum order of rl_gen.hh:22-31).  A: m x n DEVICE (rank x rank, ld rank, when diag != 0).
 * Fields as mat_gen_info; rank_out (may be NULL) receives info.rank after the call (changed only by check_true_rank). */
int rlhip_drv_mat_gen_f64(rlhip_ctx* ctx, int type, int64_t m, int64_t n, int64_t rank, double cond_num, double scaling, double exponent,
                          int diag, double theta, double perturb, double frac_spectrum_one, int check_true_rank, double* A,
                          uint32_t state[6], int64_t* rank_out);
int rlhip_drv_mat_gen_f32(rlhip_ctx* ctx, int type, int64_t m, int64_t n, int64_t rank, float cond_num, float scaling, float exponent,
                          int diag, float theta, float perturb, float frac_spectrum_one, int check_true_rank, float* A,
                          uint32_t state[6], int64_t* rank_out);

/* REVD2<SYRF<SYPS, Stabilization>>::call (drivers/rl_revd2.hh:96-243; comps/rl_syrf.hh, rl_syps.hh): A ~ V diag(eigvals) V^T for a
 * symmetric PSD A given by its `uplo` triangle (m x m, ld m, DEVICE; the other triangle is never read).  *k: in = starting rank,
 * out = rank used.  *V (m x k) and *eigvals (k) are library-allocated DEVICE buffers (rlhip_free).  orth_kind: 0 CholQRQ, 1 HQRQ,
 * 2 PLUL (the reference's tests use HQRQ).  err_out (may be NULL): the final power-method error estimate. */
int rlhip_drv_revd2_f64(rlhip_ctx* ctx, char uplo, int64_t m, const double* A, int64_t* k, double tol, int64_t syps_passes,
                        int64_t passes_per_stab, int error_est_p, int orth_kind, double** V, double** eigvals, uint32_t state[6],
                        double* err_out);
/* SYRF::call alone (comps/rl_syrf.hh:44-92): Q (m x k, DEVICE, caller-allocated) <- orth(A * SYPS sketch). */
int rlhip_drv_syrf_f64(rlhip_ctx* ctx, char uplo, int64_t m, const double* A, int64_t k, int64_t syps_passes, int64_t passes_per_stab,
                       int orth_kind, double* Q, uint32_t state[6]);

#ifdef __cplusplus
}
#endif
#endif
