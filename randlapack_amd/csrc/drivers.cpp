// C entry points over the C++ driver/comp objects (include/rlhip_drivers.h).  Host-only C++: everything that
// touches the GPU goes through the C ABI in rlhip.h.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <climits>
#include "../../include/RandLAPACK_amd.hh"
#include "../../include/rlhip_drivers.h"

namespace {

thread_local std::string g_last_error;
using RNG = r123::Philox4x32;
using State = RandBLAS::RNGState<RNG>;

State load_state(const uint32_t s[6]) {
    State st;
    for (int i = 0; i < 4; ++i) st.counter[i] = s[i];
    st.key[0] = s[4];
    st.key[1] = s[5];
    return st;
}
void store_state(State const& st, uint32_t s[6]) {
    for (int i = 0; i < 4; ++i) s[i] = st.counter[i];
    s[4] = st.key[0];
    s[5] = st.key[1];
}

template <typename T>
std::unique_ptr<RandLAPACK::Stabilization<T>> make_stab(blas::Queue& q, int kind, bool cond_check) {
    switch (kind) {
        case 0: return std::make_unique<RandLAPACK::CholQRQ<T>>(q, cond_check, false);
        case 1: return std::make_unique<RandLAPACK::HQRQ<T>>(q, cond_check, false);
        case 2: return std::make_unique<RandLAPACK::PLUL<T>>(q, cond_check, false);
        default: throw RandLAPACK::Error("unknown stabilization kind " + std::to_string(kind) + " (0 CholQRQ, 1 HQRQ, 2 PLUL)");
    }
}

// Defaults that the context's options (rlhip_set_option, RLHIP_OPT_DRV_*) hand to the objects these entry points construct; -1 keeps the
// object's own default.  C++ callers set the members themselves.
template <typename Alg>
void apply_bqrrp_options(rlhip_ctx* ctx, Alg& alg) {
    const int64_t la = rlhip_get_option(ctx, RLHIP_OPT_DRV_BQRRP_LOOKAHEAD_MIN_ELEMS);
    if (la >= 0) { alg.lookahead_min_elems = (double)la; if (la == 0) alg.lookahead_min_block = 1; }   // 0: the side-queue path whenever the dependency structure allows it
    const int64_t fb = rlhip_get_option(ctx, RLHIP_OPT_DRV_BQRRP_CHOLQR_FALLBACK);
    if (fb >= 0) alg.cholqr_fallback = (fb != 0);
}
template <typename Alg>
void apply_cqrrpt_options(rlhip_ctx* ctx, Alg& alg) {
    const int64_t fp = rlhip_get_option(ctx, RLHIP_OPT_DRV_CQRRPT_FOLD_PIVOTING);
    if (fp >= 0) alg.fold_pivoting = (fp != 0);
    const int64_t sq = rlhip_get_option(ctx, RLHIP_OPT_DRV_CQRRPT_SPLIT_QRCP);
    if (sq >= 0) alg.split_qrcp = (sq != 0);
}

template <typename F>
int guarded(F&& f) {
    try {
        return f();
    } catch (RandLAPACK::Error const& e) {
        g_last_error = e.what();
        return -100;
    } catch (std::exception const& e) {
        g_last_error = e.what();
        return -101;
    }
}


// ---- operators from descriptors: run `f` on the concrete operator type the descriptor(s) name
namespace lo = RandLAPACK::linops;
template <typename T>
std::unique_ptr<lo::DenseLinOp<T>> make_dense(blas::Queue& q, const rlhip_linop_desc& d) {
    auto op = std::make_unique<lo::DenseLinOp<T>>(d.rows, d.cols, (const T*)d.dense, d.ld, RandLAPACK::Layout::ColMajor, q);
    op->row_sharded = q.world() > 1;
    return op;
}
template <typename T>
std::unique_ptr<lo::SparseLinOp<T>> make_sparse(blas::Queue& q, const rlhip_linop_desc& d) {
    // kind 1 CSR; 2 CSC (rowptr field = colptr, colidx field = row indices); 3 COO (rowptr field = row indices, colidx field = column indices)
    std::unique_ptr<lo::SparseLinOp<T>> op;
    if (d.kind == 2) op = std::make_unique<lo::SparseLinOp<T>>(lo::SparseLinOp<T>::from_csc(d.rows, d.cols, d.nnz, d.rowptr, d.colidx, (const T*)d.vals, q));
    else if (d.kind == 3) op = std::make_unique<lo::SparseLinOp<T>>(lo::SparseLinOp<T>::from_coo(d.rows, d.cols, d.nnz, d.rowptr, d.colidx, (const T*)d.vals, q));
    else op = std::make_unique<lo::SparseLinOp<T>>(d.rows, d.cols, d.nnz, d.rowptr, d.colidx, (const T*)d.vals, q);
    op->force_densified_sketch = rlhip_get_option(q.ctx(), RLHIP_OPT_DRV_SPARSE_SKETCH_DENSIFY) == 1;              // (tests: the fallback path)
    op->row_sharded = q.world() > 1;
    return op;
}
template <typename T, typename F>
int with_operator(blas::Queue& q, const rlhip_linop_desc* left, const rlhip_linop_desc* right, F&& f) {
    if (!left) throw RandLAPACK::Error("operator descriptor is null");
    auto kind_ok = [](const rlhip_linop_desc* d) { return d->kind >= 0 && d->kind <= 3; };
    if (!kind_ok(left) || (right && !kind_ok(right))) throw RandLAPACK::Error("operator kind must be 0 (dense), 1 (CSR), 2 (CSC) or 3 (COO)");
    if (!right) {
        if (left->kind == 0) { auto A = make_dense<T>(q, *left); return f(*A); }
        auto A = make_sparse<T>(q, *left);
        return f(*A);
    }
    const int64_t m = left->rows, n = right->cols;
    if (left->kind == 0 && right->kind == 0) {
        auto L = make_dense<T>(q, *left); auto Rr = make_dense<T>(q, *right); Rr->row_sharded = false;
        lo::CompositeOperator<lo::DenseLinOp<T>, lo::DenseLinOp<T>> A(m, n, *L, *Rr);
        return f(A);
    }
    if (left->kind == 0 && right->kind >= 1) {
        auto L = make_dense<T>(q, *left); auto Rr = make_sparse<T>(q, *right); Rr->row_sharded = false;
        lo::CompositeOperator<lo::DenseLinOp<T>, lo::SparseLinOp<T>> A(m, n, *L, *Rr);
        return f(A);
    }
    if (left->kind >= 1 && right->kind == 0) {
        auto L = make_sparse<T>(q, *left); auto Rr = make_dense<T>(q, *right); Rr->row_sharded = false;
        lo::CompositeOperator<lo::SparseLinOp<T>, lo::DenseLinOp<T>> A(m, n, *L, *Rr);
        return f(A);
    }
    auto L = make_sparse<T>(q, *left); auto Rr = make_sparse<T>(q, *right); Rr->row_sharded = false;
    lo::CompositeOperator<lo::SparseLinOp<T>, lo::SparseLinOp<T>> A(m, n, *L, *Rr);
    return f(A);
}

template <typename T>
void export_q(blas::Queue& q, const T* Q, int64_t rows, int64_t cols, T** Q_out) {
    *Q_out = nullptr;
    if (!Q) return;
    *Q_out = blas::device_malloc<T>(rows * cols, q);
    blas::device_copy_vector(rows * cols, Q, *Q_out, q);
}

template <typename T>
int drv_qr_linops(rlhip_ctx* ctx, int alg, const rlhip_linop_desc* left, const rlhip_linop_desc* right, T* R, int64_t ldr, int64_t block_size,
                  T** Q_out, T d_factor, int64_t nnz, int use_dense_sketch, uint32_t state[6], const T* A_hat_in, T* A_hat_out) {
    return guarded([&] {
        blas::Queue q(ctx);
        const bool tm = Q_out != nullptr;
        const T eps = std::pow(std::numeric_limits<T>::epsilon(), (T)0.75);
        return with_operator<T>(q, left, right, [&](auto& A) -> int {
            switch (alg) {
                case 0: {
                    RandLAPACK::CholQR_linops<T> a(false, eps, tm);
                    a.block_size = block_size;
                    int rc = a.call(A, R, ldr);
                    if (tm) export_q(q, a.Q, a.Q_rows, a.Q_cols, Q_out);
                    return rc;
                }
                case 1: {
                    RandLAPACK::sCholQR3_linops<T> a(false, eps, tm);
                    a.block_size = block_size;
                    int rc = a.call(A, R, ldr);
                    if (tm) export_q(q, a.Q, a.Q_rows, a.Q_cols, Q_out);
                    return rc;
                }
                case 2: {
                    RandLAPACK::sCholQR3_linops_basic<T> a(false, eps, tm);
                    int rc = a.call(A, R, ldr);
                    if (tm) export_q(q, a.Q, a.Q_rows, a.Q_cols, Q_out);
                    return rc;
                }
                case 3: {
                    RandLAPACK::CQRRT_linops<T, RNG> a(false, eps, tm);
                    a.block_size = block_size;
                    if (nnz > 0) a.nnz = nnz;
                    a.use_dense_sketch = use_dense_sketch != 0;
                    a.sketch_override = A_hat_in;
                    a.sketch_export = A_hat_out;
                    State st = load_state(state);
                    int rc = a.call(A, R, ldr, d_factor, st);
                    store_state(st, state);
                    if (tm) export_q(q, a.Q, a.Q_rows, a.Q_cols, Q_out);
                    return rc;
                }
                default: throw RandLAPACK::Error("alg must be 0 (CholQR), 1 (sCholQR3), 2 (sCholQR3 basic) or 3 (CQRRT)");
            }
        });
    });
}


template <typename T>
int drv_mat_gen(rlhip_ctx* ctx, int type, int64_t m, int64_t n, int64_t rank, T cond_num, T scaling, T exponent, int diag, T theta, T perturb,
                T frac_spectrum_one, int check_true_rank, T* A, uint32_t state[6], int64_t* rank_out) {
    return guarded([&] {
        blas::Queue q(ctx);
        if (type < 0 || type > 7) throw RandLAPACK::Error("mat_gen: type must be 0..7 (rl_gen.hh mat_type order, custom_input excluded)");
        RandLAPACK::gen::mat_gen_info<T> info(m, n, (RandLAPACK::gen::mat_type)type);
        info.rank = rank;
        info.cond_num = cond_num; info.scaling = scaling; info.exponent = exponent; info.diag = diag != 0;
        info.theta = theta; info.perturb = perturb; info.frac_spectrum_one = frac_spectrum_one; info.check_true_rank = check_true_rank != 0;
        State st = load_state(state);
        RandLAPACK::gen::mat_gen(info, A, st, q);
        store_state(st, state);
        if (rank_out) *rank_out = info.rank;
        return 0;
    });
}
}  // namespace

extern "C" {

const char* rlhip_last_error(void) { return g_last_error.c_str(); }

int rlhip_drv_stab_f64(rlhip_ctx* ctx, int kind, int cond_check, int64_t m, int64_t k, double* A, int* chol_fail) {
    return guarded([&] {
        blas::Queue q(ctx);
        auto st = make_stab<double>(q, kind, cond_check != 0);
        blas::RowsSharded sh(q, true);                     // (a context with a communicator holds a row block of the tall matrix)
        int rc = st->call(m, k, A);
        if (chol_fail) {
            auto* c = dynamic_cast<RandLAPACK::CholQRQ<double>*>(st.get());
            *chol_fail = (c && c->chol_fail) ? 1 : 0;
        }
        return rc;
    });
}

int rlhip_drv_rs_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t k, int64_t p, int64_t q_,
                     int stab_kind, double* Omega, uint32_t state[6]) {
    return guarded([&] {
        blas::Queue q(ctx);
        auto stab = make_stab<double>(q, stab_kind, false);
        RandLAPACK::RS<double, RNG> rs(q, *stab, p, q_, false, false);
        State st = load_state(state);
        int rc = rs.call(m, n, A, k, Omega, st);
        store_state(st, state);
        return rc;
    });
}

int rlhip_drv_rf_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t k, int64_t p, int64_t q_,
                     int rs_stab, int orth_kind, double* Q, uint32_t state[6]) {
    return guarded([&] {
        blas::Queue q(ctx);
        auto s1 = make_stab<double>(q, rs_stab, false);
        auto s2 = make_stab<double>(q, orth_kind, false);
        RandLAPACK::RS<double, RNG> rs(q, *s1, p, q_, false, false);
        RandLAPACK::RF<double, RNG> rf(q, rs, *s2, false, false);
        State st = load_state(state);
        int rc = rf.call(m, n, A, k, Q, st);
        store_state(st, state);
        return rc;
    });
}

int rlhip_drv_qb_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t* k, int64_t b_sz, double tol, int64_t p,
                     int64_t q_, int rs_stab, int rf_orth, int qb_orth, int orth_check, double** Q, double** BT,
                     uint32_t state[6]) {
    return guarded([&] {
        blas::Queue q(ctx);
        auto s1 = make_stab<double>(q, rs_stab, false);
        auto s2 = make_stab<double>(q, rf_orth, false);
        auto s3 = make_stab<double>(q, qb_orth, false);
        RandLAPACK::RS<double, RNG> rs(q, *s1, p, q_, false, false);
        RandLAPACK::RF<double, RNG> rf(q, rs, *s2, false, false);
        RandLAPACK::QB<double, RNG> qb(q, rf, *s3, false, orth_check != 0);
        State st = load_state(state);
        *Q = nullptr;
        *BT = nullptr;
        int rc = qb.call(m, n, A, *k, b_sz, tol, *Q, *BT, st);
        store_state(st, state);
        return rc;
    });
}

int rlhip_drv_rsvd_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t* k, int64_t b_sz, double tol, int64_t p,
                       int64_t q_, int rs_stab, int rf_orth, int qb_orth, int orth_check, double** U, double** S,
                       double** V, uint32_t state[6], int* qb_ret) {
    return guarded([&] {
        blas::Queue q(ctx);
        auto s1 = make_stab<double>(q, rs_stab, false);
        auto s2 = make_stab<double>(q, rf_orth, false);
        auto s3 = make_stab<double>(q, qb_orth, false);
        RandLAPACK::RS<double, RNG> rs(q, *s1, p, q_, false, false);
        RandLAPACK::RF<double, RNG> rf(q, rs, *s2, false, false);
        RandLAPACK::QB<double, RNG> qb(q, rf, *s3, false, orth_check != 0);
        RandLAPACK::RSVD<double, RNG> rsvd(q, qb, b_sz);
        State st = load_state(state);
        *U = *S = *V = nullptr;
        int rc = rsvd.call(m, n, A, *k, tol, *U, *S, *V, st);
        store_state(st, state);
        if (qb_ret) *qb_ret = rsvd.qb_return;
        return rc;
    });
}

int rlhip_drv_cqrrpt_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* R, int64_t ldr,
                         int64_t* J, double d_factor, int64_t nnz, double eps, uint32_t state[6],
                         const double* A_hat_in, double* A_hat_out, int64_t* rank_out, long* times_us, int qrcp) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::CQRRPT<double, RNG> alg(q, times_us != nullptr, eps);
        apply_cqrrpt_options(ctx, alg);
        alg.nnz = nnz;
        if (qrcp >= 16) { alg.orthogonalization = true; qrcp -= 16; }      // +16: CQRRPT::orthogonalization = true (rl_cqrrpt.hh:347-367)
        if (qrcp >= 0) {
            if (qrcp > 2) throw RandLAPACK::Error("qrcp must be 0 (hqrrp), 1 (bqrrp) or 2 (geqp3)");
            alg.qrcp = (RandLAPACK::CQRRPTSubroutines::QRCP)qrcp;
        }
        alg.sketch_override = A_hat_in;
        alg.sketch_export = A_hat_out;
        State st = load_state(state);
        int rc = alg.call(m, n, A, lda, R, ldr, J, d_factor, st);
        store_state(st, state);
        if (rank_out) *rank_out = alg.rank;
        if (times_us && alg.times.size() == 8)
            for (int i = 0; i < 8; ++i) times_us[i] = alg.times[i];
        return rc;
    });
}

int rlhip_drv_hqrrp_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, int64_t* jpvt, double* tau, int64_t nb_alg,
                        int64_t pp, int64_t panel_pivoting, int64_t qr_type, uint32_t state[6], double* G_out) {
    return guarded([&] {
        blas::Queue q(ctx);
        State st = load_state(state);
        int rc = (int)RandLAPACK::hqrrp<double, RNG>(m, n, A, lda, jpvt, tau, nb_alg, pp, panel_pivoting, qr_type, st, q, G_out);
        store_state(st, state);
        return rc;
    });
}

int rlhip_drv_hqrrp_timed_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, int64_t* jpvt, double* tau, int64_t nb_alg,
                              int64_t pp, int64_t panel_pivoting, int64_t qr_type, uint32_t state[6], double times27[27]) {
    if (!times27) return -13;
    return guarded([&] {
        blas::Queue q(ctx);
        State st = load_state(state);
        double* tt = nullptr;
        int rc = (int)RandLAPACK::hqrrp<double, RNG>(m, n, A, lda, jpvt, tau, nb_alg, pp, panel_pivoting, qr_type, st, q, (double*)nullptr, &tt);
        store_state(st, state);
        for (int i = 0; i < 27; ++i) times27[i] = tt ? tt[i] : 0.0;
        std::free(tt);
        return rc;
    });
}

int rlhip_drv_bqrrp_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double d_factor, int64_t b_sz,
                        int64_t internal_nb, double tol, double* tau, int64_t* J, uint32_t state[6],
                        const double* A_sk_in, double* A_sk_out, int64_t* rank_out, long* times_us, int qrcp_wide, int qr_tall,
                        int apply_trans_q) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::BQRRP<double, RNG> alg(q, times_us != nullptr, b_sz);
        using Sub = RandLAPACK::BQRRPSubroutines;
        if (qrcp_wide >= 0) { if (qrcp_wide > 1) throw RandLAPACK::Error("qrcp_wide must be 0 (luqr) or 1 (geqp3)"); alg.qrcp_wide = (Sub::QRCPWide)qrcp_wide; }
        if (qr_tall >= 16) { alg.rows_block_cyclic = true; qr_tall -= 16; if (qr_tall == 3) qr_tall = -1; }     // + 16: block-cyclic row layout of a sharded call (16 + 3: the object's default qr_tall)
        apply_bqrrp_options(ctx, alg);
        if (qr_tall >= 0) { if (qr_tall > 2) throw RandLAPACK::Error("qr_tall must be 0 (geqrt), 1 (cholqr) or 2 (geqrf)"); alg.qr_tall = (Sub::QRTall)qr_tall; }
        if (apply_trans_q >= 0) { if (apply_trans_q > 1) throw RandLAPACK::Error("apply_trans_q must be 0 (ormqr) or 1 (gemqrt)"); alg.apply_trans_q = (Sub::ApplyTransQ)apply_trans_q; }
        if (internal_nb > 0) alg.internal_nb = internal_nb;
        if (tol > 0) alg.tol = tol;
        alg.sketch_override = A_sk_in;
        alg.sketch_export = A_sk_out;
        State st = load_state(state);
        int rc = alg.call(m, n, A, lda, d_factor, tau, J, st);
        store_state(st, state);
        if (rank_out) *rank_out = alg.rank;
        if (times_us && alg.times.size() == 9)
            for (int i = 0; i < 9; ++i) times_us[i] = alg.times[i];
        return rc;
    });
}

int rlhip_drv_cqrrt_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* R, int64_t ldr, double d_factor, int64_t nnz,
                        double eps, uint32_t state[6], const double* A_hat_in, double* A_hat_out) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::CQRRT<double, RNG> alg(q, false, eps);
        if (nnz > 0) alg.nnz = nnz;
        alg.sketch_override = A_hat_in;
        alg.sketch_export = A_hat_out;
        State st = load_state(state);
        int rc = alg.call(m, n, A, lda, R, ldr, d_factor, st);
        store_state(st, state);
        return rc;
    });
}

int rlhip_drv_abrik_f64(rlhip_ctx* ctx, int64_t m, int64_t n, const double* A, int64_t lda, int64_t k, double tol, int64_t max_krylov_iters,
                        double** U, double** Sigma, double** V, uint32_t state[6], int64_t* triplets, int64_t* iters, double* norm_R_end,
                        int qr_exp) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::ABRIK<double, RNG> alg(q, false, false, tol);
        if (qr_exp >= 0) {
            if (qr_exp > 1) throw RandLAPACK::Error("qr_exp must be 0 (geqrf_ungqr) or 1 (cqrrt)");
            alg.qr_exp = (RandLAPACK::ABRIKSubroutines::QR_explicit)qr_exp;
        }
        if (max_krylov_iters > 0) alg.max_krylov_iters = (int)std::min<int64_t>(max_krylov_iters, INT_MAX);
        State st = load_state(state);
        *U = nullptr; *Sigma = nullptr; *V = nullptr;
        int rc = alg.call(m, n, const_cast<double*>(A), lda, k, *U, *V, *Sigma, st);      // the dense-pointer overload (rl_abrik.hh:122-143)
        store_state(st, state);
        if (triplets) *triplets = alg.singular_triplets_found;
        if (iters) *iters = alg.num_krylov_iters;
        if (norm_R_end) *norm_R_end = alg.norm_R_end;
        return rc;
    });
}

// ---- the reference's device classes (drivers/rl_bqrrp_gpu.hh, rl_cqrrpt_gpu.hh)
}  // extern "C"
template <typename T>
static int drv_bqrrp_gpu(rlhip_ctx* ctx, int64_t m, int64_t n, T* A, int64_t lda, T* A_sk, int64_t d, int64_t b_sz, int qr_tall, T tol, T* tau,
                         int64_t* J, int64_t* rank_out, long* times15) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::BQRRP_GPU<T, RNG> alg(q, times15 != nullptr, b_sz);
        apply_bqrrp_options(ctx, alg);
        using Sub = RandLAPACK::BQRRPGPUSubroutines;
        if (qr_tall >= 0) {
            if (qr_tall > 1) throw RandLAPACK::Error("BQRRP_GPU qr_tall must be 0 (cholqr) or 1 (geqrf)");
            alg.qr_tall = (Sub::QRTall)qr_tall;
        }
        if (tol > 0) alg.tol = tol;
        int rc = alg.call(m, n, A, lda, A_sk, d, tau, J);
        if (rank_out) *rank_out = alg.rank;
        if (times15 && alg.times.size() == 15)
            for (int i = 0; i < 15; ++i) times15[i] = alg.times[(size_t)i];
        return rc;
    });
}
template <typename T>
static int drv_cqrrpt_gpu(rlhip_ctx* ctx, int64_t m, int64_t n, T* A_host, int64_t lda, T* R_host, int64_t ldr, int64_t* J_host, T d_factor,
                          int64_t nnz, T eps, int no_hqrrp, uint32_t state[6], T* A_hat_out_host, int64_t* rank_out, long* times8);
extern "C" {
int rlhip_drv_bqrrp_gpu_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A, int64_t lda, double* A_sk, int64_t d, int64_t b_sz, int qr_tall,
                            double tol, double* tau, int64_t* J, int64_t* rank_out, long* times15) {
    return drv_bqrrp_gpu<double>(ctx, m, n, A, lda, A_sk, d, b_sz, qr_tall, tol, tau, J, rank_out, times15);
}
int rlhip_drv_bqrrp_gpu_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* A_sk, int64_t d, int64_t b_sz, int qr_tall,
                            float tol, float* tau, int64_t* J, int64_t* rank_out, long* times15) {
    return drv_bqrrp_gpu<float>(ctx, m, n, A, lda, A_sk, d, b_sz, qr_tall, tol, tau, J, rank_out, times15);
}

}  // extern "C"
template <typename T>
static int drv_cqrrpt_gpu(rlhip_ctx* ctx, int64_t m, int64_t n, T* A_host, int64_t lda, T* R_host, int64_t ldr, int64_t* J_host, T d_factor,
                          int64_t nnz, T eps, int no_hqrrp, uint32_t state[6], T* A_hat_out_host, int64_t* rank_out, long* times8) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::CQRRPT_GPU<T, RNG> alg(q, false, times8 != nullptr, eps);
        if (nnz > 0) alg.nnz = nnz;
        if (no_hqrrp >= 0) alg.no_hqrrp = no_hqrrp;
        alg.sketch_export_host = A_hat_out_host;
        State st = load_state(state);
        int rc = alg.call(m, n, A_host, lda, R_host, ldr, J_host, d_factor, st);
        store_state(st, state);
        if (rank_out) *rank_out = alg.rank;
        if (times8 && alg.times.size() == 8)
            for (int i = 0; i < 8; ++i) times8[i] = alg.times[(size_t)i];
        return rc;
    });
}
extern "C" {
int rlhip_drv_cqrrpt_gpu_f64(rlhip_ctx* ctx, int64_t m, int64_t n, double* A_host, int64_t lda, double* R_host, int64_t ldr, int64_t* J_host,
                             double d_factor, int64_t nnz, double eps, int no_hqrrp, uint32_t state[6], double* A_hat_out_host,
                             int64_t* rank_out, long* times8) {
    return drv_cqrrpt_gpu<double>(ctx, m, n, A_host, lda, R_host, ldr, J_host, d_factor, nnz, eps, no_hqrrp, state, A_hat_out_host, rank_out, times8);
}
int rlhip_drv_cqrrpt_gpu_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A_host, int64_t lda, float* R_host, int64_t ldr, int64_t* J_host,
                             float d_factor, int64_t nnz, float eps, int no_hqrrp, uint32_t state[6], float* A_hat_out_host,
                             int64_t* rank_out, long* times8) {
    return drv_cqrrpt_gpu<float>(ctx, m, n, A_host, lda, R_host, ldr, J_host, d_factor, nnz, eps, no_hqrrp, state, A_hat_out_host, rank_out, times8);
}

// ---- fp32 instantiations of the same objects (BASELINE config 4 is fp32)
int rlhip_drv_stab_f32(rlhip_ctx* ctx, int kind, int cond_check, int64_t m, int64_t k, float* A, int* chol_fail) {
    return guarded([&] {
        blas::Queue q(ctx);
        auto st = make_stab<float>(q, kind, cond_check != 0);
        blas::RowsSharded sh(q, true);                     // (a context with a communicator holds a row block of the tall matrix)
        int rc = st->call(m, k, A);
        if (chol_fail) {
            auto* c = dynamic_cast<RandLAPACK::CholQRQ<float>*>(st.get());
            *chol_fail = (c && c->chol_fail) ? 1 : 0;
        }
        return rc;
    });
}

int rlhip_drv_rsvd_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t* k, int64_t b_sz, float tol, int64_t p,
                       int64_t q_, int rs_stab, int rf_orth, int qb_orth, int orth_check, float** U, float** S,
                       float** V, uint32_t state[6], int* qb_ret) {
    return guarded([&] {
        blas::Queue q(ctx);
        auto s1 = make_stab<float>(q, rs_stab, false);
        auto s2 = make_stab<float>(q, rf_orth, false);
        auto s3 = make_stab<float>(q, qb_orth, false);
        RandLAPACK::RS<float, RNG> rs(q, *s1, p, q_, false, false);
        RandLAPACK::RF<float, RNG> rf(q, rs, *s2, false, false);
        RandLAPACK::QB<float, RNG> qb(q, rf, *s3, false, orth_check != 0);
        RandLAPACK::RSVD<float, RNG> rsvd(q, qb, b_sz);
        State st = load_state(state);
        *U = *S = *V = nullptr;
        int rc = rsvd.call(m, n, A, *k, tol, *U, *S, *V, st);
        store_state(st, state);
        if (qb_ret) *qb_ret = rsvd.qb_return;
        return rc;
    });
}

int rlhip_drv_cqrrpt_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float* R, int64_t ldr,
                         int64_t* J, float d_factor, int64_t nnz, float eps, uint32_t state[6],
                         const float* A_hat_in, float* A_hat_out, int64_t* rank_out, long* times_us, int qrcp) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::CQRRPT<float, RNG> alg(q, times_us != nullptr, eps);
        apply_cqrrpt_options(ctx, alg);
        alg.nnz = nnz;
        if (qrcp >= 16) { alg.orthogonalization = true; qrcp -= 16; }      // +16: CQRRPT::orthogonalization = true (rl_cqrrpt.hh:347-367)
        if (qrcp >= 0) {
            if (qrcp > 2) throw RandLAPACK::Error("qrcp must be 0 (hqrrp), 1 (bqrrp) or 2 (geqp3)");
            alg.qrcp = (RandLAPACK::CQRRPTSubroutines::QRCP)qrcp;
        }
        alg.sketch_override = A_hat_in;
        alg.sketch_export = A_hat_out;
        State st = load_state(state);
        int rc = alg.call(m, n, A, lda, R, ldr, J, d_factor, st);
        store_state(st, state);
        if (rank_out) *rank_out = alg.rank;
        if (times_us && alg.times.size() == 8)
            for (int i = 0; i < 8; ++i) times_us[i] = alg.times[i];
        return rc;
    });
}

int rlhip_drv_hqrrp_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, int64_t* jpvt, float* tau, int64_t nb_alg,
                        int64_t pp, int64_t panel_pivoting, int64_t qr_type, uint32_t state[6], float* G_out) {
    return guarded([&] {
        blas::Queue q(ctx);
        State st = load_state(state);
        int rc = (int)RandLAPACK::hqrrp<float, RNG>(m, n, A, lda, jpvt, tau, nb_alg, pp, panel_pivoting, qr_type, st, q, G_out);
        store_state(st, state);
        return rc;
    });
}

int rlhip_drv_bqrrp_f32(rlhip_ctx* ctx, int64_t m, int64_t n, float* A, int64_t lda, float d_factor, int64_t b_sz,
                        int64_t internal_nb, float tol, float* tau, int64_t* J, uint32_t state[6],
                        const float* A_sk_in, float* A_sk_out, int64_t* rank_out, long* times_us, int qrcp_wide, int qr_tall,
                        int apply_trans_q) {
    return guarded([&] {
        blas::Queue q(ctx);
        RandLAPACK::BQRRP<float, RNG> alg(q, times_us != nullptr, b_sz);
        using Sub = RandLAPACK::BQRRPSubroutines;
        if (qrcp_wide >= 0) { if (qrcp_wide > 1) throw RandLAPACK::Error("qrcp_wide must be 0 (luqr) or 1 (geqp3)"); alg.qrcp_wide = (Sub::QRCPWide)qrcp_wide; }
        if (qr_tall >= 16) { alg.rows_block_cyclic = true; qr_tall -= 16; if (qr_tall == 3) qr_tall = -1; }     // + 16: block-cyclic row layout of a sharded call (16 + 3: the object's default qr_tall)
        apply_bqrrp_options(ctx, alg);
        if (qr_tall >= 0) { if (qr_tall > 2) throw RandLAPACK::Error("qr_tall must be 0 (geqrt), 1 (cholqr) or 2 (geqrf)"); alg.qr_tall = (Sub::QRTall)qr_tall; }
        if (apply_trans_q >= 0) { if (apply_trans_q > 1) throw RandLAPACK::Error("apply_trans_q must be 0 (ormqr) or 1 (gemqrt)"); alg.apply_trans_q = (Sub::ApplyTransQ)apply_trans_q; }
        if (internal_nb > 0) alg.internal_nb = internal_nb;
        if (tol > 0) alg.tol = tol;
        alg.sketch_override = A_sk_in;
        alg.sketch_export = A_sk_out;
        State st = load_state(state);
        int rc = alg.call(m, n, A, lda, d_factor, tau, J, st);
        store_state(st, state);
        if (rank_out) *rank_out = alg.rank;
        if (times_us && alg.times.size() == 9)
            for (int i = 0; i < 9; ++i) times_us[i] = alg.times[i];
        return rc;
    });
}


int rlhip_drv_qr_linops_f64(rlhip_ctx* ctx, int alg, const rlhip_linop_desc* left, const rlhip_linop_desc* right, double* R, int64_t ldr,
                            int64_t block_size, double** Q_out, double d_factor, int64_t nnz, int use_dense_sketch, uint32_t state[6],
                            const double* A_hat_in, double* A_hat_out) {
    return drv_qr_linops<double>(ctx, alg, left, right, R, ldr, block_size, Q_out, d_factor, nnz, use_dense_sketch, state, A_hat_in, A_hat_out);
}
int rlhip_drv_qr_linops_f32(rlhip_ctx* ctx, int alg, const rlhip_linop_desc* left, const rlhip_linop_desc* right, float* R, int64_t ldr,
                            int64_t block_size, float** Q_out, float d_factor, int64_t nnz, int use_dense_sketch, uint32_t state[6],
                            const float* A_hat_in, float* A_hat_out) {
    return drv_qr_linops<float>(ctx, alg, left, right, R, ldr, block_size, Q_out, d_factor, nnz, use_dense_sketch, state, A_hat_in, A_hat_out);
}

static int abrik_linop_impl(rlhip_ctx* ctx, const rlhip_linop_desc* left, const rlhip_linop_desc* right, int64_t k, double tol,
                            int64_t max_krylov_iters, double** U, double** Sigma, double** V, uint32_t state[6], int64_t* triplets,
                            int64_t* iters, double* norm_R_end, int qr_exp, long* times13) {
    return guarded([&] {
        blas::Queue q(ctx);
        if (right) throw RandLAPACK::Error("ABRIK needs fro_nrm(): single dense / sparse operators only (composites have none, as in the reference)");
        if (!left || (left->kind != 0 && left->kind != 1)) throw RandLAPACK::Error("operator kind must be 0 (dense) or 1 (CSR)");
        auto run = [&](auto&& invoke) -> int {
            RandLAPACK::ABRIK<double, RNG> alg(q, false, times13 != nullptr, tol);
            if (qr_exp >= 0) {
                if (qr_exp > 1) throw RandLAPACK::Error("qr_exp must be 0 (geqrf_ungqr) or 1 (cqrrt)");
                alg.qr_exp = (RandLAPACK::ABRIKSubroutines::QR_explicit)qr_exp;
            }
            if (max_krylov_iters > 0) alg.max_krylov_iters = (int)std::min<int64_t>(max_krylov_iters, INT_MAX);
            State st = load_state(state);
            *U = nullptr; *Sigma = nullptr; *V = nullptr;
            int rc = invoke(alg, st);
            store_state(st, state);
            if (triplets) *triplets = alg.singular_triplets_found;
            if (iters) *iters = alg.num_krylov_iters;
            if (norm_R_end) *norm_R_end = alg.norm_R_end;
            if (times13) for (size_t i = 0; i < 13; ++i) times13[i] = (i < alg.times.size()) ? alg.times[i] : 0;
            return rc;
        };
        if (left->kind == 0) {
            auto A = make_dense<double>(q, *left);
            return run([&](auto& alg, State& st) { return alg.call(*A, k, *U, *V, *Sigma, st); });                 // LinearOperator overload (:164)
        }
        // CSR operator: the sparse-matrix overload (rl_abrik.hh:146-162) over a RandBLAS-style view of the caller's arrays
        RandBLAS::sparse_data::CSRMatrix<double> M(left->rows, left->cols, left->nnz, (const double*)left->vals, left->rowptr, left->colidx);
        return run([&](auto& alg, State& st) { return alg.call(left->rows, left->cols, M, k, *U, *V, *Sigma, st); });
    });
}

int rlhip_drv_abrik_linop_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, const rlhip_linop_desc* right, int64_t k, double tol,
                              int64_t max_krylov_iters, double** U, double** Sigma, double** V, uint32_t state[6], int64_t* triplets,
                              int64_t* iters, double* norm_R_end, int qr_exp) {
    return abrik_linop_impl(ctx, left, right, k, tol, max_krylov_iters, U, Sigma, V, state, triplets, iters, norm_R_end, qr_exp, nullptr);
}

int rlhip_drv_abrik_linop_timed_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, int64_t k, double tol, int64_t max_krylov_iters, double** U,
                                    double** Sigma, double** V, uint32_t state[6], int64_t* triplets, int64_t* iters, double* norm_R_end,
                                    int qr_exp, long times[13]) {
    if (!times) return -14;
    return abrik_linop_impl(ctx, left, nullptr, k, tol, max_krylov_iters, U, Sigma, V, state, triplets, iters, norm_R_end, qr_exp, times);
}

int rlhip_linop_apply_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, const rlhip_linop_desc* right, char side, char trans, int64_t m,
                          int64_t n, int64_t k, double alpha, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
    return guarded([&] {
        blas::Queue q(ctx);
        if ((side != 'L' && side != 'R') || (trans != 'N' && trans != 'T')) throw RandLAPACK::Error("side must be L or R, trans N or T");
        return with_operator<double>(q, left, right, [&](auto& A) -> int {
            A(side == 'L' ? RandLAPACK::Side::Left : RandLAPACK::Side::Right, RandLAPACK::Layout::ColMajor,
              trans == 'N' ? RandLAPACK::Op::NoTrans : RandLAPACK::Op::Trans, RandLAPACK::Op::NoTrans, m, n, k, alpha, B, ldb, beta, C, ldc);
            return 0;
        });
    });
}


// the same product with a BLOCK VIEW of the operator: how 0 row_block(view[0], view[2]), 1 col_block(view[1], view[3]), 2 submatrix(view[0..3])
int rlhip_linop_apply_view_f64(rlhip_ctx* ctx, const rlhip_linop_desc* left, const rlhip_linop_desc* right, int how, const int64_t view[4], char side,
                               char trans, int64_t m, int64_t n, int64_t k, double alpha, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
    return guarded([&] {
        blas::Queue q(ctx);
        if ((side != 'L' && side != 'R') || (trans != 'N' && trans != 'T')) throw RandLAPACK::Error("side must be L or R, trans N or T");
        if (how < 0 || how > 2 || !view) throw RandLAPACK::Error("how must be 0 (row_block), 1 (col_block) or 2 (submatrix)");
        return with_operator<double>(q, left, right, [&](auto& A) -> int {
            auto apply = [&](auto&& V) {
                V(side == 'L' ? RandLAPACK::Side::Left : RandLAPACK::Side::Right, RandLAPACK::Layout::ColMajor,
                  trans == 'N' ? RandLAPACK::Op::NoTrans : RandLAPACK::Op::Trans, RandLAPACK::Op::NoTrans, m, n, k, alpha, B, ldb, beta, C, ldc);
                q.sync();                   // (a view may own device arrays that go away with it)
            };
            if (how == 0) apply(A.row_block(view[0], view[2]));
            else if (how == 1) apply(A.col_block(view[1], view[3]));
            else apply(A.submatrix(view[0], view[1], view[2], view[3]));
            return 0;
        });
    });
}

// linops::RegExplicitSymLinOp (rl_sym_linops.hh:134-233): C = alpha (A + mu_i I) B + beta C, A given by its UPPER triangle; regs on the HOST
int rlhip_regsym_apply_f64(rlhip_ctx* ctx, int64_t dim, const double* A, int64_t lda, const double* regs_host, int64_t num_ops, int eval_includes_reg,
                           int64_t n, double alpha, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
    return guarded([&] {
        blas::Queue q(ctx);
        lo::RegExplicitSymLinOp<double> op(dim, A, lda, regs_host, num_ops, q);
        op.set_eval_includes_reg(eval_includes_reg != 0);
        op(RandLAPACK::Layout::ColMajor, n, alpha, B, ldb, beta, C, ldc);
        q.sync();
        return 0;
    });
}

int rlhip_drv_mat_gen_f64(rlhip_ctx* ctx, int type, int64_t m, int64_t n, int64_t rank, double cond_num, double scaling, double exponent,
                          int diag, double theta, double perturb, double frac_spectrum_one, int check_true_rank, double* A,
                          uint32_t state[6], int64_t* rank_out) {
    return drv_mat_gen<double>(ctx, type, m, n, rank, cond_num, scaling, exponent, diag, theta, perturb, frac_spectrum_one, check_true_rank, A, state, rank_out);
}
int rlhip_drv_mat_gen_f32(rlhip_ctx* ctx, int type, int64_t m, int64_t n, int64_t rank, float cond_num, float scaling, float exponent,
                          int diag, float theta, float perturb, float frac_spectrum_one, int check_true_rank, float* A,
                          uint32_t state[6], int64_t* rank_out) {
    return drv_mat_gen<float>(ctx, type, m, n, rank, cond_num, scaling, exponent, diag, theta, perturb, frac_spectrum_one, check_true_rank, A, state, rank_out);
}


int rlhip_drv_revd2_f64(rlhip_ctx* ctx, char uplo, int64_t m, const double* A, int64_t* k, double tol, int64_t syps_passes,
                        int64_t passes_per_stab, int error_est_p, int orth_kind, double** V, double** eigvals, uint32_t state[6],
                        double* err_out) {
    return guarded([&] {
        blas::Queue q(ctx);
        if (uplo != 'U' && uplo != 'L') throw RandLAPACK::Error("uplo must be U or L");
        using SYPS_t = RandLAPACK::SYPS<double, RNG>;
        using Orth_t = RandLAPACK::Stabilization<double>;
        using SYRF_t = RandLAPACK::SYRF<SYPS_t, Orth_t>;
        SYPS_t syps(q, syps_passes, passes_per_stab, false, false);
        auto orth = make_stab<double>(q, orth_kind, false);
        SYRF_t syrf(syps, *orth, false, false);
        RandLAPACK::REVD2<SYRF_t> revd2(syrf, error_est_p, false);
        State st = load_state(state);
        *V = nullptr; *eigvals = nullptr;
        int rc = revd2.call(uplo == 'U' ? RandLAPACK::Uplo::Upper : RandLAPACK::Uplo::Lower, m, A, *k, tol, *V, *eigvals, st);
        store_state(st, state);
        if (err_out) *err_out = revd2.last_err;
        return rc;
    });
}

int rlhip_drv_syrf_f64(rlhip_ctx* ctx, char uplo, int64_t m, const double* A, int64_t k, int64_t syps_passes, int64_t passes_per_stab,
                       int orth_kind, double* Q, uint32_t state[6]) {
    return guarded([&] {
        blas::Queue q(ctx);
        if (uplo != 'U' && uplo != 'L') throw RandLAPACK::Error("uplo must be U or L");
        using SYPS_t = RandLAPACK::SYPS<double, RNG>;
        using Orth_t = RandLAPACK::Stabilization<double>;
        SYPS_t syps(q, syps_passes, passes_per_stab, false, false);
        auto orth = make_stab<double>(q, orth_kind, false);
        RandLAPACK::SYRF<SYPS_t, Orth_t> syrf(syps, *orth, false, false);
        State st = load_state(state);
        int rc = syrf.call(uplo == 'U' ? RandLAPACK::Uplo::Upper : RandLAPACK::Uplo::Lower, m, A, k, Q, st, nullptr);
        store_state(st, state);
        return rc;
    });
}

}  // extern "C"
