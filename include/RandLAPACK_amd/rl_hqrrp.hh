// hqrrp (reference: RandLAPACK/drivers/rl_hqrrp.hh:812-1197): Householder QR with randomized pivoting (Martinsson,
// Quintana-Orti, Heavner, van de Geijn), GEQP3-compatible output: R above the diagonal, Householder vectors below,
// tau, 1-based jpvt.  Same free-function signature as the reference plus the queue; every buffer is a DEVICE buffer.
//
// How the reference's pieces map onto the device:
//   G = Uniform(-1,1) ((nb+pp) x m), Y = G A                -> fill_dense + MFMA GEMM                       (:929-936)
//   per block of nb columns (b = actual width):
//     QRP of the current sketch Y_R, b steps, permutations
//       carried to A_R and Y_R            (NoFLA_QRPmod_WY_unb_var4 0,1,b :1040-1062)  -> qrp_partial on a copy of Y_R +
//                                                                                          col_swap of A_R, Y_R, jpvt_R
//     panel factorization of [A11; A21] + T                   (:1084-1094)
//       qr_type 0 & panel_pivoting   pivoted Householder QR   -> qrp_partial (all b steps) + col_swap of A01, Y1, jpvt + larft
//       qr_type 0 | 1, no pivoting   Householder QR           -> geqrf + larft
//       qr_type 2                    CholQR + orhr_col        -> syrk/potrf/trsm + orhr_col + sign fix (CHOLQR_mod_WY :466-512)
//     [A12; A22] <- Q^T [A12; A22]   (larfb, :1108-1118)      -> gemqrt (one compact-WY block, MFMA)
//     Y2 -= (G1 - (G1 U11 + G2 U21) T U11^T) R12 ; G_R <- G_R Q (NoFLA_Downdate_Y :210-295)
//                                                             -> G_R <- G_R Q first (larfb from the right); the bracket IS
//                                                                the updated G1, so Y2 -= G1 R12 is one more GEMM.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "rl_exceptions.hh"
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_randblas.hh"
#include "rl_util.hh"
#include "rl_sharded_panel.hh"

namespace RandLAPACK {

inline void detail_hq_axpy(int64_t n, double a, const double* x, double* y, blas::Queue& q) { blas::check(rlhip_axpby_f64(q.ctx(), n, a, x, 1.0, y), "axpby"); }
inline void detail_hq_axpy(int64_t n, float a, const float* x, float* y, blas::Queue& q) { blas::check(rlhip_axpby_f32(q.ctx(), n, a, x, 1.0f, y), "axpby"); }

/// hqrrp on a ROW-SHARDED matrix (one process per GPU; this rank holds a contiguous row block of m_loc rows; not in the reference, which has
/// no distributed code).  Replicated on every rank: the sketch Y = G A (summed once), its pivoted QR and down-dates, jpvt, tau, the b x b
/// factors.  Sharded by rows: A (reflectors and trailing matrix) and, by columns, the sketching matrix G (rank g owns the columns of its rows).
/// Exchanges per block of nb_alg columns: the panel's triangles (TSQR: one stack; two with pivoted panels, whose pivots come from the QRCP of
/// the unpivoted panel's R factor -- the tall-panel order of the single-device code), the top block of the orthonormal panel, W = V^T C for the
/// compact-WY apply, G V for the update of G, and G1 R12 for the down-date of Y.  Same pivots as the single-device factorization, factors to
/// rounding (tests/test_gpu_sharded.py).  Returns 0; 1 when a Cholesky-QR panel (qr_type 2, unpivoted) broke down.
template <typename T, typename RNG>
int64_t hqrrp_sharded(int64_t m_loc, int64_t n_A, T* buff_A, int64_t ldim_A, int64_t* buff_jpvt, T* buff_tau, int64_t nb_alg, int64_t pp,
                      int64_t panel_pivoting, int64_t qr_type, RandBLAS::RNGState<RNG>& state, blas::Queue& q, T* G_export = nullptr) {
    detail::ShardRows L;
    L.init(q, m_loc);
    const int64_t m_A = L.m_glob, mn_A = std::min(m_A, n_A);
    if (mn_A == 0) return 0;
    const int64_t m_Y = nb_alg + pp, ldY = m_Y;
    blas::Scratch ws(q);
    T* Y = ws.alloc<T>(m_Y * n_A);
    T* Vc = ws.alloc<T>(m_Y * n_A);
    T* G = ws.alloc<T>(m_Y * std::max<int64_t>(m_loc, 1));               // my columns of the global (nb + pp) x m sketching matrix
    T* T1 = ws.alloc<T>(nb_alg * nb_alg);
    T* Rp = ws.alloc<T>(nb_alg * nb_alg);
    T* Q1 = ws.alloc<T>(nb_alg * nb_alg);
    T* tau_scr = ws.alloc<T>(n_A);
    int64_t* Jloc = ws.alloc<int64_t>(n_A);
    {
        std::vector<int64_t> iota_((size_t)n_A);
        for (int64_t i = 0; i < n_A; ++i) iota_[(size_t)i] = i + 1;
        blas::copy_to_device(n_A, iota_.data(), buff_jpvt, q);
    }
    {   // G: ONE global Uniform(-1, 1) operator (rl_hqrrp.hh:929-930); this rank draws its column block (stream positions m_Y * row0 ...)
        RandBLAS::DenseDist Dall(m_Y * m_A, 1, RandBLAS::ScalarDist::Uniform);
        auto st_in = state;
        state = RandBLAS::fill_dense_rows(Dall, 0, 0, G, st_in, q);
        if (m_loc > 0) RandBLAS::fill_dense_rows(Dall, m_Y * L.row0, m_Y * m_loc, G, st_in, q);
        if (G_export && m_loc > 0) blas::device_copy_vector(m_Y * m_loc, G, G_export, q);
        if (m_loc > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m_Y, n_A, m_loc, (T)1, G, m_Y, buff_A, ldim_A, (T)0, Y, ldY, q);
        else lapack::laset(MatrixType::General, m_Y, n_A, (T)0, (T)0, Y, ldY, q);
        q.allreduce_sum(Y, m_Y * n_A);
    }
    for (int64_t j = 0; j < mn_A; j += nb_alg) {
        const int64_t b = std::min(nb_alg, std::min(n_A - j, m_A - j));
        const bool last_iter = (j + nb_alg >= m_A) || (j + nb_alg >= n_A);
        const int64_t n_R = n_A - j, n2 = std::max<int64_t>(0, n_A - j - b);
        const int64_t act_loc = L.local_from(j), loc_rows = m_loc - act_loc;
        T* A_R = &buff_A[j * ldim_A];
        T* Y_R = &Y[j * ldY];
        if (!last_iter) {                                                    // pivots of the block from the (replicated) sketch
            blas::LocalOnly replicated(q);
            lapack::lacpy(MatrixType::General, m_Y, n_R, Y_R, ldY, &Vc[j * ldY], ldY, q);
            lapack::qrp_partial(m_Y, n_R, b, &Vc[j * ldY], ldY, Jloc, tau_scr, q);
            if (m_loc > 0) util::col_swap(m_loc, n_R, n_R, A_R, ldim_A, Jloc, q);
            util::col_swap(m_Y, n_R, n_R, Y_R, ldY, Jloc, q);
            util::col_swap(n_R, n_R, &buff_jpvt[j], Jloc, q);
        }
        T* A_work = (loc_rows > 0) ? &buff_A[act_loc + j * ldim_A] : nullptr;
        if (qr_type == 2 && !panel_pivoting) {                               // Cholesky-QR panel: one b x b Gram all-reduce
            lapack::laset(MatrixType::General, nb_alg, nb_alg, (T)0, (T)0, Rp, nb_alg, q);
            if (loc_rows > 0) blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, b, loc_rows, (T)1, A_work, ldim_A, (T)0, Rp, nb_alg, q);
            q.allreduce_sum(Rp, nb_alg * nb_alg);
            if (lapack::potrf(Uplo::Upper, b, Rp, nb_alg, q)) return 1;     // (replicated Gram matrix: every rank takes the same exit)
            if (loc_rows > 0) blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, loc_rows, b, (T)1, Rp, nb_alg, A_work, ldim_A, q);
        } else {
            if (panel_pivoting) {
                // A = Q R has the same pivoted QR as its b x b factor R: the panel's pivots come from the QRCP of the R factor of its
                // unpivoted TSQR (the tall-panel order of the single-device code, here for every panel: a pivoted sweep does not shard)
                blas::Scratch w2(q);
                T* Pc = w2.alloc<T>(std::max<int64_t>(loc_rows, 1) * b);
                if (loc_rows > 0) lapack::lacpy(MatrixType::General, loc_rows, b, A_work, ldim_A, Pc, loc_rows, q);
                detail::tsqr_r(q, loc_rows, b, Pc, std::max<int64_t>(loc_rows, 1), Rp, nb_alg);
                {
                    blas::LocalOnly replicated(q);
                    lapack::qrp_partial(b, b, b, Rp, nb_alg, Jloc, tau_scr, q);
                }
                if (m_loc > 0) util::col_swap(m_loc, b, b, A_R, ldim_A, Jloc, q);     // the panel's columns, all my rows (A01 above it, the panel itself)
                util::col_swap(m_Y, b, b, Y_R, ldY, Jloc, q);
                util::col_swap(b, b, &buff_jpvt[j], Jloc, q);
            }
            detail::tsqr(q, loc_rows, b, A_work, ldim_A, Rp, nb_alg);
        }
        blas::Scratch keep(q);
        lapack::laset(MatrixType::General, nb_alg, nb_alg, (T)0, (T)0, T1, nb_alg, q);
        auto S = detail::reconstruct(q, keep, L, buff_A, ldim_A, j, j, b, Rp, nb_alg, T1, nb_alg, &buff_tau[j], Q1);
        detail::apply_qt(q, S, T1, nb_alg, buff_A, ldim_A, j + b, n2);                                       // :1108-1118
        if (!last_iter) {                                                                                   // :1135-1145, NoFLA_Downdate_Y
            // G_R <- G_R Q = G_R - (G_R V) T V^T: G_R V sums over the ranks (my columns of G_R times my rows of V)
            blas::Scratch w3(q);
            T* GV = w3.alloc<T>(m_Y * b);
            T* Z = w3.alloc<T>(m_Y * b);
            const int64_t vr = S.vrows();
            T* G_act = G + act_loc * m_Y;
            if (vr > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m_Y, b, vr, (T)1, G_act, m_Y, S.Vexp, S.ldv, (T)0, GV, m_Y, q);
            else lapack::laset(MatrixType::General, m_Y, b, (T)0, (T)0, GV, m_Y, q);
            q.allreduce_sum(GV, m_Y * b);
            blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m_Y, b, b, (T)1, GV, m_Y, T1, nb_alg, (T)0, Z, m_Y, q);
            if (vr > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::Trans, m_Y, vr, b, (T)-1, Z, m_Y, S.Vexp, S.ldv, (T)1, G_act, m_Y, q);
            // Y2 -= G1 R12: G1 = the updated columns of G of the top block's rows, R12 = those rows of the updated trailing matrix -- both
            // live on the top block's owners, whose products are summed
            if (n2 > 0) {
                T* D = w3.alloc<T>(m_Y * n2);
                if (S.tcnt > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m_Y, n2, S.tcnt, (T)1, G_act, m_Y, &buff_A[act_loc + (j + b) * ldim_A], ldim_A, (T)0, D, m_Y, q);
                else lapack::laset(MatrixType::General, m_Y, n2, (T)0, (T)0, D, m_Y, q);
                q.allreduce_sum(D, m_Y * n2);
                detail_hq_axpy(m_Y * n2, (T)-1, D, &Y[(j + b) * ldY], q);
            }
        }
    }
    return 0;
}

/// returns 0; 1 if a CholQR panel (qr_type == 2) broke down.  `G_export`, when not null, receives the (nb_alg+pp) x m
/// sketching matrix before it is updated (tests share it with the CPU path, cf. test_bqrrp_gpu.cu:91-110).
/// `timing` (the reference's last argument, rl_hqrrp.hh:815,1144-1164): when not null, *timing is (re)allocated to 27 entries
/// (microseconds; the reference documents 26 and its benchmark prints 27):
///   [0..8]  preallocation, sketching, downdating, qrcp, qr, updating_A, updating_Sketch, other, total
///   [9..17] the sketch-QRCP kernel's own breakdown, [18..26] the panel-QR kernel's -- on the device each is ONE kernel, so only
///   the total slots (17 and 26) are filled, the per-step slots stay 0.
/// Timing synchronises the queue at every stamp.
/// On a ROW-SHARDED queue (q.world() > 1) the call dispatches to hqrrp_sharded: m_A is then THIS RANK's row count (contiguous row blocks in
/// rank order), ldim_A >= m_A, buff_tau must hold min(global rows, n_A) entries and buff_jpvt n_A (both replicated on every rank), and
/// *timing receives 27 entries of which only `other` [7] and `total` [8] are filled.
template <typename T, typename RNG>
int64_t hqrrp(int64_t m_A, int64_t n_A, T* buff_A, int64_t ldim_A, int64_t* buff_jpvt, T* buff_tau, int64_t nb_alg, int64_t pp,
              int64_t panel_pivoting, int64_t qr_type, RandBLAS::RNGState<RNG>& state, blas::Queue& q, T* G_export = nullptr,
              T** timing = nullptr) {
    using hq_clk = std::chrono::steady_clock;
    long t_prealloc = 0, t_sketch = 0, t_down = 0, t_qrcp = 0, t_qr = 0, t_updA = 0, t_updS = 0;
    auto stamp = [&]() { if (timing) q.sync(); return hq_clk::now(); };
    auto us = [](hq_clk::time_point a, hq_clk::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    const auto t_begin = stamp();
    randlapack_require(m_A >= 0) << "hqrrp: m_A is < 0";                                                   // :871-877
    randlapack_require(n_A >= 0) << "hqrrp: n_A is < 0";
    randlapack_require(ldim_A >= std::max<int64_t>(1, m_A)) << "hqrrp: ldim_A is < max(1, m_A)";
    randlapack_require(nb_alg > 0 && pp >= 0) << "hqrrp: nb_alg must be > 0 and pp >= 0";
    if (q.world() > 1) {                      // row-sharded queue: m_A is this rank's row count (tau then has min(global rows, n) entries)
        const int64_t rc = hqrrp_sharded<T, RNG>(m_A, n_A, buff_A, ldim_A, buff_jpvt, buff_tau, nb_alg, pp, panel_pivoting, qr_type, state, q, G_export);
        if (timing) {                         // the sharded loop is not instrumented per stage: 27 entries, zeros except `other` and `total`
            const long total = us(t_begin, stamp());
            T* tt = (T*)std::realloc(*timing, 27 * sizeof(T));
            if (tt) { for (int i = 0; i < 27; ++i) tt[i] = (T)0; tt[7] = (T)total; tt[8] = (T)total; *timing = tt; }
        }
        return rc;
    }
    const int64_t mn_A = std::min(m_A, n_A);
    if (mn_A == 0) return 0;
    const int64_t m_Y = nb_alg + pp, n_Y = n_A, ldim_Y = m_Y, ldim_V = m_Y, m_G = nb_alg + pp, n_G = m_A, ldim_G = m_G;
    blas::Scratch ws(q);
    T* buff_Y = ws.alloc<T>(m_Y * n_Y);
    T* buff_V = ws.alloc<T>(m_Y * n_Y);
    T* buff_G = ws.alloc<T>(m_G * n_G);
    T* T1 = ws.alloc<T>(nb_alg * nb_alg);
    T* buff_R = ws.alloc<T>(nb_alg * nb_alg);
    T* buff_D = ws.alloc<T>(nb_alg);
    T* tau_scr = ws.alloc<T>(n_A);
    int64_t* Jloc = ws.alloc<int64_t>(n_A);
    {   // jpvt = 1..n  (:921)
        std::vector<int64_t> iota_((size_t)n_A);
        for (int64_t i = 0; i < n_A; ++i) iota_[(size_t)i] = i + 1;
        blas::copy_to_device(n_A, iota_.data(), buff_jpvt, q);
    }
    auto t_a = stamp();
    t_prealloc = us(t_begin, t_a);
    RandBLAS::DenseDist D(nb_alg + pp, m_A, RandBLAS::ScalarDist::Uniform);                                 // :929-930
    state = RandBLAS::fill_dense(D, buff_G, state, q);
    if (G_export) blas::device_copy_vector(m_G * n_G, buff_G, G_export, q);
    blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m_Y, n_Y, m_A, (T)1, buff_G, ldim_G, buff_A, ldim_A, (T)0, buff_Y, ldim_Y, q);
    t_sketch = us(t_a, stamp());

    for (int64_t j = 0; j < mn_A; j += nb_alg) {
        const int64_t b = std::min(nb_alg, std::min(n_A - j, m_A - j));
        const bool last_iter = (j + nb_alg >= m_A) || (j + nb_alg >= n_A);
        const int64_t n_VR = n_A - j;
        T* buff_VR = &buff_V[j * ldim_V];
        T* buff_YR = &buff_Y[j * ldim_Y];
        int64_t* buff_pB = &buff_jpvt[j];
        T* buff_sB = &buff_tau[j];
        T* buff_AR = &buff_A[j * ldim_A];
        const int64_t m_AB1 = m_A - j;
        T* buff_AB1 = &buff_A[j + j * ldim_A];
        T* buff_A01 = &buff_A[j * ldim_A];
        T* buff_Y1 = &buff_Y[j * ldim_Y];
        T* buff_A12 = &buff_A[j + std::min(n_A - 1, j + b) * ldim_A];
        const int64_t n_A12 = std::max<int64_t>(0, n_A - j - b);
        T* buff_Y2 = &buff_Y[std::min(n_Y - 1, j + b) * ldim_Y];
        T* buff_G1 = &buff_G[j * ldim_G];

        auto t0 = stamp();
        if (!last_iter) {                                                                                   // :1040-1062
            lapack::lacpy(MatrixType::General, m_Y, n_VR, buff_YR, ldim_Y, buff_VR, ldim_V, q);
            lapack::qrp_partial(m_Y, n_VR, b, buff_VR, ldim_V, Jloc, tau_scr, q);
            util::col_swap(m_A, n_VR, n_VR, buff_AR, ldim_A, Jloc, q);
            util::col_swap(m_Y, n_VR, n_VR, buff_YR, ldim_Y, Jloc, q);
            util::col_swap(n_VR, n_VR, buff_pB, Jloc, q);
        }
        auto t1 = stamp();
        t_qrcp += us(t0, t1);
        // ---- panel [A11; A21] (m_AB1 x b) and its T                                                       :1084-1094
        if (qr_type == 2 && !panel_pivoting) {                                                              // CHOLQR_mod_WY
            lapack::laset(MatrixType::General, nb_alg, nb_alg, (T)0, (T)0, buff_R, nb_alg, q);
            blas::syrk(Layout::ColMajor, Uplo::Upper, Op::Trans, b, m_AB1, (T)1, buff_AB1, ldim_A, (T)0, buff_R, nb_alg, q);
            if (lapack::potrf(Uplo::Upper, b, buff_R, nb_alg, q)) return 1;
            blas::trsm(Layout::ColMajor, Side::Right, Uplo::Upper, Op::NoTrans, Diag::NonUnit, m_AB1, b, (T)1, buff_R, nb_alg, buff_AB1, ldim_A, q);
            lapack::orhr_col(m_AB1, b, b, buff_AB1, ldim_A, T1, nb_alg, buff_D, q);
            lapack::row_sign(b, buff_R, nb_alg, buff_D, q);
            lapack::lacpy(MatrixType::Upper, b, b, buff_R, nb_alg, buff_AB1, ldim_A, q);
            lapack::tau_from_t(b, b, T1, nb_alg, buff_sB, q);
        } else if (panel_pivoting) {
            // Tall panels: the column-owning pivoted kernel keeps only b / 8 workgroups busy on a panel of thousands of rows (16384 x 256:
            // 54 ms, 88 % of hqrrp at 16384^2).  A = Q R has the same pivoted QR as its b x b factor R (equal partial column norms at every
            // step), so the pivots come from the QRCP of R -- R from the row-parallel unpivoted geqrf of a copy -- and the reflectors from
            // the unpivoted geqrf of the permuted panel: identical to the pivoted sweep up to rounding, ties aside.
            const bool tall_split = rlhip_get_option(q.ctx(), RLHIP_OPT_HQRRP_TALL_PANEL) != 0;      // (default on; 0: the reference's single pivoted sweep)
            if (tall_split && b >= 16 && m_AB1 >= 8 * b) {
                blas::Scratch w2(q);
                T* P = w2.alloc<T>(m_AB1 * b);
                T* Rs = w2.alloc<T>(b * b);
                lapack::lacpy(MatrixType::General, m_AB1, b, buff_AB1, ldim_A, P, m_AB1, q);
                lapack::geqrf(m_AB1, b, P, m_AB1, tau_scr, q);
                lapack::laset(MatrixType::General, b, b, (T)0, (T)0, Rs, b, q);
                lapack::lacpy(MatrixType::Upper, b, b, P, m_AB1, Rs, b, q);
                lapack::qrp_partial(b, b, b, Rs, b, Jloc, tau_scr, q);
                util::col_swap(m_AB1, b, b, buff_AB1, ldim_A, Jloc, q);
                lapack::geqrf(m_AB1, b, buff_AB1, ldim_A, buff_sB, q);
            } else {
                lapack::qrp_partial(m_AB1, b, std::min(m_AB1, b), buff_AB1, ldim_A, Jloc, buff_sB, q);
            }
            if (j > 0) util::col_swap(j, b, b, buff_A01, ldim_A, Jloc, q);
            util::col_swap(m_Y, b, b, buff_Y1, ldim_Y, Jloc, q);
            util::col_swap(b, b, buff_pB, Jloc, q);
            lapack::larft(m_AB1, std::min(m_AB1, b), buff_AB1, ldim_A, buff_sB, T1, nb_alg, q);
        } else {                                                                                            // GEQRF_mod_WY / unpivoted var4
            lapack::geqrf(m_AB1, b, buff_AB1, ldim_A, buff_sB, q);
            lapack::larft(m_AB1, std::min(m_AB1, b), buff_AB1, ldim_A, buff_sB, T1, nb_alg, q);
        }
        const int64_t kref = std::min(m_AB1, b);
        auto t2 = stamp();
        t_qr += us(t1, t2);
        if (j + b < n_A)                                                                                    // :1108-1118
            lapack::gemqrt(Side::Left, Op::Trans, m_AB1, n_A12, kref, kref, buff_AB1, ldim_A, T1, nb_alg, buff_A12, ldim_A, q);
        auto t3 = stamp();
        t_updA += us(t2, t3);
        if (!last_iter) {                                                                                   // :1135-1145
            lapack::gemqrt(Side::Right, Op::NoTrans, m_G, n_G - j, kref, kref, buff_AB1, ldim_A, T1, nb_alg, buff_G1, ldim_G, q);
            auto t4 = stamp();
            t_updS += us(t3, t4);
            const int64_t n_Y2 = std::max<int64_t>(0, n_Y - j - b);
            if (n_Y2 > 0)
                blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, m_Y, n_Y2, b, (T)-1, buff_G1, ldim_G, buff_A12, ldim_A, (T)1, buff_Y2, ldim_Y, q);
            t_down += us(t4, stamp());
        }
    }
    if (timing) {                                                                                           // :1144-1164
        const long total = us(t_begin, stamp());
        T* tt = (T*)std::realloc(*timing, 27 * sizeof(T));
        if (tt) {
            for (int i = 0; i < 27; ++i) tt[i] = (T)0;
            tt[0] = (T)t_prealloc; tt[1] = (T)t_sketch; tt[2] = (T)t_down; tt[3] = (T)t_qrcp; tt[4] = (T)t_qr; tt[5] = (T)t_updA; tt[6] = (T)t_updS;
            tt[7] = (T)(total - (t_prealloc + t_sketch + t_down + t_qrcp + t_qr + t_updA + t_updS)); tt[8] = (T)total;
            tt[17] = (T)t_qrcp; tt[26] = (T)t_qr;
            *timing = tt;
        }
    }
    return 0;
}

/// The reference's exact signature (rl_hqrrp.hh:811-815: no queue, `timing` last): the process-wide default queue.
template <typename T, typename RNG>
int64_t hqrrp(int64_t m_A, int64_t n_A, T* buff_A, int64_t ldim_A, int64_t* buff_jpvt, T* buff_tau, int64_t nb_alg, int64_t pp,
              int64_t panel_pivoting, int64_t qr_type, RandBLAS::RNGState<RNG>& state, T** timing = nullptr) {
    return hqrrp<T, RNG>(m_A, n_A, buff_A, ldim_A, buff_jpvt, buff_tau, nb_alg, pp, panel_pivoting, qr_type, state, blas::default_queue(), (T*)nullptr, timing);
}

}  // namespace RandLAPACK
