// Panels of a blocked Householder QR whose ROWS are sharded over the ranks of a queue (one process per GPU; contiguous row blocks: this rank
// holds global rows [row0, row0 + m_loc) of the matrix).  Used by the row-sharded hqrrp (rl_hqrrp.hh); BQRRP::call_sharded
// (rl_bqrrp.hh) runs the same three steps inline for its two row layouts.
//
//   tsqr            Householder QR of a sharded tall panel in ONE exchange: local A_g = Q_g R_g, every rank's triangle into its slot of a
//                   (P b) x b stack (zero rows where a rank has fewer than b rows or none), one all-reduce (disjoint slots: an all-gather, bit
//                   for bit), the stack = Qt R on every rank (same bits in, same kernels: same bits out), panel <- Q_g Qt_g.
//   reconstruct     the orthonormal panel as the reflectors of the WHOLE panel (lapack::orhr_col on [top block gathered from its owners;
//                   my rows below it]): V into A, the sign-fixed R11 onto its owners' rows, T and tau replicated -- LAPACK's geqrf
//                   representation, unique up to rounding, so the sharded factorization equals the single-device one.
//   apply           C <- Q^T C on the trailing columns: W = sum over the ranks of V_g^T C_g (one all-reduce), C_g -= V_g (T^T W).
#pragma once
#include <algorithm>
#include <cstdint>
#include "rl_blaspp.hh"
#include "rl_lapackpp.hh"
#include "rl_util.hh"

namespace RandLAPACK::detail {

/// this rank's rows of a row-sharded matrix (contiguous block) and the bookkeeping of one panel that starts at global row `g0`
struct ShardRows {
    int64_t m_loc = 0, m_glob = 0, row0 = 0;
    void init(blas::Queue& q, int64_t m_local) { m_loc = m_local; q.shard_extent(m_local, m_glob, row0); }
    /// local index of my first row with global index >= g (m_loc when there is none)
    int64_t local_from(int64_t g) const { return std::min(m_loc, std::max<int64_t>(0, g - row0)); }
};

/// R only: the b x b triangle of the Householder QR of the sharded panel P_g (loc_rows x b, ld ldp; DESTROYED), replicated into R (ld ldr,
/// zero below the diagonal).  One exchange.
template <typename T>
void tsqr_r(blas::Queue& q, int64_t loc_rows, int64_t b, T* Pg, int64_t ldp, T* R, int64_t ldr) {
    const int64_t P = q.world(), me = q.rank(), kr = std::min(loc_rows, b);
    blas::Scratch w(q);
    T* stack = w.alloc<T>(P * b * b);
    T* tau_l = w.alloc<T>(std::max<int64_t>(b, 1));
    lapack::laset(MatrixType::General, P * b, b, (T)0, (T)0, stack, P * b, q);
    if (loc_rows > 0) {
        lapack::geqrf(loc_rows, b, Pg, ldp, tau_l, q);
        lapack::lacpy(MatrixType::Upper, kr, b, Pg, ldp, stack + me * b, P * b, q);
    }
    q.allreduce_sum(stack, P * b * b);
    lapack::geqrf(P * b, b, stack, P * b, tau_l, q);
    lapack::laset(MatrixType::General, b, b, (T)0, (T)0, R, ldr, q);
    lapack::lacpy(MatrixType::Upper, b, b, stack, P * b, R, ldr, q);
}

/// panel A_g (loc_rows x b, ld lda) <- its rows of the orthonormal factor; R (ld ldr) <- the replicated triangle.  One exchange.
template <typename T>
void tsqr(blas::Queue& q, int64_t loc_rows, int64_t b, T* Ag, int64_t lda, T* R, int64_t ldr) {
    const int64_t P = q.world(), me = q.rank(), kr = std::min(loc_rows, b);
    blas::Scratch w(q);
    T* stack = w.alloc<T>(P * b * b);
    T* tau_l = w.alloc<T>(std::max<int64_t>(b, 1));
    T* Qg = w.alloc<T>(std::max<int64_t>(loc_rows, 1) * b);
    lapack::laset(MatrixType::General, P * b, b, (T)0, (T)0, stack, P * b, q);
    if (loc_rows > 0) {
        lapack::geqrf(loc_rows, b, Ag, lda, tau_l, q);
        lapack::lacpy(MatrixType::Upper, kr, b, Ag, lda, stack + me * b, P * b, q);
        lapack::ungqr(loc_rows, kr, kr, Ag, lda, tau_l, q);                      // Q_g: loc_rows x kr
        lapack::lacpy(MatrixType::General, loc_rows, kr, Ag, lda, Qg, loc_rows, q);
    }
    q.allreduce_sum(stack, P * b * b);
    lapack::geqrf(P * b, b, stack, P * b, tau_l, q);
    lapack::laset(MatrixType::General, b, b, (T)0, (T)0, R, ldr, q);
    lapack::lacpy(MatrixType::Upper, b, b, stack, P * b, R, ldr, q);
    lapack::ungqr(P * b, b, b, stack, P * b, tau_l, q);                          // Qt
    if (loc_rows > 0)
        blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, loc_rows, b, kr, (T)1, Qg, loc_rows, stack + me * b, P * b, (T)0, Ag, lda, q);
}

/// One reconstructed panel: what the apply (and hqrrp's update of its sketching matrix) needs afterwards.  The buffers live in the Scratch
/// the caller hands to `reconstruct`.
template <typename T>
struct ShardedPanel {
    int64_t act_loc = 0;     // my active rows are the local suffix [act_loc, m_loc)
    int64_t tcnt = 0;        // of which the first tcnt lie in the top block (global rows [g0, g0 + b)) ...
    int64_t toff = 0;        // ... at offset toff inside it
    int64_t below = 0;       // my rows below the top block
    int64_t b = 0;
    T* Vexp = nullptr;       // my active rows of V, explicit (unit diagonal, zeros above): (tcnt + below) x b, ld ldv
    int64_t ldv = 1;
    int64_t vrows() const { return tcnt + below; }
};

/// The orthonormal sharded panel (columns col .. col + b of A, global rows >= g0) -> reflectors.  R (ld ldr): in = the panel's triangle, out =
/// sign-fixed R11, also written onto its owners' rows of A (on and above the diagonal of the top block, `bcols` >= b columns wide when the
/// caller's R has more columns than reflectors).  T (ld ldt) and tau (b) replicated.  Q1 : b x b scratch.
template <typename T>
ShardedPanel<T> reconstruct(blas::Queue& q, blas::Scratch& keep, const ShardRows& L, T* A, int64_t lda, int64_t g0, int64_t col, int64_t b,
                            T* R, int64_t ldr, T* Tm, int64_t ldt, T* tau, T* Q1) {
    ShardedPanel<T> S;
    S.b = b;
    S.act_loc = L.local_from(g0);
    S.tcnt = L.local_from(g0 + b) - S.act_loc;
    S.toff = (S.tcnt > 0) ? (L.row0 + S.act_loc) - g0 : 0;
    const int64_t b_loc = S.act_loc + S.tcnt;
    S.below = L.m_loc - b_loc;
    lapack::laset(MatrixType::General, b, b, (T)0, (T)0, Q1, b, q);
    if (S.tcnt > 0) lapack::lacpy(MatrixType::General, S.tcnt, b, &A[S.act_loc + lda * col], lda, Q1 + S.toff, b, q);
    q.allreduce_sum(Q1, b * b);
    const int64_t ldp = b + S.below;
    T* Pst = keep.alloc<T>(ldp * b);
    T* Dv = keep.alloc<T>(b);
    lapack::lacpy(MatrixType::General, b, b, Q1, b, Pst, ldp, q);
    if (S.below > 0) lapack::lacpy(MatrixType::General, S.below, b, &A[b_loc + lda * col], lda, Pst + b, ldp, q);
    lapack::orhr_col(ldp, b, b, Pst, ldp, Tm, ldt, Dv, q);
    lapack::row_sign(b, R, ldr, Dv, q);
    lapack::tau_from_t(b, b, Tm, ldt, tau, q);
    if (S.tcnt > 0) lapack::lacpy(MatrixType::General, S.tcnt, b, Pst + S.toff, ldp, &A[S.act_loc + lda * col], lda, q);
    if (S.below > 0) lapack::lacpy(MatrixType::General, S.below, b, Pst + b, ldp, &A[b_loc + lda * col], lda, q);
    if (S.tcnt > 0) lapack::lacpy(MatrixType::Upper, S.tcnt, b - S.toff, R + S.toff + S.toff * ldr, ldr, &A[S.act_loc + lda * (col + S.toff)], lda, q);
    S.ldv = std::max<int64_t>(S.vrows(), 1);
    S.Vexp = keep.alloc<T>(S.ldv * b);
    if (S.tcnt > 0) lapack::vrows_explicit(b, S.toff, S.tcnt, Pst, ldp, S.Vexp, S.ldv, q);
    if (S.below > 0) lapack::lacpy(MatrixType::General, S.below, b, Pst + b, ldp, S.Vexp + S.tcnt, S.ldv, q);
    return S;
}

/// C_g (my active rows x ncols, at &A[act_loc + lda * col]) <- its rows of Q^T C.  One exchange (W, b x ncols).
template <typename T>
void apply_qt(blas::Queue& q, const ShardedPanel<T>& S, const T* Tm, int64_t ldt, T* A, int64_t lda, int64_t col, int64_t ncols) {
    if (ncols <= 0 || S.b <= 0) return;
    blas::Scratch w(q);
    const int64_t b = S.b, vr = S.vrows();
    T* W = w.alloc<T>(b * ncols);
    T* W2 = w.alloc<T>(b * ncols);
    T* Cg = (vr > 0) ? &A[S.act_loc + lda * col] : nullptr;
    if (vr > 0) blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, b, ncols, vr, (T)1, S.Vexp, S.ldv, Cg, lda, (T)0, W, b, q);
    else lapack::laset(MatrixType::General, b, ncols, (T)0, (T)0, W, b, q);
    q.allreduce_sum(W, b * ncols);
    blas::gemm(Layout::ColMajor, Op::Trans, Op::NoTrans, b, ncols, b, (T)1, Tm, ldt, W, b, (T)0, W2, b, q);
    if (vr > 0) blas::gemm(Layout::ColMajor, Op::NoTrans, Op::NoTrans, vr, ncols, b, (T)-1, S.Vexp, S.ldv, W2, b, (T)1, Cg, lda, q);
}

}  // namespace RandLAPACK::detail
