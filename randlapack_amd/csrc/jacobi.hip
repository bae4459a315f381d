// Thin SVD by one-sided (Hestenes) Jacobi, entirely on the GPU.
//
// Replaces lapack::gesdd(Job::SomeVec, ...) in the RSVD tail (RandLAPACK/drivers/rl_rsvd.hh:146).  The
// reference runs that on the host CPU; here the small factor never leaves HBM.  One-sided Jacobi is
// chosen because (a) it is made of independent column-pair rotations -> one workgroup per pair, wavefront
// shuffle reductions for the three inner products, and (b) it computes small singular values to high
// RELATIVE accuracy (better than bidiagonalisation-based gesdd), so the parity tolerance on sigma is
// met with margin.
//
// Ordering: round-robin tournament (N-1 rounds of N/2 disjoint pairs per sweep), one launch per round;
// the launches of a sweep are independent of the data, so the host just enqueues them.  A device
// counter accumulates the number of rotations applied in a sweep; the host reads it once per sweep.
// (Column exchanges a la de Rijk were tried and REJECTED: with the tournament ordering they slow
// convergence from ~11 to >30 sweeps on graded 256-column factors.)  Callers should hand in a
// pre-conditioned factor: Jacobi on R^T of a QR factorisation converges in ~11 sweeps regardless of
// grading, on R itself it can take 30+ (measured).
#include "rlhip_internal.h"
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <limits>

namespace rlhip {
template <typename T>
int gemm(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda, const T* B, int64_t ldb,
         T beta, T* C, int64_t ldc);
}
using rlhip::gemm;
namespace rlhip {
template <typename T>
int lacpy(rlhip_ctx* c, int uplo, int64_t m, int64_t n, const T* A, int64_t lda, T* B, int64_t ldb);
}

namespace {

template <typename T>
__device__ __forceinline__ T wsum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void jacobi_round_kernel(int64_t m, int n, int N, int round, T* __restrict__ A,
                                                           int64_t lda, T* __restrict__ V, int64_t ldv, T tol,
                                                           unsigned* __restrict__ nrot) {
    // pair of this workgroup (circle method)
    const int s = blockIdx.x;
    int p, q;
    if (s == 0) { p = N - 1; q = round % (N - 1); }
    else { p = (round + s) % (N - 1); q = (round - s + (N - 1)) % (N - 1); }
    if (p > q) { int t = p; p = q; q = t; }
    if (q >= n) return;  // phantom column of an odd-sized problem

    __shared__ T red[3][4];
    __shared__ T rot[3];
    T* __restrict__ ap = A + (int64_t)p * lda;
    T* __restrict__ aq = A + (int64_t)q * lda;
    T aa = 0, bb = 0, ab = 0;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        T x = ap[i], y = aq[i];
        aa += x * x; bb += y * y; ab += x * y;
    }
    aa = wsum(aa); bb = wsum(bb); ab = wsum(ab);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (lane == 0) { red[0][w] = aa; red[1][w] = bb; red[2][w] = ab; }
    __syncthreads();
    if (threadIdx.x == 0) {
        T alpha = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        T beta = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        T gamma = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        T cs = 1, sn = 0, swap = 0;
        T lim = tol * sqrt(alpha) * sqrt(beta);
        if (fabs(gamma) > lim && lim >= T(0) && alpha > T(0) && beta > T(0)) {
            T zeta = (beta - alpha) / (T(2) * gamma);
            T t = (zeta >= 0 ? T(1) : T(-1)) / (fabs(zeta) + sqrt(T(1) + zeta * zeta));
            cs = T(1) / sqrt(T(1) + t * t);
            sn = cs * t;
            atomicAdd(nrot, 1u);
        }
        rot[0] = cs; rot[1] = sn; rot[2] = swap;
    }
    __syncthreads();
    const T cs = rot[0], sn = rot[1];
    const bool swap = rot[2] != T(0);
    if (cs == T(1) && sn == T(0) && !swap) return;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        T x = ap[i], y = aq[i];
        T xn = cs * x - sn * y, yn = sn * x + cs * y;
        ap[i] = swap ? yn : xn;
        aq[i] = swap ? xn : yn;
    }
    if (!V) return;                      // caller does not want the rotations accumulated
    T* __restrict__ vp = V + (int64_t)p * ldv;
    T* __restrict__ vq = V + (int64_t)q * ldv;
    for (int i = threadIdx.x; i < n; i += 256) {
        T x = vp[i], y = vq[i];
        T xn = cs * x - sn * y, yn = sn * x + cs * y;
        vp[i] = swap ? yn : xn;
        vq[i] = swap ? xn : yn;
    }
}


// 64-lane sum without LDS traffic: four DPP row_shr steps leave each 16-lane row's total in its last lane, four
// v_readlane pick those up as scalars.  (__shfl_xor lowers to ds_bpermute_b32 pairs: 36 LDS round trips per
// column pair, which made the reductions -- not the rotations -- the cost of a round.)
__device__ __forceinline__ double dpp_shr_add(double v, const int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2, hi2;
    switch (ctrl_sel) {   // row_shr:1,2,4,8 with bound_ctrl (zero fill)
        case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true); break;
        case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xF, 0xF, true); break;
        case 4: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xF, 0xF, true); break;
        default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xF, 0xF, true); break;
    }
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_shr_add(v, 1);
    v = dpp_shr_add(v, 2);
    v = dpp_shr_add(v, 4);
    v = dpp_shr_add(v, 8);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return (r0 + r1) + (r2 + r3);
}

// fp64 reciprocal / reciprocal square root from the hardware seeds (v_rcp_f64 / v_rsq_f64) + two Newton steps:
// ~10 instructions instead of the ~40 of an IEEE division or sqrt.  The rotation parameters are wave-uniform
// scalars evaluated on the vector ALU, so this is the critical path of the LDS-resident sweep.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ double rcp1(double x) {
    double y = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ double rsq1(double x) {
    double y = __builtin_amdgcn_rsq(x);
    return y * fma(-0.5 * x * y, y, 1.5);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y * fma(-0.5 * x * y, y, 1.5);
}

// ---------------------------------------------------------------------------------------------------
// LDS-resident block Jacobi for short matrices (m <= 256 rows, e.g. the k x k factor R^T of the RSVD tail).
// The per-round launches of the kernel above cost ~9 us each (4.5 us of work + the launch gap) x 255 rounds
// x 11 sweeps = 25 ms at k = 256 -- a quarter of the whole RSVD.  Here the n columns are cut into blocks of
// 32; one workgroup owns a PAIR of blocks (64 columns = a 256 x 64 fp64 panel = 128 KiB of LDS), runs a
// complete round-robin sweep over the 2016 column pairs inside the panel out of LDS (one wavefront per
// pair, shuffle reductions, no global traffic), accumulates the rotations in a 64 x 64 matrix J (32 KiB,
// LDS: 160 KiB in total, the whole CU), writes the panel back and applies J to the matching 64 columns of
// V.  An outer tournament over the block pairs (NB-1 launches of NB/2 workgroups) visits every column
// pair once per outer sweep.
constexpr int JM = 256;           // panel rows (LDS)

// JB = block width (16 or 32), JP = 2*JB = panel width.  Narrow blocks put twice as many workgroups to work per
// launch and shrink the per-launch pair count 4x; launches per sweep double (measured trade-off in DESIGN.md 4.3).
//
// Work assignment inside a round: a column pair is handled by a QUARTER wave (16 lanes, 16 rows per lane as eight
// 16-byte LDS accesses), four pairs per wavefront.  The rotation parameters are then ordinary per-lane values
// and the instruction stream (dot products, DPP row-rotate all-reduce, rcp/rsq based rotation, update) is issued once
// for four pairs.  The earlier one-pair-per-wavefront layout was VALU-issue bound: ~150 instructions per pair with
// four wavefronts per SIMD = 1.5 us per round; this layout issues ~65 per pair from one or two wavefronts per SIMD.
__device__ __forceinline__ double dpp_ror_add(double v, const int n) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2, hi2;
    switch (n) {   // row_ror:n  (rotate inside each 16-lane row)
        case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xF, 0xF, false); break;
        case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xF, 0xF, false); break;
        case 4: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xF, 0xF, false); break;
        default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false); break;
    }
    return v + __hiloint2double(hi2, lo2);
}
// every lane of a 16-lane row ends up with the row's sum
__device__ __forceinline__ double row16_allsum(double v) {
    v = dpp_ror_add(v, 8);
    v = dpp_ror_add(v, 4);
    v = dpp_ror_add(v, 2);
    return dpp_ror_add(v, 1);
}

// every lane of a 32-lane half wave ends up with the half's sum: the 16-lane all-reduce, then v_permlane16_swap (gfx950) trades row 1 of one copy
// for row 0 of the other (rows 3 / 2 likewise), so copy a holds the even row's sum in both rows, copy b the odd row's
__device__ __forceinline__ double half32_allsum(double v) {
    v = row16_allsum(v);
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
template <int QW>
__device__ __forceinline__ double pair_allsum(double v) {
    if constexpr (QW == 16) return row16_allsum(v);
    else return half32_allsum(v);
}

// The rotation rounds of ONE block pair held in LDS (Xs: [2 JB][JMT] column-major, Js: the 2 JB x 2 JB rotation accumulator, only touched
// when want_v).  Shared by the per-launch kernel and the persistent kernel below, so both execute the same arithmetic in the same order.
//   intra = 1: the 2 x JB(JB-1)/2 pairs INSIDE the two blocks (JB-1 rounds, JB/2 pairs per block per round)
//   intra = 0: the JB x JB CROSS pairs between the blocks (JB rounds of JB pairs)
// my_rot / my_cos2 accumulate the rotation count and the largest squared cosine met (per thread; lanes with ql == 0 count).
// Cross rounds MAINTAIN the squared column norms instead of recomputing them: the quarter's own column carries its norm in a register, the
// travelling partner column carries it through the LDS array Ns (one value per panel column), and a rotation updates both by Rutishauser's
// formulas ||x'||^2 = aa - t ab, ||y'||^2 = bb + t ab.  Fresh norms are taken in round 0 of every call (= every 16 rounds), so the drift
// is that of 16 updates.  Only the inner product of the pair is reduced in a round: one dot product and one 16-lane all-reduce instead
// of three of each (a third of a round's instructions).  my_cos2 is a FLAG (1: some pair met a cosine above 1e-9), my_nmax / my_nmin the
// largest / smallest non-zero fresh squared norm seen (the persistent kernel's condition monitor).
// QW = lanes per column pair: 16 (a quarter wave, 16 rows per lane, four pairs per wavefront, 4 rotating waves at JB = 16) or 32 (a half wave,
// 8 rows per lane, two pairs per wavefront, 8 rotating waves: two per SIMD, so one wave's dependent chain -- reduce, rotation
// parameters, update, LDS round trip -- runs behind the other's).
template <typename T, int JB, int JMT, int QW>
__device__ __forceinline__ void jacobi_pair_rounds(T* __restrict__ Xs, T* __restrict__ Js, const bool want_v, const int intra, const double tol2,
                                                   unsigned& my_rot, float& my_cos2, double* __restrict__ Ns, float& my_nmax, float& my_nmin) {
    constexpr int JP = 2 * JB;
    constexpr int NROT = QW * JB;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int ql = lane & (QW - 1);
    // intra = 1: the 2 x JB(JB-1)/2 pairs INSIDE the two blocks (JB-1 rounds, JB/2 pairs per block per round)
    // intra = 0: the JB x JB CROSS pairs between the blocks (JB rounds of JB pairs): every column pair of the
    //            matrix is then visited exactly once per outer sweep (NB-1 cross launches + 1 intra launch)
    const int nrounds = intra ? (JB - 1) : JB;
    const int sl = tid / QW;                           // pair slot of this quarter / half wave: 0 .. JB-1
    constexpr int RL = JMT / (2 * QW);                 // 16-byte row pairs per lane
    constexpr bool MAINT = (JB == 16);                  // (the 32-wide panels of tiny problems fill the whole LDS: no room for Ns, norms recomputed)
    d2_t x[RL], y[RL];
    double aa = 0;                                      // squared norm of the quarter's own column (cross rounds: carried from round to round)
    for (int round = 0; round < nrounds; ++round) {
        if (tid >= NROT) { __syncthreads(); continue; }     // wave-uniform: whole wavefronts sit the rounds out
        int p, q;
        if (intra) {
            const int blk = sl / (JB / 2), s16 = sl % (JB / 2);
            if (s16 == 0) { p = JB - 1; q = round; }
            else { p = (round + s16) % (JB - 1); q = (round - s16 + (JB - 1)) % (JB - 1); }
            if (p > q) { int t = p; p = q; q = t; }
            p += JB * blk; q += JB * blk;
        } else {
            p = sl; q = JB + ((sl + round) & (JB - 1));
        }
        // rows 2*ql + 32*r, 2*ql + 32*r + 1  (r = 0..7): 16-byte accesses, a quarter wave covers 256 contiguous bytes
        d2_t* xp = reinterpret_cast<d2_t*>(Xs + p * JMT) + ql;
        d2_t* xq = reinterpret_cast<d2_t*>(Xs + q * JMT) + ql;
        // Cross rounds keep the quarter's own column p (= its slot) in registers from the first round to the last: only the
        // partner column q makes the LDS round trip.  The rounds are bound by LDS bandwidth (16 pairs x 2 columns x 2 KiB read and
        // written = 1024 clocks of the CU's 128 B/clk pipe), so this halves their cost.
        double bb = 0, ab = 0;
        const bool fresh = !MAINT || intra || round == 0;
        if (fresh) {
#pragma unroll
            for (int r = 0; r < RL; ++r) x[r] = xp[QW * r];
        }
#pragma unroll
        for (int r = 0; r < RL; ++r) y[r] = xq[QW * r];
        if (fresh) {
            aa = 0;
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                aa = fma(x[r].x, x[r].x, aa); aa = fma(x[r].y, x[r].y, aa);
                bb = fma(y[r].x, y[r].x, bb); bb = fma(y[r].y, y[r].y, bb);
                ab = fma(x[r].x, y[r].x, ab); ab = fma(x[r].y, y[r].y, ab);
            }
            aa = pair_allsum<QW>(aa);
            bb = pair_allsum<QW>(bb);
            ab = pair_allsum<QW>(ab);
            const float fa = (float)aa, fb = (float)bb;
            my_nmax = fmaxf(my_nmax, fmaxf(fa, fb));
            if (fa > 0.f) my_nmin = fminf(my_nmin, fa);
            if (fb > 0.f) my_nmin = fminf(my_nmin, fb);
        } else {
            if constexpr (MAINT) bb = Ns[q];                                // the partner's norm travels with it
#pragma unroll
            for (int r = 0; r < RL; ++r) { ab = fma(x[r].x, y[r].x, ab); ab = fma(x[r].y, y[r].y, ab); }
            ab = pair_allsum<QW>(ab);
        }
        const double nn = aa * bb, ab2 = ab * ab;
        const bool live = (aa > 0.0) && (bb > 0.0);
        const bool rot = live && (ab2 > tol2 * nn);                         // |ab| > tol*||x||*||y||
        if (live && ab2 > 1e-18 * nn) my_cos2 = 1.0f;                       // a cosine above 1e-9 was met
        // t = sign(d g) |g| / (|d| + sqrt(d^2 + g^2)),  d = bb - aa, g = 2ab.  t only has to make the (p,q) inner product small
        // (one Newton step on the seeds = ~48 bits); cs = rsqrt(1 + t^2) keeps two steps so that cs^2 + sn^2 = 1 to rounding.
        const double dd = bb - aa, gg = 2.0 * ab;
        const double h2 = fma(dd, dd, gg * gg);
        const double h = h2 * rsq1(h2);
        double tt = gg * rcp1(fabs(dd) + h);
        tt = (dd < 0.0) ? -tt : tt;
        tt = rot ? tt : 0.0;                                                // also discards inf/NaN from ab == 0
        const double cs = fast_rsqrt(fma(tt, tt, 1.0)), sn = cs * tt;
        if constexpr (MAINT) {
            if (!intra) {                                                   // (intra rounds take fresh norms every time)
                aa = fma(-tt, ab, aa);                                      // (explicit fmas: the two kernels that inline this must round alike)
                if (ql == 0) Ns[q] = fma(tt, ab, bb);
            }
        }
        if (__builtin_amdgcn_ballot_w64(rot)) {                             // skip the stores when no quarter rotates
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                d2_t xn, yn;
                xn.x = cs * x[r].x - sn * y[r].x; xn.y = cs * x[r].y - sn * y[r].y;
                yn.x = sn * x[r].x + cs * y[r].x; yn.y = sn * x[r].y + cs * y[r].y;
                if (intra) xp[QW * r] = xn;
                x[r] = xn;
                xq[QW * r] = yn;
            }
            if (want_v) {                // the rotation accumulator is only needed when V is wanted
#pragma unroll
                for (int r = 0; r < (JP + QW - 1) / QW; ++r) {
                    const int jr = ql + QW * r;
                    if (jr < JP) {
                        const double jp = Js[jr + p * JP], jq = Js[jr + q * JP];
                        Js[jr + p * JP] = cs * jp - sn * jq;
                        Js[jr + q * JP] = sn * jp + cs * jq;
                    }
                }
            }
        }
        my_rot += (rot && ql == 0) ? 1u : 0u;
        __syncthreads();
    }
    if (!intra && tid < NROT) {                        // the register-resident column goes back once
        d2_t* xp = reinterpret_cast<d2_t*>(Xs + sl * JMT) + ql;
#pragma unroll
        for (int r = 0; r < RL; ++r) xp[QW * r] = x[r];
    }
    __syncthreads();
}

// JMT = panel rows held in LDS: 256 (1024 threads) or 512 (512 threads: the rotating lanes then carry 32 rows of two columns = 128 VGPRs)
template <typename T, int JB, int JMT, int QW>
__global__ __launch_bounds__(JMT == 256 ? 1024 : 512) void jacobi_block_kernel(int m, int n, int NB, int oround, int intra, T* __restrict__ A,
                                                               int64_t lda, T* __restrict__ V, int64_t ldv, T tol,
                                                               unsigned* __restrict__ nrot, int multi) {
    constexpr int JP = 2 * JB;
    constexpr int NT = (JMT == 256) ? 1024 : 512;                           // all 16 waves move data; the first 16*JB threads rotate
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Xs = reinterpret_cast<T*>(smem_raw);            // [JP][JMT] column-major
    T* Js = Xs + JP * JMT;                              // [JP][JP] column-major
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block pair of this workgroup (circle method over NB blocks)
    int P, Q;
    {
        const int sl = blockIdx.x;
        if (intra) { P = 2 * sl; Q = 2 * sl + 1; }
        else if (sl == 0) { P = NB - 1; Q = oround % (NB - 1); }
        else { P = (oround + sl) % (NB - 1); Q = (oround - sl + (NB - 1)) % (NB - 1); }
        if (P > Q) { int t = P; P = Q; Q = t; }
    }
    auto gcol = [&](int c) { return (c < JB) ? (P * JB + c) : (Q * JB + (c - JB)); };   // panel col -> global col
    // ---- load panel (zero padded), J = I
    for (int e = tid; e < JP * JMT; e += NT) {
        const int r = e % JMT, c = e / JMT;
        const int gc = gcol(c);
        Xs[e] = (r < m && gc < n) ? A[r + (int64_t)gc * lda] : T(0);
    }
    for (int e = tid; e < JP * JP; e += NT) Js[e] = ((e % JP) == (e / JP)) ? T(1) : T(0);
    __syncthreads();
    unsigned my_rot = 0;
    float my_cos2 = 0.f;                               // 1: a cosine above 1e-9 was met before rotating (convergence shortcut)
    float my_nmax = 0.f, my_nmin = 3e38f;
    __shared__ double Ns[JB == 16 ? JP : 1];
    if (multi > 0) {
        // TWO blocks (n <= 2 JB columns: the whole matrix is this workgroup's panel): complete sweeps -- pairs inside the blocks, then the cross
        // pairs -- repeated HERE until one rotates nothing, instead of two launches and a host read per sweep (ABRIK's projected matrix, 64 x 32:
        // 10 launches + 5 round trips = 0.65 ms of a 4.8 ms call).  nrot[0] ends up with the LAST sweep's rotations (0 = converged), nrot[2] = sweeps.
        // (the workgroup-wide "did anybody rotate" goes through the device word nrot[3]: the panel and J fill the CU's LDS to the last byte
        //  -- __syncthreads_or's own LDS word no longer fits)
        int sw = 0;
        unsigned seen = 0;
        while (sw < multi) {
            unsigned rot = 0;
            jacobi_pair_rounds<T, JB, JMT, QW>(Xs, Js, V != nullptr, 1, (double)tol * (double)tol, rot, my_cos2, Ns, my_nmax, my_nmin);
            jacobi_pair_rounds<T, JB, JMT, QW>(Xs, Js, V != nullptr, 0, (double)tol * (double)tol, rot, my_cos2, Ns, my_nmax, my_nmin);
            ++sw;
            my_rot = rot;
            if (__builtin_amdgcn_ballot_w64(rot != 0) != 0 && lane == 0) atomicAdd(nrot + 3, 1u);
            __syncthreads();
            const unsigned now = __hip_atomic_load(nrot + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();                           // nobody counts the next sweep before everybody has read this one's total
            if (now == seen) break;
            seen = now;
        }
        if (tid == 0) nrot[2] = (unsigned)sw;
    } else
    jacobi_pair_rounds<T, JB, JMT, QW>(Xs, Js, V != nullptr, intra, (double)tol * (double)tol, my_rot, my_cos2, Ns, my_nmax, my_nmin);
    // one atomic pair per wavefront
    {
        unsigned r = my_rot;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
        float c2 = my_cos2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c2 = fmaxf(c2, __shfl_xor(c2, off, 64));
        if (lane == 0 && r) {
            atomicAdd(nrot, r);
            atomicMax(nrot + 1, __float_as_uint(c2));   // non-negative floats order like their bit patterns
        }
    }
    // ---- write the rotated panel back
    for (int e = tid; e < JP * JMT; e += NT) {
        const int r = e % JMT, c = e / JMT;
        const int gc = gcol(c);
        if (r < m && gc < n) A[r + (int64_t)gc * lda] = Xs[e];
    }
    // ---- V[:, panel] <- V[:, panel] * J on the matrix cores: the panel's LDS is free now, so a 256-row slab of
    //      V's panel columns is staged there; a wave owns 16-row groups of the slab x all JP columns
    //      (JP/16 MFMA tiles x JP/4 k-steps each).  Operands are swapped (J as the MFMA A operand) so that each
    //      lane's results run along rows -> 128-byte store segments.
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const int fr = lane & 15, fk = lane >> 4;
    for (int r0 = 0; V != nullptr && r0 < n; r0 += JMT) {
        __syncthreads();
        for (int e = tid; e < JP * JMT; e += NT) {
            const int r = e % JMT, c = e / JMT;
            const int gc = gcol(c);
            Xs[e] = (r0 + r < n && gc < n) ? V[(r0 + r) + (int64_t)gc * ldv] : T(0);
        }
        __syncthreads();
        constexpr int NU = JP / 16;
        for (int rg = wid; rg < JMT / 16; rg += NW) {
            d4_t acc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = d4_t{0, 0, 0, 0};
#pragma unroll 4
            for (int st = 0; st < JP / 4; ++st) {
                const double vf = (double)Xs[(16 * rg + fr) + (4 * st + fk) * JMT];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const double jf = (double)Js[(4 * st + fk) + (16 * u + fr) * JP];
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(jf, vf, acc[u], 0, 0, 0);
                }
            }
            const int row = r0 + 16 * rg + fr;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gc = gcol(16 * u + fk + 4 * r);
                    if (row < n && gc < n) V[row + (int64_t)gc * ldv] = (T)acc[u][r];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One launch for MANY sweeps (the k x k factor of the RSVD tail, singular values and left vectors only).
//
// The per-launch kernel above pays ~14 us of fixed cost around ~5.5 us of rotations for each of the 16 steps of a sweep (launch gap,
// panel load / store through L2, pipeline refill) and the host reads a counter after every sweep.  Here NB/2 workgroups stay resident
// (cooperative launch), keep their block pair in LDS and hand blocks to one another through an UNCACHED exchange buffer:
//   * a block is published as 8-byte agent-scope stores, then `s_waitcnt vmcnt(0)`, then ONE tagged word {step : block} -- the only
//     reader of version s of a block is its next owner, which is also the only next writer, so the buffer needs no double buffering;
//   * the circle-method pairing and the round order inside a pair are those of the per-launch kernel (jacobi_pair_rounds is shared), so
//     the two paths produce bitwise identical matrices -- tests compare them;
//   * at the end of a sweep every workgroup publishes {sweep : max cos^2} and {sweep : rotations} and reads everybody's: all take
//     the same decision (converged / every cosine <= 1e-9 -> hand back for the Gram verification / next sweep) without the host;
//   * the input is only READ; results leave through the exchange buffer (column-major m x NB JB, ld m) and the host copies them over A
//     when the launch reports success.  A lost word (bounded spins) reports -7 and A is untouched: the caller then takes the
//     per-launch path.
template <typename T>
struct JpArgs {
    int m, n, NB, sweep0, max_sweeps;
    const T* A;
    int64_t lda;
    unsigned long long* X;        // NB * JB * m values (as bit patterns)
    unsigned long long* bflag;    // [NB]      {step : 1}
    unsigned long long* sflag;    // [2][NB/2][2]  {sweep : cos^2 bits}, {sweep : rotations}; indexed by sweep parity
    T tol;
    int* out;                     // [0] status: 1 converged (a sweep without rotations), 2 every cosine <= 1e-9 (verify), 3 sweep limit, -7 lost word
                                  // [1] sweeps done (absolute)   [2] != 0: SOME workgroup lost a word (its blocks in X are stale whatever [0] says)
    unsigned long long* done;     // one word of the exchange buffer: set by workgroup 0 when it leaves; releases the clock holders (below)
    int hold_mode;                // what the holders issue: 1 bursts of 128 fp64 FMAs with hold_nap x s_sleep(8) between, 2 s_sleep only (experiments)
    int hold_nap, hold_delay_us;  // holders only sleep for the first hold_delay_us of the launch
    float norm_ratio_lim;         // > 0: give up (status -9) when max / min of the squared column norms exceeds it at the end of a sweep
    int trans_upper;              // 1: the input is X = R^T of the UPPER triangle stored in A (X(r, c) = A(c, r) for c <= r, else 0): the Cholesky factor as potrf left it
    const int* skip;              // nullptr, or a device word: != 0 -> the launch does nothing and reports status -8 (an earlier kernel of the stream failed)
    // same-XCD hand-over (below): local_try != 0 puts worker w on workgroup 8 w; XL / bflagL are CACHED twins of X / bflag, census [NB/2] words of the exchange buffer
    int local_try;
    unsigned long long* XL;
    unsigned long long* bflagL;
    unsigned long long* census;
};

// Clock holders.  The part's clock follows the occupancy of the shader array, not the power budget: with 8 of 256 CUs at work (this
// kernel) it sinks from 2.4 to ~2.05 GHz within 4 ms and needs ~20 ms of full load to come back (scripts/dvfs_probe.py) --
// which slows the rotation rounds (pure VALU / LDS latency chains) AND the tall product that follows.  The launch therefore covers every
// CU: workgroups >= NB/2 do nothing but keep their SIMDs issuing fp64 FMAs at low priority until workgroup 0 reports the end (one
// poller per workgroup, every ~2 us, bounded by the wall clock).  They share no CU with a worker (1024 threads x 128 VGPRs fill one).
__device__ __forceinline__ void jacobi_clock_holder(const unsigned long long* done, unsigned* lds_flag, const int mode, const int nap, const long long delay_ticks) {
    __builtin_amdgcn_s_setprio(0);
    const int tid = threadIdx.x;
    double z = 0.5;
    const double x = 1.0 - 1e-9 * (tid + 1), y = 1e-9 * (tid + 1);
    const long long t0 = wall_clock64();
    if (tid == 0) *lds_flag = 0;
    __syncthreads();
    for (int it = 0;; ++it) {
        const bool napping = (mode == 2) || (wall_clock64() - t0 < delay_ticks);
        if (!napping) {
#pragma unroll
            for (int i = 0; i < 128; ++i) z = fma(z, x, y);
            for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(8);
        } else {
            __builtin_amdgcn_s_sleep(127);
        }
        if (tid == 0 && (it & 3) == 0) {
            const unsigned long long v = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != 0ull || wall_clock64() - t0 > 3000000ll) *(volatile unsigned*)lds_flag = 1u;     // released, or 30 ms: never outlive a failed launch
        }
        if (*(volatile unsigned*)lds_flag) break;
    }
    if (z == 12345.678) *lds_flag = 2u;                // (keeps the chain)
}

__device__ __forceinline__ bool jp_wait(const unsigned long long* w, unsigned tag, unsigned long long* got) {
    for (int spins = 0; spins < (1 << 22); ++spins) {
        const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == tag) { *got = v; return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <typename T, int JB, int JMT, int QW>
__global__ __launch_bounds__(1024) void jacobi_persist_kernel(JpArgs<T> g) {
    static_assert(sizeof(T) == 8, "fp64 only");
    constexpr int JP = 2 * JB, NT = 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Xs = reinterpret_cast<T*>(smem_raw);            // [JP][JMT] column-major
    __shared__ unsigned s_rot, s_cos, s_lost, s_nmax, s_nmin;
    __shared__ double Ns[JP];
    const int tid = threadIdx.x, lane = tid & 63;
    // Same-XCD hand-over.  The eight XCDs have one L2 each: a block handed over through the uncached exchange buffer makes a round trip to
    // the memory side (~2.5 us of a ~14 us step).  Workgroup b is observed to run on XCD b % 8 (no contract: MI355X_MICROARCH.md), so with
    // local_try the workers are workgroups 0, 8, 16, ... -- one XCD if the observation holds -- and a CENSUS decides: every worker
    // publishes its HW_REG_XCC_ID through the uncached buffer and reads everybody's; only if all are equal the blocks travel through a
    // CACHED buffer with plain stores (the line stays in the XCD's L2) and sc1 loads (past the CU's L1, served by that L2).  Any other
    // placement keeps the uncached protocol -- same words, same order, same results.
    const int NW = g.NB / 2, m = g.m;
    const bool is_worker = g.local_try ? ((blockIdx.x & 7) == 0 && (int)(blockIdx.x >> 3) < NW) : ((int)blockIdx.x < NW);
    const int w = g.local_try ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (!is_worker) { jacobi_clock_holder(g.done, &s_lost, g.hold_mode, g.hold_nap, (long long)g.hold_delay_us * 100); return; }
    if (g.skip != nullptr && *g.skip != 0) {           // (every worker reads the same word: all of them leave)
        if (w == 0 && tid == 0) {
            g.out[0] = -8; g.out[1] = g.sweep0;
            __hip_atomic_store(g.done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    unsigned long long clk0 = 0, wall0 = 0;
    if (w == 0 && tid == 0) { clk0 = __builtin_readcyclecounter(); wall0 = wall_clock64(); }
    const double tol2 = (double)g.tol * (double)g.tol;
    int held[2] = {2 * w, 2 * w + 1};
    for (int e = tid; e < JP * JMT; e += NT) {
        const int r = e % JMT, c = e / JMT;
        const int gc = held[c / JB] * JB + (c % JB);
        if (g.trans_upper) Xs[e] = (r < m && gc < g.n && gc <= r) ? g.A[gc + (int64_t)r * g.lda] : T(0);
        else Xs[e] = (r < m && gc < g.n) ? g.A[r + (int64_t)gc * g.lda] : T(0);
    }
    if (tid == 0) { s_lost = 0; s_rot = 0; }
    __syncthreads();
    bool local = false;
    if (g.local_try) {
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            __hip_atomic_store(g.census + w, (0xCE5505ull << 32) | (unsigned long long)((xcc & 15u) + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < NW) {
            unsigned long long got = 0;
            if (!jp_wait(g.census + tid, 0xCE5505u, &got)) atomicExch(&s_lost, 1u);
            else atomicOr(&s_rot, 1u << ((unsigned)got & 31u));          // the set of XCDs the workers sit on
        }
        __syncthreads();
        local = !s_lost && __builtin_popcount(s_rot) == 1;
        __syncthreads();
        if (tid == 0) { s_rot = 0; if (w == 0) g.out[3] = local ? 1 : 0; }
        __syncthreads();
    }
    unsigned long long* const Xx = local ? g.XL : g.X;               // where blocks travel during the sweeps (the RESULT always leaves through g.X)
    unsigned long long* const bfx = local ? g.bflagL : g.bflag;
    // A block is JB x m values = PER_T per thread (4 at m = 256).  All of a thread's loads of a hand-over are issued before the first one is
    // consumed: one round trip to the uncached buffer instead of PER_T dependent ones (the first version took ~11 us per hand-over, two
    // thirds of every step; rocprof timeline in DESIGN.md 4.3).
    constexpr int PER_T = (JB * JMT + NT - 1) / NT;
    // (write-through stores issued from inline asm: in front of every agent-scope atomic store of such a loop hipcc places an
    // s_waitcnt vmcnt(0), i.e. the PER_T stores of a thread went out one acknowledged round trip at a time -- most of the 4.5 us a hand-over took)
    auto publish = [&](int half, int blk, unsigned long long* base, bool plain) {            // LDS half -> exchange buffer
        unsigned long long* dst = base + (size_t)blk * JB * m;
        unsigned long long val[PER_T];
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const int e = tid + q * NT;
            const int ee = (e < JB * m) ? e : 0;
            val[q] = (unsigned long long)__double_as_longlong((double)Xs[(half * JB + ee / m) * JMT + ee % m]);
        }
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const int e = tid + q * NT;
            if (e < JB * m) {
                if (plain) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst + e), "v"(val[q]) : "memory");
                else asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst + e), "v"(val[q]) : "memory");
            }
        }
    };
    auto fetch2 = [&](bool f0, int blk0, bool f1, int blk1) {      // both halves in ONE batch of loads
        unsigned long long v[2][PER_T];
        const unsigned long long* src0 = Xx + (size_t)blk0 * JB * m;
        const unsigned long long* src1 = Xx + (size_t)blk1 * JB * m;
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const int e = tid + q * NT;
            v[0][q] = (f0 && e < JB * m) ? __hip_atomic_load(src0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            v[1][q] = (f1 && e < JB * m) ? __hip_atomic_load(src1 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
            const int e = tid + q * NT;
            if (e < JB * m) {
                const int r = e % m, c = e / m;
                if (f0) Xs[c * JMT + r] = (T)__longlong_as_double((long long)v[0][q]);
                if (f1) Xs[(JB + c) * JMT + r] = (T)__longlong_as_double((long long)v[1][q]);
            }
        }
    };
    unsigned gs = 0;                                   // exchange steps taken by this launch
    int status = 3, sweep = g.sweep0;
    bool lost = false;
    for (; sweep < g.max_sweeps && !lost; ++sweep) {
        unsigned my_rot = 0;
        float my_cos2 = 0.f, my_nmax = 0.f, my_nmin = 3e38f;
        jacobi_pair_rounds<T, JB, JMT, QW>(Xs, nullptr, false, 1, tol2, my_rot, my_cos2, Ns, my_nmax, my_nmin);          // pairs inside whichever two blocks are here
        for (int oround = 0; oround < g.NB - 1; ++oround) {
            int P, Q;
            if (w == 0) { P = g.NB - 1; Q = oround % (g.NB - 1); }
            else { P = (oround + w) % (g.NB - 1); Q = (oround - w + (g.NB - 1)) % (g.NB - 1); }
            if (P > Q) { const int t = P; P = Q; Q = t; }
            const int want[2] = {P, Q};
            ++gs;
            const bool out0 = held[0] != want[0], out1 = held[1] != want[1];
            __builtin_amdgcn_s_waitcnt(0x0F70);                         // (vmcnt(0), said to the compiler: no conservative wait between the two hand-overs below)
            if (out0) publish(0, held[0], Xx, local);
            if (out1) publish(1, held[1], Xx, local);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the block's words have been acknowledged ...
            __syncthreads();
            if (tid == 0) {                                              // ... before its version word goes out
                const unsigned long long word = ((unsigned long long)gs << 32) | 1u;
                if (local) {                                             // (plain: the word stays in the XCD's L2, where the next owner's sc1 poll finds it)
                    if (out0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(bfx + held[0]), "v"(word) : "memory");
                    if (out1) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(bfx + held[1]), "v"(word) : "memory");
                } else {
                    if (out0) __hip_atomic_store(bfx + held[0], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (out1) __hip_atomic_store(bfx + held[1], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (tid < 2 && (tid == 0 ? out0 : out1)) {
                unsigned long long got;
                if (!jp_wait(bfx + want[tid], gs, &got)) atomicExch(&s_lost, 1u);
            }
            __syncthreads();
            if (s_lost) { lost = true; break; }
            fetch2(out0, want[0], out1, want[1]);
            held[0] = want[0]; held[1] = want[1];
            __syncthreads();
            jacobi_pair_rounds<T, JB, JMT, QW>(Xs, nullptr, false, 0, tol2, my_rot, my_cos2, Ns, my_nmax, my_nmin);
        }
        if (lost) break;
        // ---- end of the sweep: everybody learns the sweep's rotation count, whether a cosine above 1e-9 was met, and the range of the column norms
        if (tid == 0) { s_rot = 0; s_cos = 0; s_nmax = 0; s_nmin = __float_as_uint(3e38f); }
        __syncthreads();
        {
            unsigned r = my_rot;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
            float c2 = my_cos2, hi = my_nmax, lo = my_nmin;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                c2 = fmaxf(c2, __shfl_xor(c2, off, 64));
                hi = fmaxf(hi, __shfl_xor(hi, off, 64));
                lo = fminf(lo, __shfl_xor(lo, off, 64));
            }
            if (lane == 0) {                            // non-negative floats order like their bit patterns
                if (r) { atomicAdd(&s_rot, r); atomicMax(&s_cos, __float_as_uint(c2)); }
                atomicMax(&s_nmax, __float_as_uint(hi));
                atomicMin(&s_nmin, __float_as_uint(lo));
            }
        }
        __syncthreads();
        const unsigned tag = (unsigned)(sweep - g.sweep0 + 1);
        unsigned long long* sf = g.sflag + (size_t)(tag & 1u) * NW * 4;
        if (tid < 4) {
            const unsigned v = (tid == 0) ? s_cos : (tid == 1) ? s_rot : (tid == 2) ? s_nmax : s_nmin;
            __hip_atomic_store(sf + 4 * w + tid, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid == 0) { s_rot = 0; s_cos = 0; s_nmax = 0; s_nmin = __float_as_uint(3e38f); }
        __syncthreads();
        if (tid < 4 * NW) {
            unsigned long long got = 0;
            if (!jp_wait(sf + tid, tag, &got)) atomicExch(&s_lost, 1u);
            else if ((tid & 3) == 0) atomicMax(&s_cos, (unsigned)got);
            else if ((tid & 3) == 1) atomicAdd(&s_rot, ((unsigned)got) ? 1u : 0u);
            else if ((tid & 3) == 2) atomicMax(&s_nmax, (unsigned)got);
            else atomicMin(&s_nmin, (unsigned)got);
        }
        __syncthreads();
        if (s_lost) { lost = true; break; }
        const unsigned any_rot = s_rot;
        const float cos2 = __uint_as_float(s_cos), nhi = __uint_as_float(s_nmax), nlo = __uint_as_float(s_nmin);
        __syncthreads();
        if (any_rot == 0) { status = 1; ++sweep; break; }
        if (cos2 <= 1e-18f) { status = 2; ++sweep; break; }
        // condition monitor (the Gram route): the squared column norms approach the squared singular values from inside, so their range
        // only grows; once it exceeds the limit the caller's accuracy test cannot pass any more and the remaining sweeps are not worth running
        if (g.norm_ratio_lim > 0.f && !(nhi <= g.norm_ratio_lim * nlo)) { status = -9; ++sweep; break; }
    }
    if (lost) {
        status = -7;
        if (tid == 0) atomicExch(g.out + 2, 1);        // whoever loses a word says so: workgroup 0 may well have run to a verdict
    } else {
        publish(0, held[0], g.X, false);
        publish(1, held[1], g.X, false);
    }
    if (w == 0 && tid == 0) {
        g.out[0] = status; g.out[1] = sweep;
        __hip_atomic_store(g.done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long* tk = reinterpret_cast<unsigned long long*>(g.out + 4);      // diagnostics: shader cycles and 100 MHz ticks of this launch
        tk[0] = __builtin_readcyclecounter() - clk0; tk[1] = wall_clock64() - wall0;
    }
}

// dst (ldd) = (TD) src (lds), m x n
template <typename TS, typename TD>
__global__ void convert_kernel(int64_t m, int64_t n, const TS* __restrict__ src, int64_t lds_, TD* __restrict__ dst, int64_t ldd) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * n) return;
    const int64_t i = idx % m, j = idx / m;
    dst[i + j * ldd] = (TD)src[i + j * lds_];
}

// column norms -> S (unsorted), one workgroup per column
template <typename T>
__global__ __launch_bounds__(256) void colnorm_kernel(int64_t m, const T* __restrict__ A, int64_t lda,
                                                      T* __restrict__ S) {
    __shared__ T red[4];
    const T* col = A + (int64_t)blockIdx.x * lda;
    // scaled accumulation is unnecessary here: entries are bounded by the singular values themselves
    T acc = 0;
    for (int64_t i = threadIdx.x; i < m; i += 256) acc += col[i] * col[i];
    acc = wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) S[blockIdx.x] = sqrt(red[0] + red[1] + red[2] + red[3]);
}

// rank[j] = position of column j in descending order of S (stable)
template <typename T>
__global__ void rank_kernel(int n, const T* __restrict__ S, int* __restrict__ rank) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    T sj = S[j];
    int r = 0;
    for (int i = 0; i < n; ++i) {
        T si = S[i];
        r += (si > sj) || (si == sj && i < j);
    }
    rank[j] = r;
}

// scatter normalised columns into sorted position: Uout[:, rank[j]] = A[:, j] / S[j]; VT[rank[j], :] = V[:, j]^T
template <typename T>
__global__ __launch_bounds__(256) void finalize_kernel(int64_t m, int n, const T* __restrict__ A, int64_t lda,
                                                       const T* __restrict__ V, int64_t ldv,
                                                       const T* __restrict__ S, const int* __restrict__ rank,
                                                       T* __restrict__ Uout, int64_t ldu, T* __restrict__ Sout,
                                                       T* __restrict__ VT, int64_t ldvt) {
    const int j = blockIdx.x;
    const int r = rank[j];
    const T s = S[j];
    const T inv = (s > T(0)) ? T(1) / s : T(0);
    const T* col = A + (int64_t)j * lda;
    T* dst = Uout + (int64_t)r * ldu;
    for (int64_t i = threadIdx.x; i < m; i += 256) dst[i] = col[i] * inv;
    if (V != nullptr && VT != nullptr) {
        const T* vcol = V + (int64_t)j * ldv;
        for (int i = threadIdx.x; i < n; i += 256) VT[r + (int64_t)i * ldvt] = vcol[i];
    }
    if (threadIdx.x == 0) Sout[r] = s;
}

__global__ void zero_u32_kernel(unsigned* p) { p[0] = 0; p[1] = 0; }

// flag |= any |G_ij| > tol * sqrt(G_ii G_jj), i < j   (G = A^T A)
template <typename T>
__global__ void gram_offdiag_kernel(int n, const T* __restrict__ G, T tol, unsigned* __restrict__ flag) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx % n, j = idx / n;
    if (i >= j) return;
    const double g = (double)G[i + (int64_t)j * n], gi = (double)G[i + (int64_t)i * n], gj = (double)G[j + (int64_t)j * n];
    if (gi > 0 && gj > 0 && g * g > (double)tol * (double)tol * gi * gj) atomicOr(flag, 1u);
}

// the Gram-matrix verification shared by the two sweep drivers: true iff every off-diagonal cosine of A^T A is <= 4 tol
template <typename T>
int jacobi_verify_converged(rlhip_ctx* c, int m, int n, const T* A, int64_t lda, T tol, unsigned* d_nrot, bool* ok) {
    *ok = false;
    size_t gm = rlhip_ws_mark(c);
    T* G = ws_alloc<T>(c, (size_t)n * n);
    unsigned* flag = d_nrot + 1;
    if (G) {
        int grc = gemm<T>(c, 1, 0, n, n, m, T(1), A, lda, A, lda, T(0), G, n);
        if (!grc) {
            hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, c->stream, d_nrot);
            // (4 tol: an entry of the GEMM-computed Gram matrix carries a rounding error of the order of tol = sqrt(m) eps itself.  Tested at
            // tol the check failed on that noise for every flat-spectrum factor -- one hand-back, one Gram matrix and one sweep that then
            // rotated nothing: 0.23 ms of the 3.66 ms device SVD of the RSVD tail.  The sweeps' own rotation criterion stays at tol.)
            hipLaunchKernelGGL(gram_offdiag_kernel<T>, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, c->stream, n, G, (T)(4 * tol), flag);
            RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 16, d_nrot, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            RLHIP_CHECK(rlhip_stream_sync(c));
            *ok = (*((unsigned*)(c->h_mail + 16) + 1) == 0u);
        }
    }
    rlhip_ws_release(c, gm);
    return 0;
}

// Clock holders: bursts of FMAs with 4 naps between on every CU the workers leave idle.  A context that SHARES the device with
// another stream (a side context, or one that adopted a caller's stream: avoid_persistent) launches the workers alone -- holders would
// take CUs from the other stream's kernels and make this launch wait for the whole device.
struct JpHold { int mode, nap, delay, wgs; };
static JpHold jp_hold(const rlhip_ctx* c) {
    return (c->avoid_persistent || c->opt[RLHIP_OPT_JACOBI_CLOCK_HOLDERS] == 0) ? JpHold{0, 4, 0, 0} : JpHold{1, 4, 0, 1 << 20};
}

// clears the flag words and enqueues ONE persistent launch; g.A / lda / trans_upper / skip / sweep0 / max_sweeps / tol / out are the caller's
template <typename T, int QW>
int jp_launch_qw(rlhip_ctx* c, JpArgs<T>& g, unsigned long long* buf, int m, int NBk) {
    constexpr int JB = 16, JMT = JM;
    const int NW = NBk / 2;
    const size_t xwords = (size_t)NBk * JB * m;
    const JpHold h = jp_hold(c);
    g.m = m; g.NB = NBk; g.X = buf; g.bflag = buf + xwords; g.sflag = g.bflag + NBk; g.done = g.sflag + 8 * (size_t)NW;
    g.hold_mode = h.mode; g.hold_nap = h.nap; g.hold_delay_us = h.delay;
    g.census = g.done + 1;
    // the whole device when the clock holders are wanted and fit (one 1024-thread workgroup per CU), else the workers alone
    const unsigned grid_hold = (h.mode && c->num_cu > NW) ? (unsigned)(NW + (h.wgs < c->num_cu - NW ? h.wgs : c->num_cu - NW)) : (unsigned)NW;
    // same-XCD hand-over (kernel comment): needs the launch to cover workgroups 0, 8, ..., 8 (NW - 1) and a CU per worker inside one XCD
    unsigned long long* loc = (grid_hold >= 8u * (unsigned)NW && NW <= 16 && c->opt[RLHIP_OPT_JACOBI_PERSIST] != 2)
                                  ? (unsigned long long*)rlhip_xloc_buffer(c, (xwords + (size_t)NBk) * sizeof(unsigned long long)) : nullptr;
    g.local_try = loc ? 1 : 0; g.XL = loc; g.bflagL = loc ? loc + xwords : nullptr;
    hipError_t e1 = hipMemsetAsync(g.bflag, 0, ((size_t)NBk + 8 * (size_t)NW + 1 + (size_t)NW) * sizeof(unsigned long long), c->stream);
    if (e1 == hipSuccess && loc) e1 = hipMemsetAsync(g.bflagL, 0, (size_t)NBk * sizeof(unsigned long long), c->stream);
    if (e1 == hipSuccess) e1 = hipMemsetAsync(g.out, 0, 16 * sizeof(int), c->stream)     /* (out has >= 16 ints: the Gram route keeps its defect word behind the 8 of this launch) */;
    if (e1 != hipSuccess) return RLHIP_ERR_HIP(e1);
    constexpr int smem = 2 * JB * JMT * (int)sizeof(T);
    RLHIP_FUNC_LDS(c, (jacobi_persist_kernel<T, JB, JMT, QW>), smem);
    void* kargs[] = {(void*)&g};
    if (c->opt[RLHIP_OPT_JACOBI_PERSIST] == 3) {
        // profiling route: the workers alone through an ORDINARY launch (rocprofv3 --pmc aborts on cooperative launches).  NW <= 16 workgroups of
        // an otherwise idle device are resident together without the runtime's guarantee; same kernel, same protocol (uncached hand-over).
        g.local_try = 0; g.hold_mode = 0;
        hipLaunchKernelGGL((jacobi_persist_kernel<T, JB, JMT, QW>), dim3((unsigned)NW), dim3(1024), smem, c->stream, g);
        const hipError_t pe = hipGetLastError();
        return pe == hipSuccess ? 0 : 1;
    }
    hipError_t le = hipLaunchCooperativeKernel((const void*)jacobi_persist_kernel<T, JB, JMT, QW>, dim3(grid_hold), dim3(1024), kargs, (unsigned)smem, c->stream);
    if (le != hipSuccess && grid_hold != (unsigned)NW) {
        (void)hipGetLastError();
        g.local_try = 0;                               // (the workers alone are workgroups 0 .. NW - 1)
        le = hipLaunchCooperativeKernel((const void*)jacobi_persist_kernel<T, JB, JMT, QW>, dim3((unsigned)NW), dim3(1024), kargs, (unsigned)smem, c->stream);
    }
    if (le != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
}

template <typename T>
int jp_launch(rlhip_ctx* c, JpArgs<T>& g, unsigned long long* buf, int m, int NBk) {
    return jp_launch_qw<T, 16>(c, g, buf, m, NBk);     // a quarter wave per column pair (a half wave was measured slower: 10.98 against 10.88 ms per shard step)
}

// Sweeps of the persistent kernel (V not accumulated).  Returns 0 and the number of sweeps when it ran to a verdict, 1 when the path is
// not available or reported a lost word -- A then still holds a valid (possibly partially swept) matrix and the caller continues with the
// per-launch sweeps.
template <typename T>
int persistent_jacobi_sweeps(rlhip_ctx* c, int m, int n, T* A, int64_t lda, T tol, unsigned* d_nrot, int max_sweeps, int* sweeps_out, bool* done) {
    constexpr int JB = 16;
    *done = false;
    int NBk = (n + JB - 1) / JB;
    if (NBk < 2) NBk = 2;
    if (NBk % 2) ++NBk;
    const int NW = NBk / 2;
    if (NW > c->num_cu || NW > 512) return 1;
    const size_t xwords = (size_t)NBk * JB * m, words = xwords + (size_t)NBk + 8 * (size_t)NW + 1 + (size_t)NW;
    const int hold = jp_hold(c).mode;
    unsigned long long* buf = (unsigned long long*)rlhip_xchg_buffer(c, words * sizeof(unsigned long long));
    size_t mark = rlhip_ws_mark(c);
    int* out = ws_alloc<int>(c, 32);
    if (!buf || !out) { rlhip_ws_release(c, mark); return 1; }
    int sweep = *sweeps_out;
    while (sweep < max_sweeps) {
        JpArgs<T> g;
        g.n = n; g.sweep0 = sweep; g.max_sweeps = max_sweeps; g.A = A; g.lda = lda; g.tol = tol; g.out = out; g.trans_upper = 0; g.skip = nullptr; g.norm_ratio_lim = 0.f;
        const int lrc = jp_launch<T>(c, g, buf, m, NBk);
        if (lrc) { rlhip_ws_release(c, mark); return lrc; }
        hipError_t e2 = hipMemcpyAsync(c->h_mail + 16, out, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e2 == hipSuccess) e2 = rlhip_stream_sync(c);
        if (e2 != hipSuccess) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(e2); }
        const int status = *(int*)(c->h_mail + 16), done_sweeps = *((int*)(c->h_mail + 16) + 1), any_lost = *((int*)(c->h_mail + 16) + 2);
        {
            static int want_clk = -1;
            if (want_clk < 0) { const char* e = getenv("RLHIP_JACOBI_CLOCK"); want_clk = e ? atoi(e) : 0; }
            const unsigned long long* tk = reinterpret_cast<const unsigned long long*>((const int*)(c->h_mail + 16) + 4);
            if (want_clk && tk[1]) fprintf(stderr, "[jacobi clock] %d sweeps, %.1f us at %.0f MHz (hold %d, same-XCD hand-over %d)\n", done_sweeps - sweep, (double)tk[1] / 100.0, (double)tk[0] / ((double)tk[1] / 100.0), hold, *((const int*)(c->h_mail + 16) + 3));
        }
        if (*((const int*)(c->h_mail + 16) + 3) == 1) c->path_count[15]++;     // the workers shared one XCD and handed their blocks over through its L2
        if ((status != 1 && status != 2 && status != 3) || any_lost) { rlhip_ws_release(c, mark); *sweeps_out = sweep; return 1; }   // -7, a lost word anywhere, or nothing written: A untouched by this launch
        int rc = rlhip::lacpy<T>(c, 2, m, n, reinterpret_cast<const T*>(buf), m, A, lda);
        if (rc) { rlhip_ws_release(c, mark); return rc < 0 ? rc : 1; }
        sweep = done_sweeps;
        if (status == 1) { *done = true; break; }
        if (status == 2) {
            bool ok = false;
            jacobi_verify_converged<T>(c, m, n, A, lda, tol, d_nrot, &ok);
            if (ok) { *done = true; break; }
            continue;                                  // clustered singular values: tiny cosines, large angles -- keep sweeping
        }
        break;                                         // sweep limit
    }
    rlhip_ws_release(c, mark);
    *sweeps_out = sweep;
    return 0;
}

template <typename T, int JB, int JMT, int QW>
int block_jacobi_sweeps_qw(rlhip_ctx* c, int m, int n, T* A, int64_t lda, T* V, T tol, unsigned* d_nrot, int max_sweeps, int* sweeps_out) {
    constexpr int JP = 2 * JB;
    int NBk = (n + JB - 1) / JB;
    if (NBk < 2) NBk = 2;
    if (NBk % 2) ++NBk;
    constexpr int smem = (JP * JMT + JP * JP) * (int)sizeof(T);
    constexpr int NT = (JMT == 256) ? 1024 : 512;
    RLHIP_FUNC_LDS(c, (jacobi_block_kernel<T, JB, JMT, QW>), smem);
    int sweep = *sweeps_out;                           // sweeps already done by the persistent kernel (0 otherwise)
    if (NBk == 2 && sweep < max_sweeps) {
        // the whole matrix is one workgroup's panel: every sweep inside ONE launch, one host read
        RLHIP_CHECK(hipMemsetAsync(d_nrot, 0, 4 * sizeof(unsigned), c->stream));
        hipLaunchKernelGGL((jacobi_block_kernel<T, JB, JMT, QW>), dim3(1), dim3(NT), smem, c->stream, m, n, NBk, 0, 1, A, lda, V, (int64_t)n, tol, d_nrot,
                           max_sweeps - sweep);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 16, d_nrot, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        const unsigned* hw = (const unsigned*)(c->h_mail + 16);
        *sweeps_out = sweep + (int)hw[2];
        return 0;                                      // (hw[0] != 0: the sweep limit was reached, as the loop below would leave it)
    }
    for (; sweep < max_sweeps; ++sweep) {
        hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, c->stream, d_nrot);
        hipLaunchKernelGGL((jacobi_block_kernel<T, JB, JMT, QW>), dim3(NBk / 2), dim3(NT), smem, c->stream, m, n, NBk, 0, 1, A, lda, V,
                           (int64_t)n, tol, d_nrot, 0);
        for (int oround = 0; oround < NBk - 1; ++oround)
            hipLaunchKernelGGL((jacobi_block_kernel<T, JB, JMT, QW>), dim3(NBk / 2), dim3(NT), smem, c->stream, m, n, NBk, oround, 0, A, lda,
                               V, (int64_t)n, tol, d_nrot, 0);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 16, d_nrot, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        const unsigned nrot = *(unsigned*)(c->h_mail + 16);
        float cos2;
        memcpy(&cos2, (const char*)(c->h_mail + 16) + sizeof(unsigned), sizeof(float));
        if (nrot == 0) { ++sweep; break; }
        // Every cosine met in this sweep was <= 1e-9: for separated singular values the rotations just applied leave cosines
        // of order n * 1e-18 (quadratic convergence) and a further all-idle sweep would only confirm it.  For CLUSTERED
        // singular values that argument fails (tiny cosines still rotate by large angles), so the claim is VERIFIED with one
        // Gram matrix (2 launches instead of a 16-launch sweep): converged iff every off-diagonal cosine of A^T A is <= tol.
        if (cos2 <= 1e-18f) {
            bool ok = false;
            jacobi_verify_converged<T>(c, m, n, A, lda, tol, d_nrot, &ok);
            if (ok) { ++sweep; break; }
        }
    }
    *sweeps_out = sweep;
    return 0;
}

template <typename T, int JB, int JMT>
int block_jacobi_sweeps(rlhip_ctx* c, int m, int n, T* A, int64_t lda, T* V, T tol, unsigned* d_nrot, int max_sweeps, int* sweeps_out) {
    // (32-wide blocks with 32 lanes per pair would need 1024 rotating threads AND leave no wave to spare: they keep the quarter-wave layout)
    return block_jacobi_sweeps_qw<T, JB, JMT, 16>(c, m, n, A, lda, V, tol, d_nrot, max_sweeps, sweeps_out);
}

}  // namespace

namespace rlhip {

template <typename T>
int lacpy(rlhip_ctx* c, int uplo, int64_t m, int64_t n, const T* A, int64_t lda, T* B, int64_t ldb);
template <typename T>
int laset(rlhip_ctx* c, int uplo, int64_t m, int64_t n, T offdiag, T diag, T* A, int64_t lda);

// ENQUEUES the one-launch Jacobi sweeps of an n x n matrix X -- trans_upper = 0: X = the matrix stored in `R` (e.g. a full symmetric Gram
// matrix); 1: X = R^T of the upper triangle stored in `R` (a Cholesky factor as potrf left it) -- and returns without touching the host: the swept columns (X J, mutually orthogonal, norms = singular values) land in the
// context's exchange buffer (*X_out, column-major, leading dimension n), status / sweeps / lost flag in out_dev[0..2] (see JpArgs::out).
// `skip_dev` != nullptr: the launch does nothing (status -8) when that device word is non-zero.  Returns 0 when enqueued, 1 when the
// path is not available (the caller takes another route), < 0 on error.  fp64, 32 < n <= 256.
template <typename T>
int jacobi_enqueue_rt(rlhip_ctx* c, int n, const T* R, int64_t ldr, int trans_upper, float norm_ratio_lim, const int* skip_dev, int* out_dev, const T** X_out) {
    if constexpr (sizeof(T) != 8) { return 1; }
    else {
        constexpr int JB = 16;
        if (n <= 32 || n > JM) return 1;
        if (c->opt[RLHIP_OPT_JACOBI_PERSIST] == 0) return 1;
        int NBk = (n + JB - 1) / JB;
        if (NBk % 2) ++NBk;
        const int NW = NBk / 2;
        if (NW > c->num_cu) return 1;
        const size_t xwords = (size_t)NBk * JB * n, words = xwords + (size_t)NBk + 8 * (size_t)NW + 1 + (size_t)NW;
        unsigned long long* buf = (unsigned long long*)rlhip_xchg_buffer(c, words * sizeof(unsigned long long));
        if (!buf) return 1;
        JpArgs<T> g;
        g.n = n; g.sweep0 = 0; g.max_sweeps = 60; g.A = R; g.lda = ldr; g.out = out_dev; g.trans_upper = trans_upper; g.skip = skip_dev; g.norm_ratio_lim = norm_ratio_lim;
        g.tol = std::sqrt((T)n) * std::numeric_limits<T>::epsilon();
        const int rc = jp_launch<T>(c, g, buf, n, NBk);
        if (rc) return rc;
        *X_out = reinterpret_cast<const T*>(buf);
        c->path_count[6]++;
        return 0;
    }
}
template int jacobi_enqueue_rt<double>(rlhip_ctx*, int, const double*, int64_t, int, float, const int*, int*, const double**);
template int jacobi_enqueue_rt<float>(rlhip_ctx*, int, const float*, int64_t, int, float, const int*, int*, const float**);

template <typename T>
int gesvdj(rlhip_ctx* c, int64_t m, int64_t n64, T* A, int64_t lda, T* S, T* VT, int64_t ldvt,
           int* sweeps_host) {
    if (m < 0) return -2;
    if (n64 < 0) return -3;
    if (m < n64) return -2;  // tall only (the path's factor is n x k with n >= k)
    if (lda < (m > 1 ? m : 1)) return -5;
    if (VT != nullptr && ldvt < (n64 > 1 ? n64 : 1)) return -8;   // VT == nullptr: singular values and left vectors only
    if (sweeps_host) *sweeps_host = 0;
    if (n64 == 0) return 0;
    const int n = (int)n64;
    const int N = (n % 2) ? n + 1 : n;
    if constexpr (sizeof(T) == 4) {
        // fp32 problems run the fp64 kernels on a widened copy.  Short factors (the k x k matrix of an fp32 RSVD / ABRIK tail) then get the
        // LDS-resident block Jacobi (3.5 ms at n = 256 against 25 ms for 255 per-round launches x ~11 sweeps), and tall ones lose the
        // drift of thousands of fp32 rotations (measured at 3000 x 256: ||V^T V - I|| = 7e3 eps32 in fp32 arithmetic, 5 eps32 widened).
        if (n > 1) {
            size_t mk = rlhip_ws_mark(c);
            double* Ad = ws_alloc<double>(c, (size_t)m * n);
            double* Sd = ws_alloc<double>(c, (size_t)n);
            double* VTd = (VT != nullptr) ? ws_alloc<double>(c, (size_t)n * n) : nullptr;
            if (!Ad || !Sd || (VT != nullptr && !VTd)) { rlhip_ws_release(c, mk); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
            const unsigned gA = (unsigned)((m * n + 255) / 256), gV = (unsigned)(((int64_t)n * n + 255) / 256);
            hipLaunchKernelGGL((convert_kernel<float, double>), dim3(gA), dim3(256), 0, c->stream, m, (int64_t)n, A, lda, Ad, m);
            int info = gesvdj<double>(c, m, n64, Ad, m, Sd, VTd, (int64_t)n, sweeps_host);
            if (info >= 0) {
                hipLaunchKernelGGL((convert_kernel<double, float>), dim3(gA), dim3(256), 0, c->stream, m, (int64_t)n, Ad, m, A, lda);
                hipLaunchKernelGGL((convert_kernel<double, float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (int64_t)n, (int64_t)1, Sd, (int64_t)n, S, (int64_t)n);
                if (VT != nullptr)
                    hipLaunchKernelGGL((convert_kernel<double, float>), dim3(gV), dim3(256), 0, c->stream, (int64_t)n, (int64_t)n, VTd, (int64_t)n, VT, ldvt);
                RLHIP_LAUNCH_CHECK();
            }
            rlhip_ws_release(c, mk);
            return info;
        }
    }
    size_t mark = rlhip_ws_mark(c);
    T* V = (VT != nullptr) ? ws_alloc<T>(c, (size_t)n * n) : nullptr;
    T* W = ws_alloc<T>(c, (size_t)m * n);
    T* Sraw = ws_alloc<T>(c, (size_t)n);
    int* rank = ws_alloc<int>(c, (size_t)n);
    if ((VT != nullptr && !V) || !W || !Sraw || !rank) { rlhip_ws_release(c, mark); return RLHIP_ERR_HIP(hipErrorOutOfMemory); }
    unsigned* d_nrot = (unsigned*)(c->d_mail + 16);
    int rc = V ? laset<T>(c, 2, n, n, T(0), T(1), V, n) : 0;
    if (rc) { rlhip_ws_release(c, mark); return rc; }
    const T tol = std::sqrt((T)m) * std::numeric_limits<T>::epsilon();
    const int max_sweeps = 60;
    int sweep = 0;
    int info = 0;
    if (n > 1 && m <= 2 * JM && sizeof(T) == 8) {
        // LDS-resident block Jacobi (see jacobi_block_kernel)
        constexpr int jb_sel = 16;         // block width (32: half the launches, four times the pairs per launch -- measured slower)
        const bool persist = c->opt[RLHIP_OPT_JACOBI_PERSIST] != 0;
        bool done = false;
        if constexpr (sizeof(T) == 8) {
            // singular values / left vectors only and at most 256 rows: all sweeps in one resident launch (see jacobi_persist_kernel)
            if (persist && V == nullptr && m <= JM && jb_sel == 16 && n > 32) {
                const int prc = persistent_jacobi_sweeps<T>(c, (int)m, n, A, lda, tol, d_nrot, max_sweeps, &sweep, &done);
                if (prc < 0) { rlhip_ws_release(c, mark); return prc; }
                if (prc == 0) c->path_count[6]++;
            }
        }
        if (done || sweep >= max_sweeps) rc = 0;
        else if (m > JM) rc = block_jacobi_sweeps<T, 16, 2 * JM>(c, (int)m, n, A, lda, V, tol, d_nrot, max_sweeps, &sweep);   // 257 .. 512 rows: 32 x 512 panel = 128 KiB
        else if (jb_sel == 32 || n <= 32) rc = block_jacobi_sweeps<T, 32, JM>(c, (int)m, n, A, lda, V, tol, d_nrot, max_sweeps, &sweep);
        else rc = block_jacobi_sweeps<T, 16, JM>(c, (int)m, n, A, lda, V, tol, d_nrot, max_sweeps, &sweep);
        if (rc) { rlhip_ws_release(c, mark); return rc; }
        if (sweep >= max_sweeps) info = 1;
    } else if (n > 1) {
        for (; sweep < max_sweeps; ++sweep) {
            hipLaunchKernelGGL(zero_u32_kernel, dim3(1), dim3(1), 0, c->stream, d_nrot);
            for (int round = 0; round < N - 1; ++round) {
                hipLaunchKernelGGL(jacobi_round_kernel<T>, dim3(N / 2), dim3(256), 0, c->stream, m, n, N, round, A,
                                   lda, V, (int64_t)n, tol, d_nrot);
            }
            RLHIP_LAUNCH_CHECK();
            RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 16, d_nrot, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            RLHIP_CHECK(rlhip_stream_sync(c));
            unsigned nrot = *(unsigned*)(c->h_mail + 16);
            if (nrot == 0) { ++sweep; break; }
        }
        if (sweep >= max_sweeps) info = 1;
    }
    if (sweeps_host) *sweeps_host = sweep;
    hipLaunchKernelGGL(colnorm_kernel<T>, dim3(n), dim3(256), 0, c->stream, m, A, lda, Sraw);
    hipLaunchKernelGGL(rank_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, Sraw, rank);
    hipLaunchKernelGGL(finalize_kernel<T>, dim3(n), dim3(256), 0, c->stream, m, n, A, lda, V, (int64_t)n, Sraw, rank,
                       W, m, S, VT, ldvt);
    RLHIP_LAUNCH_CHECK();
    rc = lacpy<T>(c, 2, m, n, W, m, A, lda);
    rlhip_ws_release(c, mark);
    return rc ? rc : info;
}

template int gesvdj<double>(rlhip_ctx*, int64_t, int64_t, double*, int64_t, double*, double*, int64_t, int*);
template int gesvdj<float>(rlhip_ctx*, int64_t, int64_t, float*, int64_t, float*, float*, int64_t, int*);

}  // namespace rlhip
