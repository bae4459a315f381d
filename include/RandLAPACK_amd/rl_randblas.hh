// The RandBLAS names the hot path needs (RNGState, DenseDist, ScalarDist, fill_dense), device flavour.
// RandBLAS is an absent, un-vendored dependency of the reference; the random stream is
// this library's own (defined in randlapack_amd/csrc/fill.hip) -- "parity unpinned" for the stream,
// parity from the sketch onward.  State threading follows the reference exactly:
//     state = RandBLAS::fill_dense(D, buf, state)        (comps/rl_rs.hh:135,139)
#pragma once
#include <array>
#include <cstdint>
#include <unordered_map>
#include <vector>
#include "rl_blaspp.hh"

namespace r123 {
struct Philox4x32 {
    using ctr_type = std::array<uint32_t, 4>;
    using key_type = std::array<uint32_t, 2>;
};
}  // namespace r123

namespace RandBLAS {

using DefaultRNG = r123::Philox4x32;

template <typename RNG = DefaultRNG>
struct RNGState {
    typename RNG::ctr_type counter{};
    typename RNG::key_type key{};
    RNGState() = default;
    explicit RNGState(uint32_t seed) { key[0] = seed; }
};

enum class ScalarDist : char { Gaussian = 'G', Uniform = 'U' };

struct DenseDist {
    int64_t n_rows, n_cols;
    ScalarDist family;
    DenseDist(int64_t r, int64_t c, ScalarDist f = ScalarDist::Gaussian) : n_rows(r), n_cols(c), family(f) {}
};

// Fills the n_rows x n_cols column-major DEVICE buffer (ld = n_rows) and returns the advanced state.
template <typename RNG>
RNGState<RNG> fill_dense(DenseDist const& D, double* buf, RNGState<RNG> const& st, blas::Queue& q = blas::default_queue()) {
    RNGState<RNG> next = st;
    blas::check(rlhip_fill_dense_f64(q.ctx(), D.family == ScalarDist::Gaussian ? 0 : 1, D.n_rows, D.n_cols, buf,
                                     st.counter.data(), st.key.data(), next.counter.data()), "fill_dense");
    return next;
}
template <typename RNG>
RNGState<RNG> fill_dense(DenseDist const& D, float* buf, RNGState<RNG> const& st, blas::Queue& q = blas::default_queue()) {
    RNGState<RNG> next = st;
    blas::check(rlhip_fill_dense_f32(q.ctx(), D.family == ScalarDist::Gaussian ? 0 : 1, D.n_rows, D.n_cols, buf,
                                     st.counter.data(), st.key.data(), next.counter.data()), "fill_dense");
    return next;
}

// Rows [row0, row0 + loc_rows) of the D.n_rows x D.n_cols operator fill_dense(D, ...) produces, into buf (ld = loc_rows).
// The returned state is the one the full fill returns.
template <typename RNG>
RNGState<RNG> fill_dense_rows(DenseDist const& D, int64_t row0, int64_t loc_rows, double* buf, RNGState<RNG> const& st, blas::Queue& q = blas::default_queue()) {
    RNGState<RNG> next = st;
    blas::check(rlhip_fill_dense_rows_f64(q.ctx(), D.family == ScalarDist::Gaussian ? 0 : 1, D.n_rows, D.n_cols, row0, loc_rows, buf,
                                          loc_rows > 0 ? loc_rows : 1, st.counter.data(), st.key.data(), next.counter.data()), "fill_dense_rows");
    return next;
}
template <typename RNG>
RNGState<RNG> fill_dense_rows(DenseDist const& D, int64_t row0, int64_t loc_rows, float* buf, RNGState<RNG> const& st, blas::Queue& q = blas::default_queue()) {
    RNGState<RNG> next = st;
    blas::check(rlhip_fill_dense_rows_f32(q.ctx(), D.family == ScalarDist::Gaussian ? 0 : 1, D.n_rows, D.n_cols, row0, loc_rows, buf,
                                          loc_rows > 0 ? loc_rows : 1, st.counter.data(), st.key.data(), next.counter.data()), "fill_dense_rows");
    return next;
}

// k distinct indices from {0..n-1}, r independent repetitions (RandBLAS::repeated_fisher_yates as used at testing/rl_gen.hh:269).
// idxs: HOST buffer of k * r.  Own stream: repetition i draws its k swap targets from Philox words ctr + i*ceil(k/4) ...;
// swap j picks uniformly from [j, n) by 32x32 -> 64-bit multiply-shift.  Returns the advanced state.
template <typename RNG>
RNGState<RNG> repeated_fisher_yates(int64_t k, int64_t n, int64_t r, int64_t* idxs, RNGState<RNG> const& st, blas::Queue& q = blas::default_queue()) {
    if (k > n) throw blas::Error("repeated_fisher_yates: k > n");
    RNGState<RNG> next = st;
    if (k <= 0 || r <= 0) return next;
    const int64_t blocks_per_rep = (k + 3) / 4, nblk = blocks_per_rep * r;
    std::vector<uint32_t> words((size_t)(4 * nblk));
    {
        blas::Scratch ws(q);
        uint32_t* dev = ws.alloc<uint32_t>(4 * nblk);
        blas::check(rlhip_philox4x32_10(q.ctx(), nblk, dev, st.counter.data(), st.key.data()), "philox");
        blas::copy_to_host(4 * nblk, dev, words.data(), q);
    }
    for (int64_t rep = 0; rep < r; ++rep) {
        std::unordered_map<int64_t, int64_t> moved;            // sparse view of the partially shuffled identity
        auto at = [&](int64_t i) { auto it = moved.find(i); return it == moved.end() ? i : it->second; };
        for (int64_t j = 0; j < k; ++j) {
            const uint64_t w = words[(size_t)(4 * rep * blocks_per_rep + j)];
            const int64_t t = j + (int64_t)((w * (uint64_t)(n - j)) >> 32);
            const int64_t vj = at(j), vt = at(t);
            moved[t] = vj;
            moved[j] = vt;
            idxs[rep * k + j] = vt;
        }
    }
    uint64_t lo = ((uint64_t)st.counter[1] << 32) | st.counter[0], hi = ((uint64_t)st.counter[3] << 32) | st.counter[2];
    const uint64_t nlo = lo + (uint64_t)nblk;
    if (nlo < lo) hi += 1;
    next.counter = {(uint32_t)nlo, (uint32_t)(nlo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    return next;
}

// Dense sketching operator (RandBLAS::DenseSkOp as used at drivers/rl_cqrrt_linops.hh:196-200): owns its n_rows x n_cols
// column-major DEVICE buffer once fill_dense(S) has run.
template <typename T, typename RNG = DefaultRNG>
struct DenseSkOp {
    DenseDist dist;
    RNGState<RNG> seed_state, next_state;
    T* buff = nullptr;
    blas::Queue& q;
    DenseSkOp(DenseDist const& D, RNGState<RNG> const& st, blas::Queue& queue) : dist(D), seed_state(st), next_state(st), q(queue) {
        // the state a fill advances to depends on (dist, seed) only: compute it without touching memory
        blas::check(rlhip_fill_dense_rows_f64(q.ctx(), D.family == ScalarDist::Gaussian ? 0 : 1, D.n_rows, D.n_cols, 0, 0, nullptr, 1,
                                              st.counter.data(), st.key.data(), next_state.counter.data()), "DenseSkOp");
    }
    DenseSkOp(DenseSkOp const&) = delete;
    DenseSkOp& operator=(DenseSkOp const&) = delete;
    ~DenseSkOp() { if (buff) blas::device_free(buff, q); }
};
template <typename T, typename RNG>
void fill_dense(DenseSkOp<T, RNG>& S) {
    if (S.buff) return;
    S.buff = blas::device_malloc<T>(S.dist.n_rows * S.dist.n_cols, S.q);
    fill_dense(S.dist, S.buff, S.seed_state, S.q);
}

// ---- sparse sketching operator (short-axis-sparse), cf. RandBLAS::SparseDist / SparseSkOp / sketch_general as used at
//      drivers/rl_cqrrpt.hh:214-222.  n_rows = d (sketch dimension), n_cols = m, vec_nnz nonzeros per column.
struct SparseDist {
    int64_t n_rows, n_cols, vec_nnz;
    // extension (not in RandBLAS): -1 library default (independent columns, RandBLAS's distribution), 1 independent columns,
    // 0 block affine (one affine row map per block of n_rows columns: faster apply, see rlhip.h)
    int structure = -1;
    SparseDist(int64_t r, int64_t c, int64_t nnz) : n_rows(r), n_cols(c), vec_nnz(nnz) {}
};

template <typename T, typename RNG = DefaultRNG>
struct SparseSkOp {
    SparseDist dist;
    RNGState<RNG> seed_state, next_state;
    rlhip_saso* handle = nullptr;
    blas::Queue& q;
    SparseSkOp(SparseDist const& D, RNGState<RNG> const& st, blas::Queue& queue) : dist(D), seed_state(st), next_state(st), q(queue) {
        blas::check(rlhip_saso_create_mode(q.ctx(), D.n_rows, D.n_cols, (int)D.vec_nnz, D.structure, st.counter.data(), st.key.data(),
                                           next_state.counter.data(), &handle), "saso_create");
    }
    SparseSkOp(SparseSkOp const&) = delete;
    SparseSkOp& operator=(SparseSkOp const&) = delete;
    ~SparseSkOp() { if (handle) rlhip_saso_destroy(q.ctx(), handle); }
};

// B (d x n) = alpha * S * A (m x n) + beta * B  -- the only sketch_general shape on the path (left sketch, no offsets)
template <typename RNG>
void sketch_general(blas::Layout, blas::Op opS, blas::Op opA, int64_t d, int64_t n, int64_t m, double alpha,
                    SparseSkOp<double, RNG>& S, int64_t ro, int64_t co, double const* A, int64_t lda, double beta, double* B,
                    int64_t ldb, blas::Queue& q = blas::default_queue()) {
    if (opS != blas::Op::NoTrans || opA != blas::Op::NoTrans || ro != 0 || co != 0 || d != S.dist.n_rows || m != S.dist.n_cols)
        throw blas::Error("sketch_general: only the plain left sketch S*A is on the path");
    blas::check(rlhip_saso_apply_f64(q.ctx(), S.handle, n, alpha, A, lda, beta, B, ldb), "saso_apply");
}
template <typename RNG>
void sketch_general(blas::Layout, blas::Op opS, blas::Op opA, int64_t d, int64_t n, int64_t m, float alpha,
                    SparseSkOp<float, RNG>& S, int64_t ro, int64_t co, float const* A, int64_t lda, float beta, float* B,
                    int64_t ldb, blas::Queue& q = blas::default_queue()) {
    if (opS != blas::Op::NoTrans || opA != blas::Op::NoTrans || ro != 0 || co != 0 || d != S.dist.n_rows || m != S.dist.n_cols)
        throw blas::Error("sketch_general: only the plain left sketch S*A is on the path");
    blas::check(rlhip_saso_apply_f32(q.ctx(), S.handle, n, alpha, A, lda, beta, B, ldb), "saso_apply");
}

// one row shard's contribution to S * A: B = alpha * S[:, row0 : row0 + mloc] * A_loc + beta * B  (S built for the global row count)
template <typename RNG>
void sketch_rows(SparseSkOp<double, RNG>& S, int64_t n, double alpha, double const* A_loc, int64_t lda, int64_t row0, int64_t mloc,
                 double beta, double* B, int64_t ldb, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_saso_apply_rows_f64(q.ctx(), S.handle, n, alpha, A_loc, lda, row0, mloc, beta, B, ldb), "saso_apply_rows");
}
template <typename RNG>
void sketch_rows(SparseSkOp<float, RNG>& S, int64_t n, float alpha, float const* A_loc, int64_t lda, int64_t row0, int64_t mloc,
                 float beta, float* B, int64_t ldb, blas::Queue& q = blas::default_queue()) {
    blas::check(rlhip_saso_apply_rows_f32(q.ctx(), S.handle, n, alpha, A_loc, lda, row0, mloc, beta, B, ldb), "saso_apply_rows");
}

// ---- sparse data matrices (RandBLAS::sparse_data::CSRMatrix as the reference's ABRIK / SparseLinOp take them, rl_abrik.hh:146-162,
//      rl_sparse_linop.hh): a NON-OWNING view of device CSR arrays with int64 indices.  Field names as in RandBLAS (n_rows, n_cols, nnz,
//      vals, rowptr, colidxs).  A CSC matrix is passed as the CSR of its transpose.
namespace sparse_data {
template <typename T, typename sint_t = int64_t>
struct CSRMatrix {
    using scalar_t = T;
    using index_t = sint_t;
    const int64_t n_rows, n_cols;
    int64_t nnz;
    const T* vals;
    const sint_t* rowptr;
    const sint_t* colidxs;
    CSRMatrix(int64_t rows, int64_t cols, int64_t nnz_, const T* v, const sint_t* rp, const sint_t* ci)
        : n_rows(rows), n_cols(cols), nnz(nnz_), vals(v), rowptr(rp), colidxs(ci) {}
};
}  // namespace sparse_data
using sparse_data::CSRMatrix;

}  // namespace RandBLAS
