// The RandBLAS names the hot path needs (RNGState, DenseDist, ScalarDist, fill_dense), device flavour.
// RandBLAS is an absent, un-vendored dependency of the reference; the random stream is
// this library's own (defined in randlapack_amd/csrc/fill.hip) -- "parity unpinned" for the stream,
// parity from the sketch onward.  State threading follows the reference exactly:
//     state = RandBLAS::fill_dense(D, buf, state)        (comps/rl_rs.hh:135,139)
#pragma once
#include <array>
#include <cstdint>
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
RNGState<RNG> fill_dense(DenseDist const& D, double* buf, RNGState<RNG> const& st, blas::Queue& q) {
    RNGState<RNG> next = st;
    blas::check(rlhip_fill_dense_f64(q.ctx(), D.family == ScalarDist::Gaussian ? 0 : 1, D.n_rows, D.n_cols, buf,
                                     st.counter.data(), st.key.data(), next.counter.data()), "fill_dense");
    return next;
}
template <typename RNG>
RNGState<RNG> fill_dense(DenseDist const& D, float* buf, RNGState<RNG> const& st, blas::Queue& q) {
    RNGState<RNG> next = st;
    blas::check(rlhip_fill_dense_f32(q.ctx(), D.family == ScalarDist::Gaussian ? 0 : 1, D.n_rows, D.n_cols, buf,
                                     st.counter.data(), st.key.data(), next.counter.data()), "fill_dense");
    return next;
}

}  // namespace RandBLAS
