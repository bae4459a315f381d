// blas:: call surface of the hot path, device flavour (cf. the reference's RandLAPACK/rl_blaspp.hh:3-9, which
// pulls these names from BLAS++).  Every function takes a blas::Queue& last, exactly like BLAS++'s device
// API that the reference's GPU drivers use (drivers/rl_cqrrpt_gpu.hh:303-353, rl_bqrrp_gpu.hh:615-667), and
// forwards to the C ABI in rlhip.h.  All pointers are device pointers.
#pragma once
#include <cstdint>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>
#include "../rlhip.h"

namespace blas {

enum class Layout : char { ColMajor = 'C', RowMajor = 'R' };
enum class Op : char { NoTrans = 'N', Trans = 'T', ConjTrans = 'C' };
enum class Side : char { Left = 'L', Right = 'R' };
enum class Diag : char { NonUnit = 'N', Unit = 'U' };
enum class Uplo : char { Upper = 'U', Lower = 'L', General = 'G' };

inline char to_char(Op v) { return (char)v; }
inline char to_char(Side v) { return (char)v; }
inline char to_char(Diag v) { return (char)v; }
inline char to_char(Uplo v) { return (char)v; }
inline char to_char(Layout v) { return (char)v; }

class Error : public std::runtime_error {
public:
    explicit Error(std::string const& m) : std::runtime_error(m) {}
};

inline void check(int rc, const char* what) {
    if (rc < 0) throw Error(std::string(what) + " failed with code " + std::to_string(rc));
}

// One HIP device + stream + scratch arena.  Mirrors blas::Queue(device) / .sync() / .stream().
class Queue {
    rlhip_ctx* ctx_ = nullptr;
    bool owned_ = false;
public:
    explicit Queue(int device = 0) {
        check(rlhip_create(&ctx_, device, nullptr, 1), "rlhip_create");
        owned_ = true;
    }
    // adopt an existing context (e.g. one bound to PyTorch's stream)
    explicit Queue(rlhip_ctx* ctx) : ctx_(ctx), owned_(false) {}
    // a SIDE queue of `parent`: same device, its own high-priority stream and scratch arena, for work that runs beside the parent's
    // stream (BQRRP's look-ahead).  No communicator: it answers as one rank.  Order the two with wait_for().
    struct Side {};
    Queue(Queue& parent, Side) {
        check(rlhip_create_side(parent.ctx_, &ctx_), "rlhip_create_side");
        owned_ = true;
    }
    // the parent's cached side context, borrowed (created on first use, lives as long as the parent): for callers inside a timed path
    struct CachedSide {};
    Queue(Queue& parent, CachedSide) {
        check(rlhip_side_of(parent.ctx_, &ctx_), "rlhip_side_of");
        owned_ = false;
    }
    // what this queue enqueues from now on starts after what `other` has enqueued so far (device-side ordering, the host does not wait)
    void wait_for(Queue& other) { check(rlhip_order_after(ctx_, other.ctx_), "rlhip_order_after"); }
    // columns per workgroup of the tag-exchange pivoted QR on this queue (0: default); see rlhip_set_qrcp_cols
    void set_qrcp_cols(int cols) { check(rlhip_set_qrcp_cols(ctx_, cols), "rlhip_set_qrcp_cols"); }
    Queue(Queue const&) = delete;
    Queue& operator=(Queue const&) = delete;
    ~Queue() { if (owned_ && ctx_) rlhip_destroy(ctx_); }
    void sync() { check(rlhip_sync(ctx_), "rlhip_sync"); }

    // ---- row-block sharding (one process per GPU).  world() == 1 -> everything below is a no-op.
    int world() const { return local_only ? 1 : rlhip_comm_size(ctx_); }
    int rank() const { return local_only ? 0 : rlhip_comm_rank(ctx_); }
    // > 0: the queue answers as a single rank (LocalOnly below): replicated sub-problems of a sharded driver -- the QRCP of CQRRPT's
    // sketch -- run the ordinary single-device code on every rank and arrive at identical results without an exchange
    int local_only = 0;
    // Drivers mark, around each call that reduces over the row index, whether the operand's rows are the
    // sharded dimension (m-long objects: A, Y, Q, Omega_1) or replicated (n- or k-long objects: Omega, B^T, R).
    // Fused-norm request (QB): the next blas::gemm whose A operand is exactly this matrix also returns ||A||_F
    // (one pass over A instead of two).  Consumed at most once.
    // `defer`: the product does not wait for the norm; collect_norm() fetches it after the caller's next synchronisation (QB's ||B_i||_F).
    struct NormRequest { const void* ptr = nullptr; int64_t rows = 0, cols = 0, ld = 0; bool done = false; double value = 0; bool defer = false, pending = false; };
    NormRequest norm_req;
    // over_ranks: the matrix is row-sharded and the norm of the whole matrix is wanted (the sum rides on CholQRQ's Gram all-reduce when it can)
    double collect_norm(bool over_ranks) {
        if (norm_req.pending) { check(rlhip_norma_collect_f64(ctx_, over_ranks ? 1 : 0, &norm_req.value), "norma_collect"); norm_req.pending = false; }
        else if (over_ranks && world() > 1) { double ssq = norm_req.value * norm_req.value; allreduce_sum_host(&ssq, 1); norm_req.value = std::sqrt(ssq); }
        return norm_req.value;
    }
    bool rows_sharded = false;
    bool reduce_over_rows() const { return rows_sharded && world() > 1; }
    void allreduce_sum(double* buf, int64_t count) { check(rlhip_allreduce_sum_f64(ctx_, buf, count), "allreduce"); }
    void allreduce_sum(float* buf, int64_t count) { check(rlhip_allreduce_sum_f32(ctx_, buf, count), "allreduce"); }
    void allreduce_sum_host(double* x, int64_t n) { check(rlhip_allreduce_sum_host_f64(ctx_, x, n), "allreduce_host"); }
    // this rank's slice of a row-sharded dimension: (global length, first global row) from the local lengths
    void shard_extent(int64_t m_local, int64_t& m_global, int64_t& row0) {
        const int w = world(), r = rank();
        if (w <= 1) { m_global = m_local; row0 = 0; return; }
        std::vector<double> len((size_t)w, 0.0);
        len[(size_t)r] = (double)m_local;
        allreduce_sum_host(len.data(), w);
        m_global = 0; row0 = 0;
        for (int i = 0; i < w; ++i) { if (i < r) row0 += (int64_t)len[(size_t)i]; m_global += (int64_t)len[(size_t)i]; }
    }
    void* stream() const { return rlhip_stream(ctx_); }
    rlhip_ctx* ctx() const { return ctx_; }
};

// The queue the reference's own device drivers construct inside every call (`blas::Queue blas_queue(0)`, rl_bqrrp_gpu.hh:232,
// rl_cqrrpt_gpu.hh:196): device 0, one stream.  Here it is ONE process-wide object, created on first use; every function and
// constructor of this layer takes it when the caller passes no queue, so that code written against the reference's signatures
// (no queue argument) compiles unchanged.
inline Queue& default_queue() {
    static Queue q0(0);
    return q0;
}

template <typename T>
T* device_malloc(int64_t n, Queue& q = blas::default_queue()) {
    void* p = nullptr;
    check(rlhip_malloc(q.ctx(), &p, (size_t)(n > 0 ? n : 1) * sizeof(T)), "device_malloc");
    return (T*)p;
}
inline void device_free(void* p, Queue& q = blas::default_queue()) { check(rlhip_free(q.ctx(), p), "device_free"); }
template <typename T>
void device_memset(T* p, int byte, int64_t n, Queue& q = blas::default_queue()) { check(rlhip_memset(q.ctx(), p, byte, (size_t)n * sizeof(T)), "memset"); }
template <typename T>
void device_copy_vector(int64_t n, T const* src, T* dst, Queue& q = blas::default_queue()) {
    check(rlhip_memcpy_d2d(q.ctx(), dst, src, (size_t)n * sizeof(T)), "device_copy_vector");
}
template <typename T>
void copy_to_host(int64_t n, T const* dev, T* host, Queue& q = blas::default_queue()) { check(rlhip_memcpy_d2h(q.ctx(), host, dev, (size_t)n * sizeof(T)), "d2h"); }
template <typename T>
void copy_to_device(int64_t n, T const* host, T* dev, Queue& q = blas::default_queue()) { check(rlhip_memcpy_h2d(q.ctx(), dev, host, (size_t)n * sizeof(T)), "h2d"); }

// blas::copy(n, x, incx, y, incy) for contiguous vectors (test/comps/test_qb.cc:83-85,144)
template <typename T>
void copy(int64_t n, T const* x, int64_t incx, T* y, int64_t incy, Queue& q = default_queue()) {
    if (incx != 1 || incy != 1) throw Error("blas::copy: unit strides only");
    device_copy_vector(n, x, y, q);
}

// RAII: declare whether row-reductions issued inside the scope run over the sharded dimension
class RowsSharded {
    Queue& q_;
    bool prev_;
public:
    RowsSharded(Queue& q, bool on) : q_(q), prev_(q.rows_sharded) { q.rows_sharded = on; }
    ~RowsSharded() { q_.rows_sharded = prev_; }
};

// RAII: inside the scope the queue answers as ONE rank (no collectives, nothing sharded)
class LocalOnly {
    Queue& q_;
public:
    explicit LocalOnly(Queue& q) : q_(q) { ++q.local_only; }
    ~LocalOnly() { --q_.local_only; }
};

// RAII: a named phase for profilers (roctx range; the reference's NVTX ranges, drivers/rl_bqrrp_gpu.hh:335-403).  Free when no profiler listens.
class Range {
public:
    explicit Range(const char* name) { rlhip_range_push(name); }
    ~Range() { rlhip_range_pop(); }
    Range(Range const&) = delete;
    Range& operator=(Range const&) = delete;
};

// a sequence of phases inside one scope: ph("a") ... ph("b") closes "a" and opens "b"; the destructor (or end()) closes the last one
class Phases {
    bool open_ = false;
public:
    void operator()(const char* name) { end(); rlhip_range_push(name); open_ = true; }
    void end() { if (open_) { rlhip_range_pop(); open_ = false; } }
    ~Phases() { end(); }
};

// RAII: inside the scope the queue launches no kernel that holds every CU until it is done (tiled GEMMs instead of the persistent stream-K
// ones), so that a side queue's kernels run beside its products
class GiveWay {
    Queue& q_;
    int prev_;
public:
    explicit GiveWay(Queue& q) : q_(q), prev_(rlhip_avoid_persistent(q.ctx(), 1)) {}
    ~GiveWay() { rlhip_avoid_persistent(q_.ctx(), prev_ > 0 ? 1 : 0); }
};

// RAII scope over the queue's stream-ordered scratch arena
class Scratch {
    Queue& q_;
    size_t mark_;
public:
    explicit Scratch(Queue& q) : q_(q), mark_(rlhip_scratch_mark(q.ctx())) {}
    ~Scratch() { rlhip_scratch_release(q_.ctx(), mark_); }
    template <typename T>
    T* alloc(int64_t n) {
        void* p = nullptr;
        check(rlhip_scratch_alloc(q_.ctx(), &p, (size_t)(n > 0 ? n : 1) * sizeof(T)), "scratch_alloc");
        return (T*)p;
    }
    /// nullptr instead of an exception when the arena cannot grow by n elements (optional workspaces: the caller falls back)
    template <typename T>
    T* try_alloc(int64_t n) {
        void* p = nullptr;
        if (rlhip_scratch_alloc(q_.ctx(), &p, (size_t)(n > 0 ? n : 1) * sizeof(T)) < 0) return nullptr;
        return (T*)p;
    }
};

// ---- level 3 (ColMajor only, as on the whole reference path)
inline void gemm(Layout, Op ta, Op tb, int64_t m, int64_t n, int64_t k, double alpha, double const* A, int64_t lda,
                 double const* B, int64_t ldb, double beta, double* C, int64_t ldc, Queue& q = blas::default_queue()) {
    auto& nr = q.norm_req;
    if (nr.ptr == (const void*)A && !nr.done && nr.ld == lda &&
        ((ta == Op::NoTrans && nr.rows == m && nr.cols == k) || (ta != Op::NoTrans && nr.rows == k && nr.cols == m))) {
        double nrm = 0;
        check(rlhip_gemm_norma_f64(q.ctx(), (char)ta, (char)tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, nr.defer ? nullptr : &nrm,
                                   nullptr), "gemm_norma");
        nr.done = true;
        nr.pending = nr.defer;
        nr.value = nrm;
        return;
    }
    check(rlhip_gemm_f64(q.ctx(), (char)ta, (char)tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc), "gemm");
}
inline void gemm(Layout, Op ta, Op tb, int64_t m, int64_t n, int64_t k, float alpha, float const* A, int64_t lda,
                 float const* B, int64_t ldb, float beta, float* C, int64_t ldc, Queue& q = blas::default_queue()) {
    check(rlhip_gemm_f32(q.ctx(), (char)ta, (char)tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc), "gemm");
}
inline void syrk(Layout, Uplo u, Op t, int64_t n, int64_t k, double alpha, double const* A, int64_t lda, double beta,
                 double* C, int64_t ldc, Queue& q = blas::default_queue()) {
    check(rlhip_syrk_f64(q.ctx(), (char)u, (char)t, n, k, alpha, A, lda, beta, C, ldc), "syrk");
}
inline void syrk(Layout, Uplo u, Op t, int64_t n, int64_t k, float alpha, float const* A, int64_t lda, float beta,
                 float* C, int64_t ldc, Queue& q = blas::default_queue()) {
    check(rlhip_syrk_f32(q.ctx(), (char)u, (char)t, n, k, alpha, A, lda, beta, C, ldc), "syrk");
}
inline void trsm(Layout, Side s, Uplo u, Op t, Diag d, int64_t m, int64_t n, double alpha, double const* A, int64_t lda,
                 double* B, int64_t ldb, Queue& q = blas::default_queue()) {
    check(rlhip_trsm_f64(q.ctx(), (char)s, (char)u, (char)t, (char)d, m, n, alpha, A, lda, B, ldb), "trsm");
}
inline void trsm(Layout, Side s, Uplo u, Op t, Diag d, int64_t m, int64_t n, float alpha, float const* A, int64_t lda,
                 float* B, int64_t ldb, Queue& q = blas::default_queue()) {
    check(rlhip_trsm_f32(q.ctx(), (char)s, (char)u, (char)t, (char)d, m, n, alpha, A, lda, B, ldb), "trsm");
}
// (extension) B = alpha * (Bsrc * P) * inv(A): the out-of-place right-upper solve with the column pivoting of CQRRPT folded in
// (jpvt: 1-based, device; nullptr = no permutation); see rlhip_trsm_gather_f64
inline void trsm_gather(Diag d, int64_t m, int64_t n, double alpha, double const* A, int64_t lda, double const* Bsrc, int64_t ldsrc,
                        int64_t const* jpvt, double* B, int64_t ldb, Queue& q = blas::default_queue()) {
    check(rlhip_trsm_gather_f64(q.ctx(), (char)d, m, n, alpha, A, lda, Bsrc, ldsrc, jpvt, B, ldb), "trsm_gather");
}
inline void trsm_gather(Diag d, int64_t m, int64_t n, float alpha, float const* A, int64_t lda, float const* Bsrc, int64_t ldsrc,
                        int64_t const* jpvt, float* B, int64_t ldb, Queue& q = blas::default_queue()) {
    check(rlhip_trsm_gather_f32(q.ctx(), (char)d, m, n, alpha, A, lda, Bsrc, ldsrc, jpvt, B, ldb), "trsm_gather");
}
// (extension) columns [col0, col1) of that solve, B[:, 0 : col0) holding the leading columns already; jpvt maps into nsrc source columns.
// false: outside the fused kernel's domain, nothing was written -- the caller runs trsm_gather on the whole matrix.  See rlhip_trsm_gather_range_f64.
inline bool trsm_gather_range(Diag d, int64_t m, int64_t nsrc, double alpha, double const* A, int64_t lda, double const* Bsrc, int64_t ldsrc,
                              int64_t const* jpvt, double* B, int64_t ldb, int64_t col0, int64_t col1, Queue& q = blas::default_queue()) {
    const int rc = rlhip_trsm_gather_range_f64(q.ctx(), (char)d, m, nsrc, alpha, A, lda, Bsrc, ldsrc, jpvt, B, ldb, col0, col1);
    if (rc == 1) return false;
    check(rc, "trsm_gather_range");
    return true;
}
inline bool trsm_gather_range(Diag d, int64_t m, int64_t nsrc, float alpha, float const* A, int64_t lda, float const* Bsrc, int64_t ldsrc,
                              int64_t const* jpvt, float* B, int64_t ldb, int64_t col0, int64_t col1, Queue& q = blas::default_queue()) {
    const int rc = rlhip_trsm_gather_range_f32(q.ctx(), (char)d, m, nsrc, alpha, A, lda, Bsrc, ldsrc, jpvt, B, ldb, col0, col1);
    if (rc == 1) return false;
    check(rc, "trsm_gather_range");
    return true;
}
inline void trmm(Layout, Side s, Uplo u, Op t, Diag d, int64_t m, int64_t n, double alpha, double const* A, int64_t lda,
                 double* B, int64_t ldb, Queue& q = blas::default_queue()) {
    check(rlhip_trmm_f64(q.ctx(), (char)s, (char)u, (char)t, (char)d, m, n, alpha, A, lda, B, ldb), "trmm");
}
inline void trmm(Layout, Side s, Uplo u, Op t, Diag d, int64_t m, int64_t n, float alpha, float const* A, int64_t lda,
                 float* B, int64_t ldb, Queue& q = blas::default_queue()) {
    check(rlhip_trmm_f32(q.ctx(), (char)s, (char)u, (char)t, (char)d, m, n, alpha, A, lda, B, ldb), "trmm");
}

}  // namespace blas

namespace RandLAPACK {
using blas::Layout;
using blas::Op;
using blas::Side;
using blas::Diag;
using blas::Uplo;
}  // namespace RandLAPACK
