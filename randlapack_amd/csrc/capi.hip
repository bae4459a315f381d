// extern "C" surface of librlhip.so (declared in include/rlhip.h) + context / scratch-arena management.
#include "rlhip_internal.h"
#include "../../include/rlhip.h"
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>

namespace rlhip {
struct SasoOp;
int saso_build(rlhip_ctx* c, int64_t d, int64_t m, int nnz, int mode, const uint32_t ctr[4], const uint32_t key[2],
               uint32_t next_ctr[4], SasoOp** out);
int saso_destroy(rlhip_ctx* c, SasoOp* op);
template <typename T> int saso_dense(rlhip_ctx* c, const SasoOp* op, T* S);
template <typename T> int saso_apply_rows(rlhip_ctx* c, const SasoOp* op, int64_t n, T alpha, const T* A, int64_t lda, int64_t row0, int64_t mloc,
                                          T beta, T* B, int64_t ldb);
template <typename T> int saso_apply(rlhip_ctx* c, const SasoOp* op, int64_t n, T alpha, const T* A, int64_t lda, T beta,
                                     T* B, int64_t ldb);
template <typename T> int saso_apply_csr(rlhip_ctx* c, const SasoOp* op, int64_t n, T alpha, const int64_t* rowptrT, const int64_t* colidxT,
                                         const T* valsT, T beta, T* B, int64_t ldb, int64_t row0);
template <typename T> int col_swap(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, T* A, int64_t lda, const int64_t* idx);
template <typename T> int cholqrq(rlhip_ctx* c, int64_t m, int64_t k, T* A, int64_t lda, T* R, int reduce_gram, int* info_host);
int col_swap_i64(rlhip_ctx* c, int64_t n, int64_t k, int64_t* A, const int64_t* idx_dev);
template <typename T> int geqp3(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, int64_t* jpvt_dev, T* tau_dev);
template <typename T> int orhr_col(rlhip_ctx*, int64_t, int64_t, int64_t, T*, int64_t, T*, int64_t, T*);
template <typename T> int geqrf_q(rlhip_ctx*, int64_t, int64_t, T*, int64_t, T*, int64_t, int*);
template <typename T> int gemqrt_lt(rlhip_ctx*, int64_t, int64_t, int64_t, int64_t, const T*, int64_t, const T*, int64_t, T*, int64_t);
template <typename T> int gemqrt_lt_head(rlhip_ctx*, int64_t, int64_t, int64_t, const T*, int64_t, const T*, int64_t, T*, int64_t, T*);
template <typename T> int gemqrt_lt_tail(rlhip_ctx*, int64_t, int64_t, int64_t, const T*, int64_t, const T*, T*, int64_t);
template <typename T> int larft_gram(rlhip_ctx*, int64_t, int64_t, const T*, int64_t, const T*, T*, int64_t);
template <typename T> int row_sign(rlhip_ctx*, int64_t, T*, int64_t, const T*);
template <typename T> int tau_from_t(rlhip_ctx*, int64_t, int64_t, const T*, int64_t, T*);
template <typename T> int any_abs_gt(rlhip_ctx*, int64_t, const T*, T, int*);
template <typename T> int getrf(rlhip_ctx*, int64_t, int64_t, T*, int64_t, int64_t*, int*, int pivots_only);
int luqrcp_piv(rlhip_ctx*, int64_t, int64_t, const int64_t*, int64_t*);
template <typename T> int geqrf(rlhip_ctx*, int64_t, int64_t, T*, int64_t, T*);
template <typename T> int vrows_explicit(rlhip_ctx*, int64_t, int64_t, int64_t, const T*, int64_t, T*, int64_t);
template <typename T> int qrp_partial(rlhip_ctx*, int64_t, int64_t, int64_t, T*, int64_t, int64_t*, T*);
template <typename T> int geqp3_steps(rlhip_ctx*, int64_t, int64_t, int64_t, T*, int64_t, int64_t*, T*);
template <typename T> int gemqrt_rn(rlhip_ctx*, int64_t, int64_t, int64_t, const T*, int64_t, const T*, int64_t, T*, int64_t);
template <typename T> int ungqr(rlhip_ctx*, int64_t, int64_t, T*, int64_t, const T*);
template <typename T> int laswp(rlhip_ctx*, int64_t, T*, int64_t, int64_t, int64_t, const int64_t*);
template <typename T> int fill_dense_rows(rlhip_ctx*, int, int64_t, int64_t, int64_t, int64_t, T*, int64_t, const uint32_t*, const uint32_t*, uint32_t*);
int philox_raw(rlhip_ctx* c, int64_t nblk, uint32_t* out_dev, const uint32_t ctr[4], const uint32_t key[2]);
}

// ------------------------------------------------------------------ scratch arena
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static size_t seg_vstart(const rlhip_ctx* c, int k) {
    size_t v = 0;
    for (int i = 0; i < k; ++i) v += c->segs[i].size;
    return v;
}

// RLHIP_POISON=1 (diagnostic): every block handed out by the scratch arena and by the output pool is filled with 0xFF bytes (NaN as a float
// or double, -1 as an integer) on the context's stream first.  A fresh arena is zero-filled by the driver, so a kernel that reads scratch it
// never wrote passes every test on a new context and fails on a used one (round 6: ABRIK's sharded default panels); the GPU suite is run
// once per round with this set.
static int rlhip_poison() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("RLHIP_POISON"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

void* rlhip_ws_alloc(rlhip_ctx* c, size_t bytes) {
    bytes = align_up(bytes ? bytes : 1, 256);
    for (;;) {
        if (c->cur_seg < c->nsegs && c->cur_used + bytes <= c->segs[c->cur_seg].size) {
            void* p = c->segs[c->cur_seg].base + c->cur_used;
            c->cur_used += bytes;
            size_t v = seg_vstart(c, c->cur_seg) + c->cur_used;
            if (v > c->ws_highwater) c->ws_highwater = v;
            if (rlhip_poison()) (void)hipMemsetAsync(p, 0xFF, bytes, c->stream);
            return p;
        }
        if (c->cur_seg + 1 < c->nsegs) { ++c->cur_seg; c->cur_used = 0; continue; }   // reuse a later segment
        if (c->nsegs >= 32) return nullptr;
        size_t total = seg_vstart(c, c->nsegs);
        size_t want = bytes > total ? bytes : total;          // at least double the arena
        if (want < ((size_t)256 << 20)) want = (size_t)256 << 20;
        want = align_up(want, 1 << 20);
        char* p = nullptr;
        if (hipMalloc((void**)&p, want) != hipSuccess) {
            (void)hipGetLastError();                           // a failed allocation must not poison the next launch check
            want = align_up(bytes, 1 << 20);                   // memory is tight: take exactly what is needed
            if (hipMalloc((void**)&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        c->segs[c->nsegs].base = p;
        c->segs[c->nsegs].size = want;
        c->cur_seg = c->nsegs++;
        c->cur_used = 0;
    }
}

void* rlhip_xchg_buffer(rlhip_ctx* c, size_t bytes) {
    if (bytes <= c->xchg_bytes) return c->xchg;
    constexpr int kind = 2;   // uncached: polled words never sit in an L2 (measured against ordinary / fine-grained memory: geqp3 1280 x 1024 8.37 -> 8.15 ms)
    if (c->xchg) { rlhip_stream_sync(c); hipFree(c->xchg); c->xchg = nullptr; c->xchg_bytes = 0; }
    bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    void* p = nullptr;
    hipError_t e;
    if (kind == 1) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    else if (kind == 2) e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    else e = hipMalloc(&p, bytes);
    if (e != hipSuccess && kind != 0) { (void)hipGetLastError(); e = hipMalloc(&p, bytes); }
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    c->xchg = p; c->xchg_bytes = bytes;
    return p;
}

void* rlhip_xloc_buffer(rlhip_ctx* c, size_t bytes) {
    if (bytes <= c->xloc_bytes) return c->xloc;
    if (c->xloc) { rlhip_stream_sync(c); hipFree(c->xloc); c->xloc = nullptr; c->xloc_bytes = 0; }
    bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    c->xloc = p; c->xloc_bytes = bytes;
    return p;
}

size_t rlhip_ws_mark(rlhip_ctx* c) { return c->nsegs ? seg_vstart(c, c->cur_seg) + c->cur_used : 0; }

void rlhip_ws_release(rlhip_ctx* c, size_t mark) {
    // find the segment holding `mark`
    size_t v = 0;
    int k = 0;
    for (; k < c->nsegs; ++k) {
        if (mark <= v + c->segs[k].size) break;
        v += c->segs[k].size;
    }
    if (k >= c->nsegs) k = c->nsegs ? c->nsegs - 1 : 0;
    c->cur_seg = k;
    c->cur_used = (c->nsegs && mark >= v) ? (mark - v) : 0;
    if (mark == 0 && c->nsegs > 1) {
        // stack empty: merge the segments into one arena of the high-water size
        rlhip_stream_sync(c);
        for (int i = 0; i < c->nsegs; ++i) hipFree(c->segs[i].base);
        c->nsegs = 0; c->cur_seg = 0; c->cur_used = 0;
        size_t want = align_up(c->ws_highwater + (c->ws_highwater >> 3), 1 << 20);
        char* p = nullptr;
        if (hipMalloc((void**)&p, want) == hipSuccess) { c->segs[0].base = p; c->segs[0].size = want; c->nsegs = 1; }
    }
}

namespace rlhip {
template <typename T>
int gemm_impl(rlhip_ctx* c, int transA, int transB, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
              const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri, double* ssqA_dev, int* ssq_done);
}
__global__ void rlhip_zero_f64_kernel(double* p) { *p = 0.0; }

extern "C" {

const char* rlhip_version(void) { return "rlhip 0.1 (gfx950)"; }

int rlhip_create(rlhip_ctx** out, int device, void* hip_stream, int own_stream) {
    if (!out) return -1;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[rlhip] no HIP device visible (%s)\n", hipGetErrorString(e));
        return RLHIP_ERR_HIP(e == hipSuccess ? hipErrorNoDevice : e);
    }
    if (device < 0 || device >= ndev) return -2;
    RLHIP_CHECK(hipSetDevice(device));
    rlhip_ctx* c = new rlhip_ctx();
    c->device = device;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->num_cu = ncu;
    }
    if (!own_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->owns_stream = false;
    } else {
        RLHIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->owns_stream = true;
    }
    RLHIP_CHECK(hipHostMalloc((void**)&c->h_mail, 64 * sizeof(int64_t), hipHostMallocDefault));
    RLHIP_CHECK(hipMalloc((void**)&c->d_mail, 64 * sizeof(int64_t)));
    RLHIP_CHECK(hipEventCreate(&c->ev0));
    RLHIP_CHECK(hipEventCreate(&c->ev1));
    RLHIP_CHECK(hipEventCreateWithFlags(&c->ev_flag, hipEventDisableTiming));
    {
        char* p = nullptr;
        RLHIP_CHECK(hipMalloc((void**)&p, (size_t)64 << 20));
        c->segs[0].base = p; c->segs[0].size = (size_t)64 << 20; c->nsegs = 1;
    }
    {
        size_t fr = 0, tot = 0;
        c->pool_cap_bytes = (hipMemGetInfo(&fr, &tot) == hipSuccess) ? tot / 8 : ((size_t)8 << 30);   // <= 1/8 of HBM idles here
    }
    *out = c;
    return 0;
}

// A second context on the parent's device with its own HIGH-PRIORITY stream and its own scratch arena: work that should run BESIDE the
// parent's stream and get CUs as soon as they free up (BQRRP's look-ahead: the latency-bound pivoting of the next panel beside the
// trailing update).  Ordering between the two streams is explicit: rlhip_order_after.
int rlhip_create_side(rlhip_ctx* parent, rlhip_ctx** out) {
    if (!parent || !out) return -1;
    RLHIP_CHECK(hipSetDevice(parent->device));
    int lo = 0, hi = 0;
    RLHIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));             // (numerically lower = higher priority)
    hipStream_t st = nullptr;
    RLHIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
    const int rc = rlhip_create(out, parent->device, (void*)st, 0);
    if (rc) { hipStreamDestroy(st); return rc; }
    (*out)->owns_stream = true;
    for (int i = 0; i < RLHIP_OPT_COUNT; ++i) (*out)->opt[i] = parent->opt[i];
    (*out)->avoid_persistent = 1;        // a side context shares the device by definition: no kernel of its own may wait for every CU to be free
    return 0;
}

// everything enqueued on `waiter`'s stream from now on starts after everything enqueued on `signaler`'s stream so far (no host wait)
// The parent's OWN side context: created on the first call, handed out again on every later one, destroyed with the parent.  For callers
// inside a timed path (CQRRPT's split QRCP: a stream, two mailboxes and a scratch arena per call would cost more than the overlap wins).
int rlhip_side_of(rlhip_ctx* parent, rlhip_ctx** out) {
    if (!parent || !out) return -1;
    if (!parent->side_ctx) {
        const int rc = rlhip_create_side(parent, &parent->side_ctx);
        if (rc) { parent->side_ctx = nullptr; return rc; }
    }
    *out = parent->side_ctx;
    return 0;
}

// columns per workgroup of the tag-exchange pivoted QR on this context (0: the default, 4): a factorization meant to run BESIDE another
// stream's kernel is packed into fewer workgroups, i.e. fewer CUs
int rlhip_set_qrcp_cols(rlhip_ctx* c, int cols) {
    if (!c || cols < 0) return -1;
    c->qrcp_cols_per_wg = cols;
    return 0;
}

int rlhip_set_option(rlhip_ctx* c, int option, int64_t value) {
    if (!c || option < 0 || option >= RLHIP_OPT_COUNT) return -1;
    c->opt[option] = value;
    if (c->side_ctx) c->side_ctx->opt[option] = value;
    return 0;
}
int64_t rlhip_get_option(rlhip_ctx* c, int option) {
    if (!c || option < 0 || option >= RLHIP_OPT_COUNT) return INT64_MIN;
    return c->opt[option];
}

int rlhip_order_after(rlhip_ctx* waiter, rlhip_ctx* signaler) {
    if (!waiter || !signaler) return -1;
    if (waiter->stream == signaler->stream) return 0;
    hipEvent_t ev = nullptr;
    RLHIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, signaler->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter->stream, ev, 0);
    hipEventDestroy(ev);                                                 // (released once the wait has been consumed)
    return e == hipSuccess ? 0 : RLHIP_ERR_HIP(e);
}

int rlhip_destroy(rlhip_ctx* c) {
    if (!c) return 0;
    rlhip_comm_destroy(c);
    hipSetDevice(c->device);
    rlhip_stream_sync(c);
    for (int i = 0; i < c->npool; ++i) hipFree(c->pool[i].p);
    for (int i = 0; i < c->nsegs; ++i) hipFree(c->segs[i].base);
    if (c->d_mail) hipFree(c->d_mail);
    if (c->xchg) hipFree(c->xchg);
    if (c->xloc) hipFree(c->xloc);
    if (c->h_mail) hipHostFree(c->h_mail);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev_flag) hipEventDestroy(c->ev_flag);
    if (c->side_ctx) { rlhip_destroy(c->side_ctx); c->side_ctx = nullptr; hipSetDevice(c->device); }
    if (c->side) hipStreamDestroy(c->side);
    if (c->owns_stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int rlhip_sync(rlhip_ctx* c) { RLHIP_CHECK(rlhip_stream_sync(c)); return 0; }
void* rlhip_stream(rlhip_ctx* c) { return (void*)c->stream; }

static void pool_drop(rlhip_ctx* c, int i) {
    hipFree(c->pool[i].p);
    c->pool_idle_bytes -= c->pool[i].bytes;
    c->pool[i] = c->pool[--c->npool];
}
int rlhip_malloc_host(rlhip_ctx* c, void** p, size_t bytes) {
    RLHIP_CHECK(hipSetDevice(c->device));
    RLHIP_CHECK(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocPortable));
    return 0;
}
int rlhip_free_host(rlhip_ctx* c, void* p) {
    if (!p) return 0;
    RLHIP_CHECK(rlhip_stream_sync(c));
    RLHIP_CHECK(hipHostFree(p));
    return 0;
}
int rlhip_trim(rlhip_ctx* c) {
    RLHIP_CHECK(rlhip_stream_sync(c));
    for (int i = c->npool - 1; i >= 0; --i)
        if (!c->pool[i].in_use) pool_drop(c, i);
    return 0;
}
int rlhip_malloc(rlhip_ctx* c, void** p, size_t bytes) {
    RLHIP_CHECK(hipSetDevice(c->device));
    bytes = bytes ? ((bytes + 255) & ~(size_t)255) : 256;
    for (int i = 0; i < c->npool; ++i)
        if (!c->pool[i].in_use && c->pool[i].bytes == bytes) {
            c->pool[i].in_use = true;
            c->pool_idle_bytes -= bytes;
            *p = c->pool[i].p;
            if (rlhip_poison()) (void)hipMemsetAsync(*p, 0xFF, bytes, c->stream);
            return 0;
        }
    static int trace = -1;
    if (trace < 0) { const char* t = getenv("RLHIP_POOL_TRACE"); trace = t ? atoi(t) : 0; }
    if (trace) fprintf(stderr, "[rlhip pool] miss: %zu bytes (%d blocks, %zu idle bytes)\n", bytes, c->npool, c->pool_idle_bytes);
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {                       // give the idle blocks back and try once more
        (void)hipGetLastError();
        rlhip_trim(c);
        e = hipMalloc(p, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return RLHIP_ERR_HIP(e); }
    }
    if (c->npool >= 256) {
        // table full: the least recently freed idle block makes room, so that every block stays tracked and a request that repeats (a
        // driver's sketching operator, call after call) is served from the table -- an untracked block would be hipMalloc'ed and hipFree'd
        // by every call (a device synchronisation each time, and memory unmapped and remapped between launches)
        int lru = -1;
        for (int j = 0; j < c->npool; ++j)
            if (!c->pool[j].in_use && (lru < 0 || c->pool[j].stamp < c->pool[lru].stamp)) lru = j;
        if (lru >= 0) { (void)rlhip_stream_sync(c); pool_drop(c, lru); }
    }
    if (c->npool < 256) c->pool[c->npool++] = {*p, bytes, true, 0};
    if (rlhip_poison()) (void)hipMemsetAsync(*p, 0xFF, bytes, c->stream);
    return 0;
}
int rlhip_free(rlhip_ctx* c, void* p) {
    if (!p) return 0;
    for (int i = 0; i < c->npool; ++i)
        if (c->pool[i].p == p) {
            c->pool[i].in_use = false;
            c->pool[i].stamp = ++c->pool_clock;
            c->pool_idle_bytes += c->pool[i].bytes;
            if (c->pool_idle_bytes > c->pool_cap_bytes) {          // over the cap: release least recently freed blocks
                RLHIP_CHECK(rlhip_stream_sync(c));
                while (c->pool_idle_bytes > c->pool_cap_bytes) {
                    int lru = -1;
                    for (int j = 0; j < c->npool; ++j)
                        if (!c->pool[j].in_use && (lru < 0 || c->pool[j].stamp < c->pool[lru].stamp)) lru = j;
                    if (lru < 0) break;
                    pool_drop(c, lru);
                }
            }
            return 0;
        }
    RLHIP_CHECK(rlhip_stream_sync(c));                 // not one of ours (table was full): plain free
    RLHIP_CHECK(hipFree(p));
    return 0;
}
int rlhip_memcpy_h2d(rlhip_ctx* c, void* dst, const void* src, size_t bytes) {
    RLHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    RLHIP_CHECK(rlhip_stream_sync(c));
    return 0;
}
int rlhip_memcpy_d2h(rlhip_ctx* c, void* dst, const void* src, size_t bytes) {
    RLHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(rlhip_stream_sync(c));
    return 0;
}
int rlhip_memcpy_d2d(rlhip_ctx* c, void* dst, const void* src, size_t bytes) {
    RLHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int rlhip_memset(rlhip_ctx* c, void* dst, int byte, size_t bytes) {
    RLHIP_CHECK(hipMemsetAsync(dst, byte, bytes, c->stream));
    return 0;
}
int rlhip_reserve_workspace(rlhip_ctx* c, size_t bytes) {
    if (rlhip_ws_mark(c) != 0) return -2;  // only legal between top-level calls
    size_t total = 0;
    for (int i = 0; i < c->nsegs; ++i) total += c->segs[i].size;
    if (c->nsegs == 1 && bytes <= total) return 0;
    RLHIP_CHECK(rlhip_stream_sync(c));
    for (int i = 0; i < c->nsegs; ++i) hipFree(c->segs[i].base);
    c->nsegs = 0; c->cur_seg = 0; c->cur_used = 0;
    bytes = align_up(bytes > total ? bytes : total, 1 << 20);
    char* p = nullptr;
    RLHIP_CHECK(hipMalloc((void**)&p, bytes));
    c->segs[0].base = p; c->segs[0].size = bytes; c->nsegs = 1;
    return 0;
}
size_t rlhip_workspace_highwater(rlhip_ctx* c) { return c->ws_highwater; }

/* scratch arena (stream-ordered bump allocator) for driver temporaries */
size_t rlhip_scratch_mark(rlhip_ctx* c) { return rlhip_ws_mark(c); }
int rlhip_scratch_alloc(rlhip_ctx* c, void** p, size_t bytes) {
    *p = rlhip_ws_alloc(c, bytes);
    return *p ? 0 : RLHIP_ERR_HIP(hipErrorOutOfMemory);
}
int rlhip_scratch_release(rlhip_ctx* c, size_t mark) { rlhip_ws_release(c, mark); return 0; }

int rlhip_timer_start(rlhip_ctx* c) { RLHIP_CHECK(hipEventRecord(c->ev0, c->stream)); return 0; }
int rlhip_timer_stop_ms(rlhip_ctx* c, float* ms) {
    RLHIP_CHECK(hipEventRecord(c->ev1, c->stream));
    RLHIP_CHECK(hipEventSynchronize(c->ev1));
    RLHIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

// ------------------------------------------------------------------ RNG
int rlhip_philox4x32_10(rlhip_ctx* c, int64_t nblocks, uint32_t* out_dev, const uint32_t ctr[4],
                        const uint32_t key[2]) {
    return rlhip::philox_raw(c, nblocks, out_dev, ctr, key);
}
int rlhip_fill_dense_f64(rlhip_ctx* c, int dist, int64_t rows, int64_t cols, double* buf, const uint32_t ctr[4],
                         const uint32_t key[2], uint32_t next_ctr[4]) {
    return rlhip::fill_dense<double>(c, dist, rows, cols, buf, ctr, key, next_ctr);
}
int rlhip_fill_dense_rows_f64(rlhip_ctx* c, int dist, int64_t glob_rows, int64_t cols, int64_t row0, int64_t loc_rows, double* buf,
                              int64_t ld, const uint32_t ctr[4], const uint32_t key[2], uint32_t next_ctr[4]) {
    return rlhip::fill_dense_rows<double>(c, dist, glob_rows, cols, row0, loc_rows, buf, ld, ctr, key, next_ctr);
}
int rlhip_fill_dense_rows_f32(rlhip_ctx* c, int dist, int64_t glob_rows, int64_t cols, int64_t row0, int64_t loc_rows, float* buf,
                              int64_t ld, const uint32_t ctr[4], const uint32_t key[2], uint32_t next_ctr[4]) {
    return rlhip::fill_dense_rows<float>(c, dist, glob_rows, cols, row0, loc_rows, buf, ld, ctr, key, next_ctr);
}
int rlhip_fill_dense_f32(rlhip_ctx* c, int dist, int64_t rows, int64_t cols, float* buf, const uint32_t ctr[4],
                         const uint32_t key[2], uint32_t next_ctr[4]) {
    return rlhip::fill_dense<float>(c, dist, rows, cols, buf, ctr, key, next_ctr);
}

int rlhip_saso_create(rlhip_ctx* c, int64_t d, int64_t m, int nnz, const uint32_t ctr[4], const uint32_t key[2],
                      uint32_t next_ctr[4], rlhip_saso** out) {
    return rlhip::saso_build(c, d, m, nnz, -1, ctr, key, next_ctr, (rlhip::SasoOp**)out);
}
int rlhip_saso_create_mode(rlhip_ctx* c, int64_t d, int64_t m, int nnz, int mode, const uint32_t ctr[4], const uint32_t key[2],
                           uint32_t next_ctr[4], rlhip_saso** out) {
    return rlhip::saso_build(c, d, m, nnz, mode, ctr, key, next_ctr, (rlhip::SasoOp**)out);
}
int rlhip_saso_destroy(rlhip_ctx* c, rlhip_saso* S) { return rlhip::saso_destroy(c, (rlhip::SasoOp*)S); }
int rlhip_col_swap_i64(rlhip_ctx* c, int64_t n, int64_t k, int64_t* A, const int64_t* idx) {
    return rlhip::col_swap_i64(c, n, k, A, idx);
}

int rlhip_luqrcp_piv(rlhip_ctx* c, int64_t sd, int64_t cols, const int64_t* ipiv, int64_t* J) {
    return rlhip::luqrcp_piv(c, sd, cols, ipiv, J);
}

// ------------------------------------------------------------------ BLAS-3
static inline int op_flag(char t, int* out) {
    if (t == 'N' || t == 'n') { *out = 0; return 0; }
    if (t == 'T' || t == 't' || t == 'C' || t == 'c') { *out = 1; return 0; }
    return 1;
}

#define RLHIP_DEFINE_BLAS3(SUF, T)                                                                              \
    int rlhip_gemm_##SUF(rlhip_ctx* c, char ta, char tb, int64_t m, int64_t n, int64_t k, T alpha, const T* A,  \
                         int64_t lda, const T* B, int64_t ldb, T beta, T* C, int64_t ldc) {                     \
        int fa, fb;                                                                                             \
        if (op_flag(ta, &fa)) return -2;                                                                        \
        if (op_flag(tb, &fb)) return -3;                                                                        \
        return rlhip::gemm<T>(c, fa, fb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);                          \
    }                                                                                                           \
    int rlhip_syrk_##SUF(rlhip_ctx* c, char uplo, char trans, int64_t n, int64_t k, T alpha, const T* A,        \
                         int64_t lda, T beta, T* C, int64_t ldc) {                                              \
        int ft;                                                                                                 \
        if (uplo != 'U' && uplo != 'u') return -2;                                                              \
        if (op_flag(trans, &ft)) return -3;                                                                     \
        return rlhip::syrk<T>(c, rlhip::Upper, ft, n, k, alpha, A, lda, beta, C, ldc);                           \
    }                                                                                                           \
    int rlhip_trsm_##SUF(rlhip_ctx* c, char side, char uplo, char trans, char diag, int64_t m, int64_t n,       \
                         T alpha, const T* A, int64_t lda, T* B, int64_t ldb) {                                 \
        if (side != 'R' && side != 'r') return -2;                                                              \
        if (uplo != 'U' && uplo != 'u') return -3;                                                              \
        if (trans != 'N' && trans != 'n') return -4;                                                            \
        int fd = (diag == 'U' || diag == 'u') ? 1 : 0;                                                          \
        return rlhip::trsm_right_upper<T>(c, fd, m, n, alpha, A, lda, B, ldb);                                   \
    }                                                                                                           \
    int rlhip_trsm_gather_##SUF(rlhip_ctx* c, char diag, int64_t m, int64_t n, T alpha, const T* A, int64_t lda, \
                                const T* Bsrc, int64_t ldsrc, const int64_t* jpvt_dev, T* B, int64_t ldb) {     \
        int fd = (diag == 'U' || diag == 'u') ? 1 : 0;                                                          \
        return rlhip::trsm_right_upper_oop<T>(c, fd, m, n, alpha, A, lda, Bsrc, ldsrc, jpvt_dev, B, ldb);        \
    }                                                                                                           \
    int rlhip_trsm_gather_range_##SUF(rlhip_ctx* c, char diag, int64_t m, int64_t nsrc, T alpha, const T* A, int64_t lda, \
                                      const T* Bsrc, int64_t ldsrc, const int64_t* jpvt_dev, T* B, int64_t ldb, int64_t col0, int64_t col1) { \
        int fd = (diag == 'U' || diag == 'u') ? 1 : 0;                                                          \
        return rlhip::trsm_right_upper_oop_range<T>(c, fd, m, nsrc, alpha, A, lda, Bsrc, ldsrc, jpvt_dev, B, ldb, col0, col1); \
    }                                                                                                           \
    int rlhip_trmm_##SUF(rlhip_ctx* c, char side, char uplo, char trans, char diag, int64_t m, int64_t n,       \
                         T alpha, const T* A, int64_t lda, T* B, int64_t ldb) {                                 \
        if (uplo != 'U' && uplo != 'u') return -3;                                                              \
        int fd = (diag == 'U' || diag == 'u') ? 1 : 0;                                                          \
        if (side == 'L' || side == 'l')                                                                         \
            return rlhip::trmm_left_upper<T>(c, (trans != 'N' && trans != 'n') ? 1 : 0, fd, m, n, alpha, A, lda, B, ldb); \
        if (side != 'R' && side != 'r') return -2;                                                              \
        if (trans != 'N' && trans != 'n') return -4;                                                            \
        return rlhip::trmm_right_upper<T>(c, fd, m, n, alpha, A, lda, B, ldb);                                   \
    }                                                                                                           \
    int rlhip_potrf_##SUF(rlhip_ctx* c, char uplo, int64_t n, T* A, int64_t lda) {                              \
        if (uplo != 'U' && uplo != 'u') return -2;                                                              \
        int info = 0;                                                                                           \
        int rc = rlhip::potrf_upper<T>(c, n, A, lda, &info);                                                     \
        return rc ? rc : info;                                                                                  \
    }                                                                                                           \
    int rlhip_cholqrq_##SUF(rlhip_ctx* c, int64_t m, int64_t k, T* A, int64_t lda, T* R, int reduce_gram, int* info_host) { \
        if (!info_host) return -8;                                                                              \
        if (m < 0) return -2;                                                                                   \
        if (k < 0) return -3;                                                                                   \
        return rlhip::cholqrq<T>(c, m, k, A, lda, R, reduce_gram, info_host);                                    \
    }                                                                                                           \
    int rlhip_lange_fro_##SUF(rlhip_ctx* c, int64_t m, int64_t n, const T* A, int64_t lda, T* res) {            \
        return rlhip::lange_fro<T>(c, m, n, A, lda, res);                                                        \
    }                                                                                                           \
    int rlhip_lacpy_##SUF(rlhip_ctx* c, char uplo, int64_t m, int64_t n, const T* A, int64_t lda, T* B,         \
                          int64_t ldb) {                                                                        \
        int u = (uplo == 'U' || uplo == 'u') ? 0 : (uplo == 'L' || uplo == 'l') ? 1 : 2;                        \
        return rlhip::lacpy<T>(c, u, m, n, A, lda, B, ldb);                                                      \
    }                                                                                                           \
    int rlhip_laset_##SUF(rlhip_ctx* c, char uplo, int64_t m, int64_t n, T offd, T diag, T* A, int64_t lda) {   \
        int u = (uplo == 'U' || uplo == 'u') ? 0 : (uplo == 'L' || uplo == 'l') ? 1 : 2;                        \
        return rlhip::laset<T>(c, u, m, n, offd, diag, A, lda);                                                  \
    }                                                                                                           \
    int rlhip_saso_apply_rows_##SUF(rlhip_ctx* c, const rlhip_saso* S, int64_t n, T alpha, const T* A, int64_t lda, int64_t row0, \
                                    int64_t mloc, T beta, T* B, int64_t ldb) {                                  \
        return rlhip::saso_apply_rows<T>(c, (const rlhip::SasoOp*)S, n, alpha, A, lda, row0, mloc, beta, B, ldb); \
    }                                                                                                           \
    int rlhip_saso_apply_##SUF(rlhip_ctx* c, const rlhip_saso* S, int64_t n, T alpha, const T* A, int64_t lda, T beta, \
                               T* B, int64_t ldb) {                                                             \
        return rlhip::saso_apply<T>(c, (const rlhip::SasoOp*)S, n, alpha, A, lda, beta, B, ldb);                 \
    }                                                                                                           \
    int rlhip_saso_apply_csr_##SUF(rlhip_ctx* c, const rlhip_saso* S, int64_t n, T alpha, const int64_t* rowptrT, const int64_t* colidxT, \
                                   const T* valsT, T beta, T* B, int64_t ldb, int64_t row0) {                   \
        return rlhip::saso_apply_csr<T>(c, (const rlhip::SasoOp*)S, n, alpha, rowptrT, colidxT, valsT, beta, B, ldb, row0); \
    }                                                                                                           \
    int rlhip_saso_dense_##SUF(rlhip_ctx* c, const rlhip_saso* S, T* dense) {                                   \
        return rlhip::saso_dense<T>(c, (const rlhip::SasoOp*)S, dense);                                          \
    }                                                                                                           \
    int rlhip_col_swap_##SUF(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, T* A, int64_t lda, const int64_t* idx) { \
        return rlhip::col_swap<T>(c, m, n, k, A, lda, idx);                                                      \
    }                                                                                                           \
    int rlhip_geqp3_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, int64_t* jpvt, T* tau) {       \
        return rlhip::geqp3<T>(c, m, n, A, lda, jpvt, tau);                                                      \
    }                                                                                                           \
    int rlhip_get_diag_##SUF(rlhip_ctx* c, int64_t n, const T* A, int64_t lda, T* diag_host) {                  \
        if (n <= 0) return 0;                                                                                   \
        RLHIP_CHECK(hipMemcpy2DAsync(diag_host, sizeof(T), A, (size_t)(lda + 1) * sizeof(T), sizeof(T), (size_t)n, \
                                     hipMemcpyDeviceToHost, c->stream));                                        \
        RLHIP_CHECK(rlhip_stream_sync(c));                                                           \
        return 0;                                                                                               \
    }                                                                                                           \
    int rlhip_orhr_col_##SUF(rlhip_ctx* c, int64_t m, int64_t n, int64_t nb, T* A, int64_t lda, T* Tm, int64_t ldt, T* D) { \
        return rlhip::orhr_col<T>(c, m, n, nb, A, lda, Tm, ldt, D);                                              \
    }                                                                                                           \
    int rlhip_gemqrt_##SUF(rlhip_ctx* c, char side, char trans, int64_t m, int64_t n, int64_t k, int64_t nb, const T* V, \
                           int64_t ldv, const T* Tm, int64_t ldt, T* C, int64_t ldc) {                          \
        const bool left = (side == 'L' || side == 'l'), tr = (trans == 'T' || trans == 't');                   \
        if (left && tr) return rlhip::gemqrt_lt<T>(c, m, n, k, nb, V, ldv, Tm, ldt, C, ldc);                    \
        if (!left && (side == 'R' || side == 'r') && (trans == 'N' || trans == 'n')) {                         \
            if (nb < k) return -7; /* one compact-WY block on this side */                                      \
            return rlhip::gemqrt_rn<T>(c, m, n, k, V, ldv, Tm, ldt, C, ldc);                                    \
        }                                                                                                       \
        return left ? -3 : -2;                                                                                  \
    }                                                                                                           \
    int rlhip_gemqrt_head_##SUF(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, const T* V, int64_t ldv, const T* Tm, int64_t ldt, T* C, int64_t ldc, T* W2) { \
        return rlhip::gemqrt_lt_head<T>(c, m, n, k, V, ldv, Tm, ldt, C, ldc, W2);                               \
    }                                                                                                           \
    int rlhip_gemqrt_tail_##SUF(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, const T* V, int64_t ldv, const T* W2, T* C, int64_t ldc) { \
        return rlhip::gemqrt_lt_tail<T>(c, m, n, k, V, ldv, W2, C, ldc);                                        \
    }                                                                                                           \
    int rlhip_geqrf_q_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* R, int64_t ldr) {         \
        int done = 0;                                                                                            \
        const int rc = rlhip::geqrf_q<T>(c, m, n, A, lda, R, ldr, &done);                                        \
        return rc ? rc : (done ? 0 : 1);                                                                        \
    }                                                                                                           \
    int rlhip_vrows_explicit_##SUF(rlhip_ctx* c, int64_t br, int64_t toff, int64_t tcnt, const T* Vtop, int64_t ldv, T* out, int64_t ldo) { \
        return rlhip::vrows_explicit<T>(c, br, toff, tcnt, Vtop, ldv, out, ldo);                                \
    }                                                                                                           \
    int rlhip_qrp_partial_##SUF(rlhip_ctx* c, int64_t m, int64_t n, int64_t steps, T* A, int64_t lda, int64_t* jpvt, T* tau) { \
        return rlhip::qrp_partial<T>(c, m, n, steps, A, lda, jpvt, tau);                                        \
    }                                                                                                           \
    int rlhip_geqp3_steps_##SUF(rlhip_ctx* c, int64_t m, int64_t n, int64_t steps, T* A, int64_t lda, int64_t* jpvt, T* tau) { \
        return rlhip::geqp3_steps<T>(c, m, n, steps, A, lda, jpvt, tau);                                        \
    }                                                                                                           \
    int rlhip_larft_##SUF(rlhip_ctx* c, int64_t m, int64_t k, const T* V, int64_t ldv, const T* tau, T* Tm, int64_t ldt) { \
        return rlhip::larft_gram<T>(c, m, k, V, ldv, tau, Tm, ldt);                                              \
    }                                                                                                           \
    int rlhip_row_sign_##SUF(rlhip_ctx* c, int64_t n, T* R, int64_t ldr, const T* D) {                          \
        return rlhip::row_sign<T>(c, n, R, ldr, D);                                                              \
    }                                                                                                           \
    int rlhip_tau_from_t_##SUF(rlhip_ctx* c, int64_t k, int64_t nb, const T* Tm, int64_t ldt, T* tau) {         \
        return rlhip::tau_from_t<T>(c, k, nb, Tm, ldt, tau);                                                     \
    }                                                                                                           \
    int rlhip_any_abs_gt_##SUF(rlhip_ctx* c, int64_t n, const T* x, T thr, int* any_host) {                     \
        return rlhip::any_abs_gt<T>(c, n, x, thr, any_host);                                                     \
    }                                                                                                           \
    int rlhip_getrf_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv) {               \
        int info = 0;                                                                                           \
        int rc = rlhip::getrf<T>(c, m, n, A, lda, ipiv, &info, 0);                                               \
        return rc ? rc : info;                                                                                  \
    }                                                                                                           \
    int rlhip_getrf_piv_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, int64_t* ipiv) {           \
        int info = 0;                                                                                           \
        int rc = rlhip::getrf<T>(c, m, n, A, lda, ipiv, &info, 1);                                               \
        return rc ? rc : info;                                                                                  \
    }                                                                                                           \
    int rlhip_geqrf_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* tau) {                       \
        return rlhip::geqrf<T>(c, m, n, A, lda, tau);                                                           \
    }                                                                                                           \
    int rlhip_ungqr_##SUF(rlhip_ctx* c, int64_t m, int64_t n, int64_t k, T* A, int64_t lda, const T* tau) {      \
        if (k != n) return -4;                                                                                  \
        return rlhip::ungqr<T>(c, m, n, A, lda, tau);                                                           \
    }                                                                                                           \
    int rlhip_laswp_##SUF(rlhip_ctx* c, int64_t n, T* A, int64_t lda, int64_t k1, int64_t k2, const int64_t* ipiv) { \
        return rlhip::laswp<T>(c, n, A, lda, k1, k2, ipiv);                                                     \
    }                                                                                                           \
    int rlhip_add_diag_##SUF(rlhip_ctx* c, int64_t n, T alpha, T* A, int64_t lda) {                             \
        return rlhip::add_diag<T>(c, n, alpha, A, lda);                                                          \
    }                                                                                                           \
    int rlhip_gesdd_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* S, T* U, int64_t ldu, T* VT,  \
                          int64_t ldvt, int* sweeps) {                                                          \
        return rlhip::gesdd_tall<T>(c, m, n, A, lda, S, U, ldu, VT, ldvt, sweeps);                               \
    }                                                                                                           \
    int rlhip_transpose_##SUF(rlhip_ctx* c, int64_t m, int64_t n, const T* A, int64_t lda, T* AT, int64_t ldat,  \
                              int upper_only) {                                                                 \
        return rlhip::transpose<T>(c, m, n, A, lda, AT, ldat, upper_only);                                       \
    }                                                                                                           \
    int rlhip_gesvdj_##SUF(rlhip_ctx* c, int64_t m, int64_t n, T* A, int64_t lda, T* S, T* VT, int64_t ldvt,    \
                           int* sweeps) {                                                                       \
        return rlhip::gesvdj<T>(c, m, n, A, lda, S, VT, ldvt, sweeps);                                           \
    }



/* C = alpha*op(A)*op(B) + beta*C  AND  ||A||_F in one pass over A when the stream-K kernel takes the problem
 * (the A tiles are in LDS anyway); otherwise gemm followed by lange.  A is (m x k) for 'N', (k x m) for 'T'. */
int rlhip_gemm_norma_f64(rlhip_ctx* c, char ta, char tb, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
                         int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                         double* norm_a_host, int* fused_host) {
    int fa, fb;
    if (op_flag(ta, &fa)) return -2;
    if (op_flag(tb, &fb)) return -3;
    double* d_ssq = (double*)(c->d_mail + 40);
    hipLaunchKernelGGL(rlhip_zero_f64_kernel, dim3(1), dim3(1), 0, c->stream, d_ssq);
    int done = 0;
    int rc = rlhip::gemm_impl<double>(c, fa, fb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0, d_ssq, &done);
    if (rc) return rc;
    if (fused_host) *fused_host = done ? 1 : 0;
    const int64_t arows = fa ? k : m, acols = fa ? m : k;
    c->norma_state = 0;
    c->norma_reduced = 0;
    if (done == 2 && norm_a_host == nullptr) {
        // deferred: the sum of squares travels to the pinned mailbox behind the stream; rlhip_norma_collect_f64 picks it up after whatever
        // synchronisation comes next (QB reads ||A||_F and ||B_i||_F with ONE host round trip this way)
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 40, d_ssq, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        c->norma_state = 1;
        c->norma_epoch = c->sync_epoch;
        return 0;
    }
    double result = 0;
    if (!done) {
        rc = rlhip::lange_fro<double>(c, arows, acols, A, lda, &result);
        if (rc) return rc;
    } else {
        double ssq_main = 0, rest = 0;
        RLHIP_CHECK(hipMemcpyAsync(c->h_mail + 40, d_ssq, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        RLHIP_CHECK(rlhip_stream_sync(c));
        ssq_main = *(double*)(c->h_mail + 40);
        const int64_t m_main = (m / 128) * 128;
        if (m_main < m && done != 2) {   // rows (or, for op = T, columns) peeled off to the generic kernel
            const double* A2 = fa ? (A + m_main * lda) : (A + m_main);
            rc = rlhip::lange_fro<double>(c, fa ? k : (m - m_main), fa ? (m - m_main) : k, A2, lda, &rest);
            if (rc) return rc;
        }
        result = sqrt(ssq_main + rest * rest);
    }
    if (norm_a_host) *norm_a_host = result;
    else { c->norma_value = result; c->norma_state = 2; }
    return 0;
}

int rlhip_norma_collect_f64(rlhip_ctx* c, int over_ranks, double* norm_a_host) {
    if (!norm_a_host) return -3;
    double ssq = 0;
    bool global = false;
    if (c->norma_state == 1) {
        if (c->sync_epoch == c->norma_epoch) RLHIP_CHECK(rlhip_stream_sync(c));      // (normally a later call has already waited on the stream: no second round trip)
        if (over_ranks && c->norma_reduced) { ssq = *(double*)(c->h_mail + 41); global = true; }
        else ssq = *(double*)(c->h_mail + 40);
    } else if (c->norma_state == 2) {
        ssq = c->norma_value * c->norma_value;
    } else {
        return -4;                                               // nothing pending
    }
    c->norma_state = 0;
    c->norma_reduced = 0;
    if (over_ranks && !global && rlhip_comm_size(c) > 1) {       // no all-reduce has carried it yet: one of its own
        const int rc = rlhip_allreduce_sum_host_f64(c, &ssq, 1);
        if (rc) return rc;
    }
    *norm_a_host = sqrt(ssq);
    return 0;
}

RLHIP_DEFINE_BLAS3(f64, double)
RLHIP_DEFINE_BLAS3(f32, float)

}  // extern "C"

// ------------------------------------------------------------------ microbenchmarks
namespace {
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_peak_f64_kernel(int iters, double* out) {
    d4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a7, 0, 0, 0);
    }
    d4_t s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s[0] == 12345.678) out[0] = s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void mfma_peak_f32_kernel(int iters, float* out) {
    f4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    float x = 1.0f + threadIdx.x * 1e-6f, y = 1.0f - threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a7, 0, 0, 0);
    }
    f4_t s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s[0] == 12345.678f) out[0] = s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void hbm_read_kernel(const double2* __restrict__ p, size_t n16, double* out) {
    double acc = 0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}
// keeps `blockDim.x / 64` waves per workgroup busy for `ticks` of the 100 MHz wall clock: mode 0 s_sleep, 1 fp64 FMA chain, 2 fp64 MFMA stream
__global__ __launch_bounds__(256) void dvfs_burn_kernel(int mode, long long ticks, double* out) {
    const long long t0 = wall_clock64();
    d4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9, z = 0.5;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(64);
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 64; ++i) z = fma(z, x, y);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
        }
    }
    d4_t s = a0 + a1 + a2 + a3;
    if (s[0] + z == 12345.678) out[0] = s[1];
}
}  // namespace

// diagnostic: occupy `blocks` workgroups of 256 threads for `usec` microseconds (mode 0 sleeping, 1 fp64 FMA, 2 fp64 MFMA) on the context's
// stream (side = 0) or on a second stream beside it (side = 1).  scripts/dvfs_probe.py uses it to map how the part's clock follows the load.
extern "C" int rlhip_dvfs_burn(rlhip_ctx* c, int blocks, int mode, int usec, int side) {
    if (blocks <= 0 || usec <= 0) return 0;
    hipStream_t st = c->stream;
    if (side) {
        if (!c->side) RLHIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        st = c->side;
    }
    hipLaunchKernelGGL(dvfs_burn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mode, (long long)usec * 100, (double*)c->d_mail + 32);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t rlhip_path_count(rlhip_ctx* c, int which) {
    return (c && which >= 0 && which < 16) ? c->path_count[which] : -1;
}
extern "C" int rlhip_path_note(rlhip_ctx* c, int which, int64_t delta) {
    if (!c || which < 0 || which >= 16) return -1;
    c->path_count[which] += delta;
    return 0;
}

// ---- profiler phase markers: roctx bound at run time (no link-time dependency on the profiler SDK)
namespace {
typedef int (*fn_roctx_push)(const char*);
typedef int (*fn_roctx_pop)(void);
struct Roctx { int state = 0; fn_roctx_push push = nullptr; fn_roctx_pop pop = nullptr; };   // state: 0 untried, 1 bound, -1 absent
Roctx g_roctx;
void roctx_bind() {
    g_roctx.state = -1;
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    void* h = nullptr;
    const char* want = getenv("RLHIP_ROCTX");
    if (want && want[0] == '0') return;                                                             // RLHIP_ROCTX=0: never
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }          // a profiler brought it along
    if (!h && want && want[0] == '1')
        for (const char* n : names) { h = dlopen(n, RTLD_NOW); if (h) break; }
    if (!h) return;
    g_roctx.push = (fn_roctx_push)dlsym(h, "roctxRangePushA");
    g_roctx.pop = (fn_roctx_pop)dlsym(h, "roctxRangePop");
    if (g_roctx.push && g_roctx.pop) g_roctx.state = 1;
}
}  // namespace
extern "C" int rlhip_range_push(const char* name) {
    if (g_roctx.state == 0) roctx_bind();
    if (g_roctx.state == 1 && name) g_roctx.push(name);
    return 0;
}
extern "C" int rlhip_range_pop(void) {
    if (g_roctx.state == 1) g_roctx.pop();
    return 0;
}
extern "C" int rlhip_avoid_persistent(rlhip_ctx* c, int on) {
    if (!c) return -1;
    const int prev = c->avoid_persistent;
    c->avoid_persistent = on ? 1 : 0;
    return prev;
}

extern "C" int rlhip_mfma_peak(rlhip_ctx* c, int is_f64, int iters, double* tflops) {
    const int blocks = 256 * 8, threads = 256;  // 2 waves per SIMD
    double* d = (double*)c->d_mail;
    for (int rep = 0; rep < 2; ++rep) {
        RLHIP_CHECK(hipEventRecord(c->ev0, c->stream));
        if (is_f64)
            hipLaunchKernelGGL(mfma_peak_f64_kernel, dim3(blocks), dim3(threads), 0, c->stream, iters, d);
        else
            hipLaunchKernelGGL(mfma_peak_f32_kernel, dim3(blocks), dim3(threads), 0, c->stream, iters, (float*)d);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipEventRecord(c->ev1, c->stream));
        RLHIP_CHECK(hipEventSynchronize(c->ev1));
    }
    float ms = 0;
    RLHIP_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    double flops = (double)blocks * (threads / 64) * (double)iters * 8.0 * (2.0 * 16 * 16 * 4);
    *tflops = flops / (ms * 1e-3) / 1e12;
    return 0;
}

extern "C" int rlhip_hbm_read_peak(rlhip_ctx* c, const void* buf, size_t bytes, double* gbps) {
    size_t n16 = bytes / 16;
    double* d = (double*)c->d_mail;
    for (int rep = 0; rep < 2; ++rep) {
        RLHIP_CHECK(hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(hbm_read_kernel, dim3(256 * 16), dim3(256), 0, c->stream, (const double2*)buf, n16, d);
        RLHIP_LAUNCH_CHECK();
        RLHIP_CHECK(hipEventRecord(c->ev1, c->stream));
        RLHIP_CHECK(hipEventSynchronize(c->ev1));
    }
    float ms = 0;
    RLHIP_CHECK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *gbps = (double)(n16 * 16) / (ms * 1e-3) / 1e9;
    return 0;
}
