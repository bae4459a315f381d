// Row-block sharding support: sum all-reduce of small replicated factors (k x k Gram, n x k B^T, ...) across
// the GPUs of one node, enqueued on the context's HIP stream via RCCL over xGMI.
//
// The reference has no distributed code at all; this is new design.  RCCL is
// bound at run time (dlopen) so librlhip.so loads on machines without it, and so that a process that already
// carries PyTorch's RCCL shares that copy.  One process per GPU; the ncclUniqueId is created by rank 0
// (rlhip_comm_unique_id) and distributed by whatever rendezvous the host program already has
// (torch.distributed in bench.py).  Alternatively a host callback can be installed
// (rlhip_comm_set_hook) -- used by hosts that own their collectives.
#include "rlhip_internal.h"
#include <cstdlib>
#include "../../include/rlhip.h"
#include <dlfcn.h>
#include <cstring>

namespace {

struct NcclId { char internal[128]; };
typedef void* ncclComm_t;
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(ncclComm_t*, int, NcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*fn_destroy)(ncclComm_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
};
Rccl g_rccl;

const char* g_rccl_origin = "";

// Bind RCCL.  A process that already maps an RCCL (PyTorch ships its own librccl.so inside torch/lib) must not get a SECOND copy:
// two RCCL instances in one process each bring their own bootstrap / proxy state.  So: (1) look for an already-loaded copy by
// soname (RTLD_NOLOAD finds a library whatever directory it came from), (2) look for a mapped file named librccl* in
// /proc/self/maps and re-open exactly that path, (3) only then load the system copy.
int load_rccl() {
    if (g_rccl.h) return 0;
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (h) { g_rccl_origin = "already loaded (soname)"; break; }
    }
    if (!h) {
        if (FILE* f = fopen("/proc/self/maps", "r")) {
            char line[1024];
            static char path[1024];
            while (!h && fgets(line, sizeof line, f)) {
                char* p = strstr(line, "/");
                if (!p || !strstr(p, "librccl")) continue;
                size_t len = strcspn(p, "\n");
                if (len >= sizeof path) continue;
                memcpy(path, p, len); path[len] = 0;
                h = dlopen(path, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
                if (h) g_rccl_origin = path;
            }
            fclose(f);
        }
    }
    if (!h) {
        const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : fresh) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) { g_rccl_origin = n; break; }
        }
    }
    if (!h) { fprintf(stderr, "[rlhip] cannot load RCCL: %s\n", dlerror()); return -1; }
    g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.allreduce || !g_rccl.destroy) return -1;
    g_rccl.h = h;
    return 0;
}

}  // namespace

// per-context communicator state lives in a side table keyed by ctx (keeps rlhip_ctx POD-simple)
struct rlhip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    rlhip_allreduce_hook hook = nullptr;
    void* hook_user = nullptr;
};

static rlhip_comm* comm_of(rlhip_ctx* c) { return (rlhip_comm*)c->comm; }

extern "C" {

/* 1 when RCCL can be bound in this process (ncclCommInitRank is collective: every rank must know that EVERY rank can join before
 * any of them calls rlhip_comm_init).  rlhip_comm_rccl_origin: where the bound copy came from (diagnostics). */
int rlhip_comm_can_load(void) { return load_rccl() == 0 ? 1 : 0; }
const char* rlhip_comm_rccl_origin(void) { return g_rccl_origin; }
/* ncclGetVersion of the bound RCCL (e.g. 22203), 0 when RCCL cannot be bound or has no such entry point */
int rlhip_comm_rccl_version(void) {
    if (load_rccl()) return 0;
    typedef int (*fn_version)(int*);
    fn_version f = (fn_version)dlsym(g_rccl.h, "ncclGetVersion");
    int v = 0;
    if (!f || f(&v)) return 0;
    return v;
}
/* 1 when this context's collectives run through its own RCCL communicator, 2 through the caller's hook, 0 when it has neither (one rank) */
int rlhip_comm_kind(rlhip_ctx* c) {
    if (!c->comm) return 0;
    rlhip_comm* cm = (rlhip_comm*)c->comm;
    return cm->hook ? 2 : (cm->comm ? 1 : 0);
}

int rlhip_comm_unique_id(unsigned char id_out[128]) {
    if (load_rccl()) return -1001;
    NcclId id;
    int rc = g_rccl.get_id(&id);
    if (rc) return -1100 - rc;
    memcpy(id_out, id.internal, 128);
    return 0;
}

int rlhip_comm_init(rlhip_ctx* c, int nranks, int rank, const unsigned char id[128]) {
    if (nranks < 1 || rank < 0 || rank >= nranks) return -2;
    if (!c->comm) c->comm = new rlhip_comm();
    rlhip_comm* cm = comm_of(c);
    cm->rank = rank;
    cm->nranks = nranks;
    // one rank needs no communicator; RLHIP_COMM_SINGLE_RANK_NCCL=1 builds one anyway so that a 1-GPU box can exercise the real
    // ncclCommInitRank / ncclAllReduce bindings (tests/test_gpu_drivers.py::test_comm_world1_allreduce_is_identity)
    if (nranks == 1 && !getenv("RLHIP_COMM_SINGLE_RANK_NCCL")) return 0;
    if (load_rccl()) return -1001;
    RLHIP_CHECK(hipSetDevice(c->device));
    NcclId nid;
    memcpy(nid.internal, id, 128);
    int rc = g_rccl.init_rank(&cm->comm, nranks, nid, rank);
    if (rc) {
        fprintf(stderr, "[rlhip] ncclCommInitRank: %s\n", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
        return -1100 - rc;
    }
    return 0;
}

int rlhip_comm_set_hook(rlhip_ctx* c, rlhip_allreduce_hook hook, void* user, int nranks, int rank) {
    if (!c->comm) c->comm = new rlhip_comm();
    rlhip_comm* cm = comm_of(c);
    cm->hook = hook;
    cm->hook_user = user;
    cm->nranks = nranks;
    cm->rank = rank;
    return 0;
}

int rlhip_comm_size(rlhip_ctx* c) { return c->comm ? comm_of(c)->nranks : 1; }
int rlhip_comm_rank(rlhip_ctx* c) { return c->comm ? comm_of(c)->rank : 0; }

int rlhip_comm_destroy(rlhip_ctx* c) {
    if (!c->comm) return 0;
    rlhip_comm* cm = comm_of(c);
    if (cm->comm && g_rccl.destroy) {
        rlhip_stream_sync(c);
        g_rccl.destroy(cm->comm);
    }
    delete cm;
    c->comm = nullptr;
    return 0;
}

static int allreduce_impl(rlhip_ctx* c, void* buf, int64_t count, int is_f64) {
    if (count <= 0 || !c->comm) return 0;
    rlhip_comm* cm = comm_of(c);
    if (cm->nranks <= 1 && !cm->comm) return 0;
    if (cm->hook) return cm->hook(cm->hook_user, buf, count, is_f64);
    if (!cm->comm) return -1002;
    int rc = g_rccl.allreduce(buf, buf, (size_t)count, is_f64 ? 8 /*ncclFloat64*/ : 7 /*ncclFloat32*/, 0 /*ncclSum*/,
                              cm->comm, c->stream);
    if (rc) {
        fprintf(stderr, "[rlhip] ncclAllReduce: %s\n", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
        return -1100 - rc;
    }
    return 0;
}

int rlhip_allreduce_sum_f64(rlhip_ctx* c, double* buf, int64_t count) { return allreduce_impl(c, buf, count, 1); }
int rlhip_allreduce_sum_f32(rlhip_ctx* c, float* buf, int64_t count) { return allreduce_impl(c, buf, count, 0); }

int rlhip_allreduce_sum_host_f64(rlhip_ctx* c, double* x_host, int64_t n) {
    if (n <= 0 || n > 16 || !c->comm || comm_of(c)->nranks <= 1) return (n > 16) ? -3 : 0;
    double* d = (double*)(c->d_mail + 32);
    RLHIP_CHECK(hipMemcpyAsync(d, x_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = allreduce_impl(c, d, n, 1);
    if (rc) return rc;
    RLHIP_CHECK(hipMemcpyAsync(x_host, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RLHIP_CHECK(rlhip_stream_sync(c));
    return 0;
}

}  // extern "C"
