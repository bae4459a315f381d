// TEST INFRASTRUCTURE ONLY (see lapack_bind.h).
#include "lapack_bind.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <string>

namespace orc {

static Lapack g_lp;
static char g_err[512];

Lapack& lapack() { return g_lp; }

template <typename F>
static bool bind(void* h, const char* prefix, const char* name, F& fn) {
    std::string s = std::string(prefix) + name;
    void* p = dlsym(h, s.c_str());
    if (!p) {
        snprintf(g_err, sizeof(g_err), "symbol %s not found", s.c_str());
        return false;
    }
    fn = reinterpret_cast<F>(p);
    return true;
}

const char* lapack_open(const char* path) {
    if (g_lp.handle) return nullptr;
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        snprintf(g_err, sizeof(g_err), "dlopen(%s): %s", path, dlerror());
        return g_err;
    }
    const char* prefix = dlsym(h, "scipy_dgemm_") ? "scipy_" : "";
    Lapack L;
    L.handle = h;
    L.prefix = prefix;
    bool ok = bind(h, prefix, "dgemm_", L.dgemm) && bind(h, prefix, "dsyrk_", L.dsyrk) &&
              bind(h, prefix, "dtrsm_", L.dtrsm) && bind(h, prefix, "dtrmm_", L.dtrmm) &&
              bind(h, prefix, "dger_", L.dger) && bind(h, prefix, "dscal_", L.dscal) &&
              bind(h, prefix, "dpotrf_", L.dpotrf) && bind(h, prefix, "dgesdd_", L.dgesdd) &&
              bind(h, prefix, "dgeqrf_", L.dgeqrf) && bind(h, prefix, "dorgqr_", L.dorgqr) &&
              bind(h, prefix, "dormqr_", L.dormqr) && bind(h, prefix, "dgeqp3_", L.dgeqp3) &&
              bind(h, prefix, "dgetrf_", L.dgetrf) && bind(h, prefix, "dlaswp_", L.dlaswp) &&
              bind(h, prefix, "dlange_", L.dlange) && bind(h, prefix, "dlacpy_", L.dlacpy) &&
              bind(h, prefix, "dlaset_", L.dlaset) && bind(h, prefix, "dlapmt_", L.dlapmt) &&
              bind(h, prefix, "dorhr_col_", L.dorhr_col) && bind(h, prefix, "dgeqrt_", L.dgeqrt) &&
              bind(h, prefix, "dgemqrt_", L.dgemqrt) && bind(h, prefix, "dlarfg_", L.dlarfg) &&
              bind(h, prefix, "dlarf_", L.dlarf) && bind(h, prefix, "dlarfb_", L.dlarfb) &&
              bind(h, prefix, "dlarft_", L.dlarft) && bind(h, prefix, "dnrm2_", L.dnrm2) &&
              bind(h, prefix, "idamax_", L.idamax) && bind(h, prefix, "dswap_", L.dswap) &&
              bind(h, prefix, "sgemm_", L.sgemm) && bind(h, prefix, "ssyrk_", L.ssyrk) && bind(h, prefix, "strsm_", L.strsm) &&
              bind(h, prefix, "strmm_", L.strmm) && bind(h, prefix, "spotrf_", L.spotrf) && bind(h, prefix, "sgeqrf_", L.sgeqrf) &&
              bind(h, prefix, "sormqr_", L.sormqr) && bind(h, prefix, "sgeqp3_", L.sgeqp3) && bind(h, prefix, "sgetrf_", L.sgetrf) &&
              bind(h, prefix, "slacpy_", L.slacpy) && bind(h, prefix, "slaset_", L.slaset) && bind(h, prefix, "sorhr_col_", L.sorhr_col) &&
              bind(h, prefix, "sgeqrt_", L.sgeqrt) && bind(h, prefix, "sgemqrt_", L.sgemqrt) && bind(h, prefix, "sorgqr_", L.sorgqr);
    if (!ok) {
        dlclose(h);
        return g_err;
    }
    L.set_threads = nullptr;
    L.get_threads = nullptr;
    {
        std::string s = std::string(prefix) + "openblas_set_num_threads";
        void* p = dlsym(h, s.c_str());
        if (p) L.set_threads = reinterpret_cast<void (*)(int)>(p);
        s = std::string(prefix) + "openblas_get_num_threads";
        p = dlsym(h, s.c_str());
        if (p) L.get_threads = reinterpret_cast<int (*)(void)>(p);
    }
    g_lp = L;
    return nullptr;
}

}  // namespace orc
