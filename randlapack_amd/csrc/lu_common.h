// Shared by lu.hip (general panel kernels, host driver) and lu_f32.hip (the fp32 panel step): argument block, register state and the
// batched tagged-word fetch.  Two translation units because each unrolls 32 column steps and takes minutes to compile: side by side
// the build is as long as the longer one.
#pragma once
#include "rlhip_internal.h"

namespace rlhip_lu {

constexpr int PB = 32;

template <typename T>
struct LuArgs {
    int64_t m, n;             // full matrix
    T* A; int64_t lda;
    int64_t j0; int pb;       // panel [j0, j0+pb)
    int64_t* ipiv;            // 1-based, device
    T* cand_val; int64_t* cand_row;   // 2 x G
    T* cand_data;             // 2 x G x PB  : candidate row contents
    T* diag_data;             // 2 x PB      : contents of the current diagonal row
    unsigned* bar;
    int* info;                // first zero pivot (1-based), 0 if none
    int64_t rpw;              // rows per workgroup
    unsigned long long* tw;   // tagged 8-byte words of the flag-less exchange (fp32 register kernel): 2 x (2 G + G PB + PB)
    unsigned tag_base;        // tags of this launch are tag_base + 1 .. tag_base + PB (unique across launches)
};

// N tagged words in ONE batch of loads (re-read together until every needed word carries the tag): data that is already there costs
// a single round trip however many words a thread needs
template <int N>
__device__ __forceinline__ void lu_tag_get_n(const unsigned long long* const (&ad)[N], const bool (&need)[N], unsigned tag, unsigned (&out)[N], int* info) {
    for (int spins = 0;; ++spins) {
        unsigned long long w[N];
#pragma unroll
        for (int i = 0; i < N; ++i) w[i] = need[i] ? __hip_atomic_load(ad[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && ((unsigned)(w[i] >> 32) == tag);
        if (ok || spins > (1 << 22)) {
            if (!ok) atomicExch(info, -7);
#pragma unroll
            for (int i = 0; i < N; ++i) out[i] = (unsigned)w[i];
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <typename T, int RPT>
struct LuRegState {
    T x[RPT][PB];
    int64_t gr[RPT];
#ifdef RLHIP_LU_PROF
    long long pf[5], pt;
#endif
};
#ifdef RLHIP_LU_PROF
#define LU_MARK(i) { const long long now_ = wall_clock64(); st.pf[i] += now_ - st.pt; st.pt = now_; }
#else
#define LU_MARK(i)
#endif

// launcher of the fp32 panel step (lu_f32.hip): grid G <= 64 workgroups of 256 threads, 1024 rows each
void launch_getrf_panel_f32(const LuArgs<float>& g, unsigned G, hipStream_t stream);
// launcher of the fp64 panel step (lu_f64.hip): G <= 64 workgroups of 256 threads, 512 rows each
void launch_getrf_panel_f64(const LuArgs<double>& g, unsigned G, hipStream_t stream);

}  // namespace rlhip_lu
