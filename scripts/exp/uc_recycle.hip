// Does a VA recycled from an UNCACHED allocation (hipExtMallocWithFlags, hipDeviceMallocUncached) behave like ordinary memory after hipFree + hipMalloc?
// build: hipcc --offload-arch=gfx950 -O2 uc_recycle.hip -o uc_recycle.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tagw(unsigned long long* p, size_t n, unsigned tag) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long w = ((unsigned long long)tag << 32) | (unsigned)i;
        __hip_atomic_store(p + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void pollr(const unsigned long long* p, size_t n, unsigned long long* acc) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicAdd(acc, s);
}
__global__ void fillk(double* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0 + (double)i;
}
__global__ void checkk(const double* p, size_t n, unsigned long long* bad) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long b = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) b += (p[i] != 1.0 + (double)i);
    if (b) atomicAdd(bad, b);
}
int main() {
    const size_t bytes = 4u << 20, n = bytes / 8;
    unsigned long long *acc, h = 0;
    CK(hipMalloc(&acc, 16));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 6; ++rep) {
        void* uc = nullptr;
        CK(hipExtMallocWithFlags(&uc, bytes, hipDeviceMallocUncached));
        CK(hipMemsetAsync(uc, 0, bytes, s2));
        tagw<<<256, 256, 0, s2>>>((unsigned long long*)uc, n, 7u + rep);
        pollr<<<256, 256, 0, s2>>>((const unsigned long long*)uc, n, acc);
        CK(hipStreamSynchronize(s2));
        CK(hipFree(uc));
        void* p = nullptr;
        CK(hipMalloc(&p, bytes));
        CK(hipMemsetAsync(acc, 0, 16, s1));
        fillk<<<1024, 256, 0, s1>>>((double*)p, n);
        checkk<<<1024, 256, 0, s1>>>((const double*)p, n, acc);
        CK(hipMemcpyAsync(&h, acc, 8, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1));
        printf("rep %d: uncached block %p, recycled as %p (%s): %llu of %zu doubles read back wrong\n", rep, uc, p, uc == p ? "SAME address" : "different", h, n);
        CK(hipFree(p));
    }
    return 0;
}
