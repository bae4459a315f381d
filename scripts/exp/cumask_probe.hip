// Probe: does a stream created with hipExtStreamCreateWithCUMask confine its workgroups to the masked CUs on this device, and how are the
// mask bits laid out?  Prints, per mask, the number of distinct (XCC, SE, CU) triples the workgroups of a 2048-WG kernel ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    // burn a little so that workgroups spread
    float x = threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (x == 123.f) out[0] = 0;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    const int G = 4096;
    unsigned* d; hipMalloc(&d, G * 8);
    std::vector<unsigned> h(2 * G);
    for (int nbits : {0, 32, 64, 128, 256}) {
        hipStream_t st;
        hipError_t e;
        if (nbits == 0) e = hipStreamCreate(&st);
        else {
            std::vector<uint32_t> mask(8, 0);
            for (int b = 0; b < nbits; ++b) mask[b / 32] |= 1u << (b % 32);
            e = hipExtStreamCreateWithCUMask(&st, 8, mask.data());
        }
        if (e != hipSuccess) { printf("mask %d: create failed: %s\n", nbits, hipGetErrorString(e)); continue; }
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(where_kernel, dim3(G), dim3(256), 0, st, d);
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        hipLaunchKernelGGL(where_kernel, dim3(G), dim3(256), 0, st, d);
        hipEventRecord(b, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        std::set<unsigned long long> cus; std::set<unsigned> xccs;
        for (int i = 0; i < G; ++i) {
            const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            cus.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
        }
        printf("mask bits %3d: %zu distinct CUs on %zu XCCs, kernel %.3f ms\n", nbits, cus.size(), xccs.size(), ms);
        hipStreamDestroy(st);
    }
    return 0;
}
