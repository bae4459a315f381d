// Do two CU-masked streams run their kernels at the same time?  A burn kernel on a 64-CU stream and one on the complementary 192-CU stream,
// each sized for ~4 ms alone; wall time of both launched together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void burn(float* out, int iters) {
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 123.f) out[0] = x;
}
static hipStream_t masked(int first, int count) {
    std::vector<uint32_t> mask(8, 0);
    for (int b = first; b < first + count; ++b) mask[b / 32] |= 1u << (b % 32);
    hipStream_t st; if (hipExtStreamCreateWithCUMask(&st, 8, mask.data()) != hipSuccess) { printf("create failed\n"); exit(1); }
    return st;
}
int main() {
    float* d; (void)hipMalloc(&d, 1024);
    hipStream_t s64 = masked(0, 64), s192 = masked(64, 192), plain; (void)hipStreamCreate(&plain);
    auto run = [&](hipStream_t a, int ga, hipStream_t b, int gb, const char* tag) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            if (ga) hipLaunchKernelGGL(burn, dim3(ga), dim3(256), 0, a, d, 400000);
            if (gb) hipLaunchKernelGGL(burn, dim3(gb), dim3(256), 0, b, d, 400000);
            (void)hipDeviceSynchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%-48s %.3f ms\n", tag, ms);
        }
    };
    run(s64, 64 * 8, s64, 0, "64-CU stream alone (8 WGs per CU)");
    run(s192, 192 * 8, s192, 0, "192-CU stream alone (8 WGs per CU)");
    run(s64, 64 * 8, s192, 192 * 8, "both at once");
    run(plain, 256 * 8, plain, 0, "unmasked stream, 256 x 8 WGs");
    run(s64, 64 * 8, plain, 192 * 8, "64-CU stream + unmasked stream (192 x 8 WGs)");
    return 0;
}
