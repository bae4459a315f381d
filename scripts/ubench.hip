// MFMA f64 issue-rate microbenchmark variants (diagnostic, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(int iters, double* out, long long* cyc, double xv, double yv) {
    d4 a[NACC];
    for (int i = 0; i < NACC; ++i) a[i] = d4{0, 0, 0, 0};
    double x = xv + threadIdx.x * 1e-3, y = yv - threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) a[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[j], 0, 0, 0);
    }
    long long t1 = clock64();
    d4 s = a[0];
    for (int i = 1; i < NACC; ++i) s += a[i];
    if (s[0] == 12345.678) out[0] = s[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
void run(int blocks, int threads, int iters, double xv, double yv) {
    double* d; long long* c; hipMalloc(&d, 8); hipMalloc(&c, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, iters, d, c, xv, yv);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    double fl = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
    printf("nacc=%d blocks=%d thr=%d x=%g: %.2f TF, %.3f ms, wave cycles/mfma = %.1f (clock64), eff clock if 64cyc/mfma: %.2f GHz\n", NACC, blocks, threads, xv,
           fl / ms / 1e9, ms, (double)cy / ((double)iters * NACC), fl/ms/1e9/78.6*2.4);
}
int main() {
    int it = 4000;
    run<4>(256, 256, it, 1.0, 1.0);     // 1 wave/SIMD
    run<8>(256, 256, it, 1.0, 1.0);
    run<12>(256, 256, it, 1.0, 1.0);
    run<4>(256, 512, it, 1.0, 1.0);     // 2 waves/SIMD
    run<8>(256, 512, it, 1.0, 1.0);
    run<12>(256, 512, it, 1.0, 1.0);
    run<8>(512, 512, it, 1.0, 1.0);     // 4 waves/SIMD
    run<8>(256, 512, it, 0.0, 0.0);
    run<8>(256, 512, 40000, 1.0, 1.0);
    return 0;
}
