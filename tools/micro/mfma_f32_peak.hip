// Sustained rate of v_mfma_f32_32x32x2_f32 with every CU issuing (operands in registers): the ceiling an fp32 GEMM can reach
// at the clock the chip holds under this load.  Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float x = 1.f + threadIdx.x * 1e-3f, y = 0.5f - threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
    float* d; (void)hipMalloc(&d, 1024 * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    for (int blocks : {256, 512}) {
        const int iters = 20000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, clk, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, clk, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 /*waves*/ * iters * 16 /*mfma*/ * 32.0 * 32 * 2 * 2;
        printf("blocks/CU=%d: %.2f ms, %.1f TFLOP/s, in-kernel clock %.2f GHz\n", blocks / 256, ms, flops / ms / 1e9, (double)h[0] / h[1] * 0.1);
    }
    return 0;
}
