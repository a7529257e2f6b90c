// Can vector instructions run in the shadow of v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles)?  Loop body: 4 independent
// MFMAs (4 accumulators) with NV independent v_fma_f32 after each; cycles per MFMA for NV = 0..10, 1 and 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    uint4 ub = make_uint4(0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    bf16x8 x = __builtin_bit_cast(bf16x8, ub), y = x;
    float v[12];
    for (int j = 0; j < 12; ++j) v[j] = 1.f + threadIdx.x * 1e-3f + j;
    const float m = 1.0000001f, c = 1e-9f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    for (int j = 0; j < 12; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NV>
void run(float* d, unsigned long long* clk, int threads) {
    const int iters = 4000;
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(threads), 0, 0, d, clk, iters);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(threads), 0, 0, d, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("  NV=%2d: %6.1f cycles per MFMA (per wave)\n", NV, (double)h / (iters * 4.0));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    for (int threads : {256, 512}) {
        printf("%d wave(s) per SIMD (s_memtime cycles):\n", threads / 256);
        run<0>(d, clk, threads); run<2>(d, clk, threads); run<4>(d, clk, threads); run<6>(d, clk, threads);
        run<8>(d, clk, threads); run<10>(d, clk, threads); run<12>(d, clk, threads);
    }
    return 0;
}
