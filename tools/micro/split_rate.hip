// Throughput of the three-way bf16 operand split (the vector work of resel_gemm_f32's split modes) per SIMD, alone and with a
// second / third / fourth wave on the SIMD: cycles per split of 8 floats (44 vector instructions: 12 v_perm_b32, 16 v_and_b32,
// 16 v_sub_f32).  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize split_rate.hip -o split_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void split8(const float (&x)[8], uint32_t (&w1)[4], uint32_t (&w2)[4], uint32_t (&w3)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t u0 = __float_as_uint(x[2 * q]), u1 = __float_as_uint(x[2 * q + 1]);
        w1[q] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        const float r0 = x[2 * q] - __uint_as_float(u0 & 0xffff0000u), r1 = x[2 * q + 1] - __uint_as_float(u1 & 0xffff0000u);
        const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
        w2[q] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
        const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
        w3[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
}
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* clk, int iters) {
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = 1.f + threadIdx.x * 1e-3f + j * 0.37f;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t w1[4], w2[4], w3[4];
            split8(x, w1, w2, w3);
#pragma unroll
            for (int q = 0; q < 4; ++q) { asm volatile("" : "+v"(w1[q]), "+v"(w2[q]), "+v"(w3[q])); }
            acc ^= w3[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(acc) + x[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    for (int wps : {1, 2, 3, 4}) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(256), dim3(wps * 256), 0, 0, d, clk, iters); (void)hipDeviceSynchronize(); }
        unsigned long long h; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        const double per = (double)h / (iters * 4.0);
        printf("%d wave(s) per SIMD: %.1f cycles per split of 8 floats per wave (%.2f per instruction), per SIMD %.1f cycles per split\n", wps, per, per / 45.0, per / wps);
    }
    return 0;
}
