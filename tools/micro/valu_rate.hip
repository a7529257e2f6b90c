// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction for v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_mul
// at 1, 2, 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 1) {   // 4 independent v_pk_fma_f32 (= 8 fmas)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
        } else if (MODE == 2) {   // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) {   // 8 v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 4) {   // mixed like the scan step: mul, exp, mul, fma, fma  x 8 (independent chains)
#define STEP(x) "v_mul_f32 %" #x ", %" #x ", %8\n v_exp_f32 %" #x ", %" #x "\n v_mul_f32 %" #x ", %" #x ", %8\n v_fma_f32 %" #x ", %" #x ", %8, %9\n v_fma_f32 %" #x ", %" #x ", %8, %9\n"
            asm volatile(STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int MODE>
void run(const char* name, int instr_per_iter, float* d) {
    const int iters = 4000;
    for (int wps : {1, 2, 4, 8}) {                 // waves per SIMD: block of wps*4 waves, one block per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<256, wps * 256>>>(d, iters, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<256, wps * 256>>>(d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float ticks; hipMemcpy(&ticks, d, 4, hipMemcpyDeviceToHost);
        // s_memtime ticks are shader cycles (per doc); wall-based estimate assumes 2.4 GHz
        printf("%-12s waves/SIMD=%d  wall=%.3f ms  cycles/wave-instr (memtime)=%.2f  per-SIMD cycles/instr (wall@2.1GHz)=%.2f\n", name, wps, ms,
               ticks / (double)(iters * instr_per_iter), ms * 1e-3 * 2.1e9 / (double)(iters * instr_per_iter * wps));
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4);
    run<0>("v_fma_f32", 8, d);
    run<1>("v_pk_fma_f32", 4, d);
    run<2>("v_exp_f32", 8, d);
    run<3>("v_mul_f32", 8, d);
    run<4>("scan-mix", 40, d);
    return 0;
}
