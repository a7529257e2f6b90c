// VALU issue-rate microbenchmark for gfx950 (second edition): long runs (tens of ms, so the clock has ramped), per-SIMD
// throughput from wall time, the in-kernel clock from s_memtime / s_memrealtime.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define R8_(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R8(OP) R8_(OP) R8_(OP) R8_(OP) R8_(OP) R8_(OP) R8_(OP) R8_(OP) R8_(OP)   /* 64 instructions per loop trip: the taken branch costs ~28 cycles */
#define FMA(x) "v_fma_f32 %" #x ", %" #x ", %8, %9\n"
#define MUL(x) "v_mul_f32 %" #x ", %" #x ", %8\n"
#define EXP(x) "v_exp_f32 %" #x ", %" #x "\n"
#define MULLO(x) "v_mul_lo_u32 %" #x ", %" #x ", %8\n"
#define MAD24(x) "v_mad_u32_u24 %" #x ", %" #x ", %8, %9\n"
#define XOR(x) "v_xor_b32 %" #x ", %" #x ", %8\n"
#define LSHR(x) "v_lshrrev_b32 %" #x ", 3, %" #x "\n"
#define CND(x) "v_cndmask_b32_e64 %" #x ", %" #x ", %8, %10\n"
#define MIX(x) "v_mul_f32 %" #x ", %" #x ", %8\n v_exp_f32 %" #x ", %" #x "\n v_mul_f32 %" #x ", %" #x ", %8\n v_fma_f32 %" #x ", %" #x ", %8, %9\n v_fma_f32 %" #x ", %" #x ", %8, %9\n"
// the forward step per state PAIR on packed math: pk_mul, 2 exp, pk_mul, pk_fma, pk_fma  (4 pairs = 8 states)
#define PK(x) "v_pk_mul_f32 %" #x ", %" #x ", %4\n v_exp_f32 %" #x ", %" #x "\n v_pk_mul_f32 %" #x ", %" #x ", %4\n v_pk_fma_f32 %" #x ", %" #x ", %4, %5\n v_pk_fma_f32 %" #x ", %" #x ", %4, %5\n"

#define PKF "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
#define PKM "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
template <int MODE>
__global__ void k(float* out, unsigned long long* clk, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
    const unsigned long long msk = __ballot(threadIdx.x & 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(R8(FMA) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        else if (MODE == 1) asm volatile(PKF PKF PKF PKF PKF PKF PKF PKF PKF PKF PKF PKF PKF PKF PKF PKF
                                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
        else if (MODE == 2) asm volatile(R8(EXP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        else if (MODE == 3) asm volatile(R8(MUL) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        else if (MODE == 4) asm volatile(R8(MIX) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        else if (MODE == 5) asm volatile(R8(MULLO) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        else if (MODE == 6) asm volatile(R8(MAD24) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        else if (MODE == 7) asm volatile(R8(XOR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        else if (MODE == 8) asm volatile(R8(CND) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(msk));
        else if (MODE == 9) asm volatile(PKM PKM PKM PKM PKM PKM PKM PKM PKM PKM PKM PKM PKM PKM PKM PKM
                                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
        else if (MODE == 10) asm volatile(R8(LSHR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int instr_per_iter, float* d, unsigned long long* clk) {
    for (int wps : {1, 2, 3, 4, 8}) {                 // waves per SIMD: block of wps*4 waves, one block per CU
        const int iters = 50000 / wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<wps > 4 ? 512 : 256, wps > 4 ? 1024 : wps * 256>>>(d, clk, iters, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<wps > 4 ? 512 : 256, wps > 4 ? 1024 : wps * 256>>>(d, clk, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / (double)h[1] * 0.1;       // memtime ticks per 100 MHz tick
        const double ns_per_simd_instr = ms * 1e6 / ((double)iters * instr_per_iter * wps);
        printf("%-14s waves/SIMD=%d wall=%7.2f ms  clock(memtime/realtime)=%.2f GHz  ns per SIMD-instr=%.3f  -> cycles at that clock=%.2f\n", name, wps, ms, ghz,
               ns_per_simd_instr, ns_per_simd_instr * ghz);
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    run<0>("v_fma_f32", 64, d, clk);
    run<1>("v_pk_fma_f32", 64, d, clk);
    run<9>("v_pk_mul_f32", 64, d, clk);
    run<2>("v_exp_f32", 64, d, clk);
    run<3>("v_mul_f32", 64, d, clk);
    run<4>("scan-mix", 320, d, clk);
    run<5>("v_mul_lo_u32", 64, d, clk);
    run<6>("v_mad_u32_u24", 64, d, clk);
    run<7>("v_xor_b32", 64, d, clk);
    run<10>("v_lshrrev_b32", 64, d, clk);
    run<8>("v_cndmask_b32", 64, d, clk);
    return 0;
}
