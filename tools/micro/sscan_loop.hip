// The scan phase of the selective-scan forward in isolation: 4 waves x 8 states per workgroup, B_t / C_t rows broadcast from
// LDS, y partials written to LDS, no global memory in the loop.  Variants of HOW the operands reach the VALU, timed at one
// and two workgroups per CU.  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize sscan_loop.hip -o sscan_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int TC = 32, NS = 8, NW = 4, N = NS * NW, NP = NS / 2;

__device__ __forceinline__ void coef(const float* p, f2 (&dst)[NP]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    dst[0] = f2{a.x, a.y}; dst[1] = f2{a.z, a.w}; dst[2] = f2{b.x, b.y}; dst[3] = f2{b.z, b.w};
}

// VAR 0: straight-line, operands read where used (compiler placement)
// VAR 1: explicit one-step-ahead register ping-pong, no scheduling fences
// VAR 2: ping-pong + sched_barrier fences (the round-1 kernel's form)
// VAR 3: no LDS operand reads at all (B / C constant in registers): the arithmetic alone
// VAR 4: as 1, y kept in registers and written once per 4 steps as ds_write_b128-shaped [t/4][lane][4]
// VAR 5: as 1 with two-steps-ahead prefetch (three register sets)
template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int chunks, float seed) {
    __shared__ __attribute__((aligned(16))) float s_B[TC][N];
    __shared__ __attribute__((aligned(16))) float s_C[TC][N];
    __shared__ __attribute__((aligned(16))) float s_y[NW][TC][64];
    __shared__ __attribute__((aligned(16))) float s_dl[TC][64];
    __shared__ __attribute__((aligned(16))) float s_du[TC][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < TC * N; i += 256) { (&s_B[0][0])[i] = 0.3f + 1e-3f * i; (&s_C[0][0])[i] = 0.2f - 1e-3f * i; }
    for (int i = tid; i < TC * 64; i += 256) { (&s_dl[0][0])[i] = seed + 1e-4f * i; (&s_du[0][0])[i] = 0.5f * seed + 1e-4f * i; }
    f2 A2[NP], h[NP];
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) { A2[kk] = f2{-0.01f * (2 * kk + 1) - 1e-3f * lane, -0.01f * (2 * kk + 2) - 1e-3f * lane}; h[kk] = f2{0.f, 0.f}; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < chunks; ++c) {
        float dlr[TC], dur[TC];
#pragma unroll
        for (int t = 0; t < TC; ++t) { dlr[t] = s_dl[t][lane]; dur[t] = s_du[t][lane]; }
        auto arith = [&](int t, const f2 (&Bq)[NP], const f2 (&Cq)[NP]) -> float {
            const f2 dl2 = {dlr[t], dlr[t]}, du2 = {dur[t], dur[t]};
            f2 y = {0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) {
                const f2 arg = dl2 * A2[kk];
                f2 dA = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                h[kk] = __builtin_elementwise_fma(dA, h[kk], du2 * Bq[kk]);
                y = __builtin_elementwise_fma(Cq[kk], h[kk], y);
            }
            return y.x + y.y;
        };
        if (VAR == 0) {
#pragma unroll
            for (int t = 0; t < TC; ++t) {
                f2 Bq[NP], Cq[NP];
                coef(&s_B[t][w * NS], Bq); coef(&s_C[t][w * NS], Cq);
                s_y[w][t][lane] = arith(t, Bq, Cq);
            }
        } else if (VAR == 1 || VAR == 2 || VAR == 4) {
            f2 B0[NP], C0[NP], B1[NP], C1[NP];
            coef(&s_B[0][w * NS], B0); coef(&s_C[0][w * NS], C0);
            float yk[4];
#pragma unroll
            for (int t = 0; t < TC; t += 2) {
                coef(&s_B[t + 1][w * NS], B1); coef(&s_C[t + 1][w * NS], C1);
                if (VAR == 2) __builtin_amdgcn_sched_barrier(0);
                const float ya = arith(t, B0, C0);
                if (VAR != 4) s_y[w][t][lane] = ya; else yk[t & 3] = ya;
                if (t + 2 < TC) { coef(&s_B[t + 2][w * NS], B0); coef(&s_C[t + 2][w * NS], C0); }
                if (VAR == 2) __builtin_amdgcn_sched_barrier(0);
                const float yb = arith(t + 1, B1, C1);
                if (VAR != 4) s_y[w][t + 1][lane] = yb; else yk[(t + 1) & 3] = yb;
                if (VAR == 4 && (t & 3) == 2) *reinterpret_cast<float4*>(&s_y[w][t & ~3][0] + lane * 4) = make_float4(yk[0], yk[1], yk[2], yk[3]);
            }
        } else if (VAR == 3) {
            f2 Bq[NP], Cq[NP];
            coef(&s_B[c & 31][w * NS], Bq); coef(&s_C[c & 31][w * NS], Cq);
#pragma unroll
            for (int t = 0; t < TC; ++t) s_y[w][t][lane] = arith(t, Bq, Cq);
        } else if (VAR == 5) {
            f2 Bs[3][NP], Cs[3][NP];
            coef(&s_B[0][w * NS], Bs[0]); coef(&s_C[0][w * NS], Cs[0]);
            coef(&s_B[1][w * NS], Bs[1]); coef(&s_C[1][w * NS], Cs[1]);
#pragma unroll
            for (int t = 0; t < TC; ++t) {
                if (t + 2 < TC) { coef(&s_B[t + 2][w * NS], Bs[(t + 2) % 3]); coef(&s_C[t + 2][w * NS], Cs[(t + 2) % 3]); }
                s_y[w][t][lane] = arith(t, Bs[t % 3], Cs[t % 3]);
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) r += h[kk].x + h[kk].y;
    out[blockIdx.x * 256 + tid] = r + s_y[w][lane & 31][lane];
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int VAR>
void run(const char* name, float* d, unsigned long long* clk) {
    for (int wgs : {256, 512}) {
        const int chunks = 400;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(256), 0, 0, d, clk, chunks, 0.3f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(256), 0, 0, d, clk, chunks, 0.3f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / (double)h[1] * 0.1;
        const double wave_steps_per_simd = (double)chunks * TC * (wgs / 256);          // one wave of each resident WG per SIMD
        printf("%-44s WGs/CU=%d wall=%7.2f ms clk=%.2f GHz  cycles per wave-step per SIMD=%.1f (= %.2f per state)\n", name, wgs / 256, ms, ghz,
               ms * 1e6 * ghz / wave_steps_per_simd, ms * 1e6 * ghz / wave_steps_per_simd / NS);
    }
}

int main() {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    run<3>("arithmetic only (B/C in registers)", d, clk);
    run<0>("straight-line, compiler placement", d, clk);
    run<1>("one step ahead, no fences", d, clk);
    run<2>("one step ahead, sched_barrier fences", d, clk);
    run<5>("two steps ahead, no fences", d, clk);
    run<4>("one step ahead, y as b128 per 4 steps", d, clk);
    return 0;
}
