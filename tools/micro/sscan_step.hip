// Issue-rate study of the selective-scan forward step (per wave: NS states of one channel per lane, sequential in time).
// Pure register arithmetic, no memory in the loop: what does the 5-op recurrence (mul, exp2, mul, fma, fma per state) cost
// per state-step as a function of waves per SIMD and of the way it is written?
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize sscan_step.hip -o sscan_step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int VAR, int NS, int WPS>
__global__ __launch_bounds__(WPS * 256) void k(float* out, unsigned long long* clk, int steps, float seed, float bq, float cq) {
    float A2[NS], h[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) { A2[j] = -0.01f * (j + 1) - 1e-3f * threadIdx.x; h[j] = 0.f; }
    float dlr[32], dur[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) { dlr[t] = seed + 0.01f * t + 1e-4f * threadIdx.x; dur[t] = dlr[t] * 0.5f; }
    float B[NS], C[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) { B[j] = bq + 0.1f * j; C[j] = cq - 0.1f * j; }
    float ysum = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < steps; c += 32) {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            float dl = dlr[t], du = dur[t];
            asm volatile("" : "+v"(dl), "+v"(du));          // opaque: nothing of the step may be hoisted out of the time loop
            if (VAR == 0) {                       // scalar ops, straightforward
                float y = 0.f;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const float dA = __builtin_amdgcn_exp2f(dl * A2[j]);
                    h[j] = __builtin_fmaf(dA, h[j], du * B[j]);
                    y = __builtin_fmaf(C[j], h[j], y);
                }
                ysum += y;
            } else if (VAR == 1) {                // packed pairs (the round-1 kernel's form)
                f2 y = {0.f, 0.f};
                const f2 dl2 = {dl, dl}, du2 = {du, du};
#pragma unroll
                for (int j = 0; j < NS; j += 2) {
                    const f2 arg = dl2 * f2{A2[j], A2[j + 1]};
                    f2 dA = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                    f2 hh = {h[j], h[j + 1]};
                    hh = __builtin_elementwise_fma(dA, hh, du2 * f2{B[j], B[j + 1]});
                    h[j] = hh.x; h[j + 1] = hh.y;
                    y = __builtin_elementwise_fma(f2{C[j], C[j + 1]}, hh, y);
                }
                ysum += y.x + y.y;
            } else if (VAR == 2) {                // scalar, two y chains, exps first
                float dA[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) dA[j] = __builtin_amdgcn_exp2f(dl * A2[j]);
                float y0 = 0.f, y1 = 0.f;
#pragma unroll
                for (int j = 0; j < NS; j += 2) {
                    h[j] = __builtin_fmaf(dA[j], h[j], du * B[j]);
                    h[j + 1] = __builtin_fmaf(dA[j + 1], h[j + 1], du * B[j + 1]);
                    y0 = __builtin_fmaf(C[j], h[j], y0);
                    y1 = __builtin_fmaf(C[j + 1], h[j + 1], y1);
                }
                ysum += y0 + y1;
            }
        }
        // perturb so that nothing is loop-invariant
#pragma unroll
        for (int t = 0; t < 32; t += 8) dlr[t] += 1e-6f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = ysum;
#pragma unroll
    for (int j = 0; j < NS; ++j) r += h[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int VAR, int NS, int wps>
void run1(const char* name, float* d, unsigned long long* clk) {
    {
        const int steps = 32 * 4000 / wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<VAR, NS, wps>), dim3(256), dim3(wps * 256), 0, 0, d, clk, steps, 0.3f, 0.7f, 0.2f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<VAR, NS, wps>), dim3(256), dim3(wps * 256), 0, 0, d, clk, steps, 0.3f, 0.7f, 0.2f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / (double)h[1] * 0.1;
        const double ns = ms * 1e6 / ((double)steps * NS * wps);       // per SIMD per (wave, state, step)
        printf("%-28s NS=%d waves/SIMD=%d wall=%7.2f ms clk=%.2f GHz  cycles per state-step per SIMD=%.2f\n", name, NS, wps, ms, ghz, ns * ghz);
    }
}

template <int VAR, int NS>
void run(const char* name, float* d, unsigned long long* clk) {
    run1<VAR, NS, 1>(name, d, clk); run1<VAR, NS, 2>(name, d, clk); run1<VAR, NS, 3>(name, d, clk); run1<VAR, NS, 4>(name, d, clk);
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    run<0, 8>("scalar", d, clk);
    run<1, 8>("packed pairs", d, clk);
    run<2, 8>("scalar exps-first 2 chains", d, clk);
    run<0, 4>("scalar", d, clk);
    run<2, 4>("scalar exps-first 2 chains", d, clk);
    run<0, 16>("scalar", d, clk);
    return 0;
}
