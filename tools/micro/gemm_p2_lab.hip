// Verdict r05 item 1 (c): "two co-resident 4-wave workgroups per CU on 128 x 128 tiles, so one's C-store epilogue overlaps the other's K loop".
// A stand-alone mode-2 GEMM (fp16 planes of the scaled operands, three plane products on v_mfma_f32_32x32x16_f16, one fp32 accumulator - the
// product of csrc/gemm_bf3.hip) with a 128 x 128 block tile, 256 threads (2 x 2 waves of 64 x 64, every wave does everything), two LDS stages of
// 32 KB: 64 KB per workgroup, ~150 VGPRs - TWO workgroups per CU.  One tile per workgroup (no persistence): the two residents of a CU drift apart
// by themselves.  C = act(A [M][K] . B [N][K]^T + bias), K a multiple of 32.  Timed against resel_gemm_f32x of the shipped library (dlopen) on the
// same operands; checked against fp64 on sampled rows.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/micro/gemm_p2_lab.hip -o tools/micro/bin/gemm_p2_lab -ldl
//   run:   tools/micro/bin/gemm_p2_lab M N K [act=0|1] [path to libresel_hip.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, ROWB = 64;
constexpr int PL = BM * ROWB;                       // one plane of one operand tile: 8 KB
constexpr int STAGE = 4 * PL;                       // a1 a2 b1 b2

__device__ __forceinline__ int plane_off(int row, int c) {      // byte offset of (row, 16-byte chunk c = k / 8) inside a plane (the library's swizzle)
    const int q = row >> 2;
    return ((row ^ (q & 1)) << 6) + ((c ^ (q & 3)) << 4);
}
template <bool WIDE>
__device__ __forceinline__ void split_pair(float x0, float x1, f32x2_t sc, uint32_t& p1, uint32_t& p2) {
    const f32x2_t xs = {x0 * sc.x, x1 * sc.x};
    const f16x2_t h = __builtin_convertvector(xs, f16x2_t);
    p1 = __builtin_bit_cast(uint32_t, h);
    f32x2_t r;
    if (WIDE) r = f32x2_t{__builtin_fmaf((float)h.x, -2048.f, x0 * sc.y), __builtin_fmaf((float)h.y, -2048.f, x1 * sc.y)};
    else r = f32x2_t{xs.x - (float)h.x, xs.y - (float)h.y};
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));
}
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f; }

struct P { const float *A, *B, *bias; float* C; int64_t lda, ldb, ldc; int M, N, K, act; float sa, sb; };

template <int OCC, int V = 0>       // V = 1: loads two K steps ahead (two register sets) + C through LDS as 16-byte full-line stores
__global__ __launch_bounds__(256, OCC) void gemm_p2_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64, li = lane & 31, lh = lane >> 5;
    // tile of this workgroup: ids congruent mod 8 (one XCD) own a contiguous range of tiles, N fastest (neighbours share their A rows in that L2)
    const int nt = (p.N + BN - 1) / BN, ntile = ((p.M + BM - 1) / BM) * nt;
    const int q = ntile / 8, r = ntile % 8, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    if (j >= q + (x < r ? 1 : 0)) return;
    const int m0 = (bid / nt) * BM, n0 = (bid % nt) * BN;
    const f32x2_t scA = {p.sa, 2048.f * p.sa}, scB = {p.sb, 2048.f * p.sb};
    // loads: thread = (row tid >> 3 (+ 32 i), float4 k4 = tid & 7) of both operand tiles
    const int lr = tid >> 3, k4 = tid & 7;
    const float* ag[4];
    const float* bg[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = lr + 32 * i;
        ag[i] = p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + 4 * k4;
        bg[i] = p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + 4 * k4;
        loff[i] = plane_off(row, k4 >> 1) + 8 * (k4 & 1);
    }
    float4 RA[2][4], RB[2][4];
    auto gload_set = [&](float4 (&ra)[4], float4 (&rb)[4], int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = *reinterpret_cast<const float4*>(ag[i] + k0); rb[i] = *reinterpret_cast<const float4*>(bg[i] + k0); }
    };
    auto stage_store_set = [&](const float4 (&ra)[4], const float4 (&rb)[4], char* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 a1, a2, b1, b2;
            split_pair<true>(ra[i].x, ra[i].y, scA, a1.x, a2.x); split_pair<true>(ra[i].z, ra[i].w, scA, a1.y, a2.y);
            split_pair<false>(rb[i].x, rb[i].y, scB, b1.x, b2.x); split_pair<false>(rb[i].z, rb[i].w, scB, b1.y, b2.y);
            *reinterpret_cast<uint2*>(st + loff[i]) = a1;
            *reinterpret_cast<uint2*>(st + PL + loff[i]) = a2;
            *reinterpret_cast<uint2*>(st + 2 * PL + loff[i]) = b1;
            *reinterpret_cast<uint2*>(st + 3 * PL + loff[i]) = b2;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const f16x8 k11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f,
                       (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f, (_Float16)0.00048828125f};
    auto gload = [&](int k0) { gload_set(RA[0], RB[0], k0); };
    auto stage_store = [&](char* st) { stage_store_set(RA[0], RB[0], st); };
    const int nk = p.K / BK;
    gload(0);
    stage_store(lds);
    if (V == 1) {
        if (nk > 1) gload_set(RA[1], RB[1], BK);
        if (nk > 2) gload_set(RA[0], RB[0], 2 * BK);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        char* st = lds + (ks & 1) * STAGE;
        if (V == 0 && ks + 1 < nk) gload((ks + 1) * BK);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 fa[2][2], fb[2][2], fs[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oa = (wm + 32 * t) * ROWB + plane_off(li, 2 * s + lh), ob = (wn + 32 * t) * ROWB + plane_off(li, 2 * s + lh);
                fa[0][t] = *reinterpret_cast<const f16x8*>(st + oa);
                fa[1][t] = *reinterpret_cast<const f16x8*>(st + PL + oa);
                fb[0][t] = *reinterpret_cast<const f16x8*>(st + 2 * PL + ob);
                fb[1][t] = *reinterpret_cast<const f16x8*>(st + 3 * PL + ob);
                fs[t] = fb[0][t] * k11;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][a], fs[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int qq = 1; qq >= 0; --qq)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][a], fb[qq][b], acc[a][b], 0, 0, 0);
        }
        if (V == 0) {
            if (ks + 1 < nk) stage_store(lds + ((ks + 1) & 1) * STAGE);
        } else if (ks + 1 < nk) {                  // the set of step ks + 1 (loaded two steps ago) goes to LDS, then it is reloaded for step ks + 3
            if ((ks + 1) & 1) { stage_store_set(RA[1], RB[1], lds + STAGE); if (ks + 3 < nk) gload_set(RA[1], RB[1], (ks + 3) * BK); }
            else { stage_store_set(RA[0], RB[0], lds); if (ks + 3 < nk) gload_set(RA[0], RB[0], (ks + 3) * BK); }
        }
        __syncthreads();
    }
    const float unscale = (1.f / p.sa) * (1.f / p.sb);
    if (V == 1 && m0 + BM <= p.M && n0 + BN <= p.N) {
        // whole tiles: each wave turns its 64 x 64 tile through 8.7 KB of LDS (the stages are free now: the loop ended with a barrier), 32 rows at a
        // time, rows of 68 words, and stores 16 bytes per lane - four whole 256-byte row segments per instruction
        float* sc = reinterpret_cast<float*>(lds) + w * (32 * 68);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float bv = p.bias ? p.bias[n0 + wn + 32 * b + li] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[a][b][e] * unscale + bv;
                    if (p.act == 1) v = elu1(v);
                    sc[((e & 3) + 8 * (e >> 2) + 4 * lh) * 68 + 32 * b + li] = v;
                }
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int idx = g * 64 + lane, row = idx >> 4, c4 = (idx & 15) * 4;
                const float4 t = *reinterpret_cast<const float4*>(sc + row * 68 + c4);
                typedef float f4v __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(f4v{t.x, t.y, t.z, t.w}, reinterpret_cast<f4v*>(p.C + (int64_t)(m0 + wm + 32 * a + row) * p.ldc + n0 + wn + c4));
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn + 32 * b + li;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int mb = m0 + wm + 32 * a + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                float v = acc[a][b][e] * unscale + bv;
                if (p.act == 1) v = elu1(v);
                if (m < p.M) __builtin_nontemporal_store(v, p.C + (int64_t)m * p.ldc + n);
            }
        }
    }
}

static float f16_scale(float amax) {
    uint32_t u; memcpy(&u, &amax, 4);
    int f = (u >> 23) & 0xff, e = 268 - f; e = e < 1 ? 1 : (e > 240 ? 240 : e);
    uint32_t o = (uint32_t)e << 23; float s; memcpy(&s, &o, 4); return s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 66752, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 384, act = argc > 4 ? atoi(argv[4]) : 0;
    const char* libp = argc > 5 ? argv[5] : "recurrent-offpolicy-rl_amd/offpolicy_rnn/hip/libresel_hip.so";
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb(N);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    float amA = 0, amB = 0;
    for (auto& v : hA) { v = rnd() + rnd() + rnd(); amA = fmaxf(amA, fabsf(v)); }
    for (auto& v : hB) { v = (rnd() + rnd()) / sqrtf((float)K); amB = fmaxf(amB, fabsf(v)); }
    for (auto& v : hb) v = rnd();
    float *A, *B, *bias, *C, *C2;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&bias, N * 4));
    CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&C2, (size_t)M * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    P p{A, B, bias, C, K, K, N, M, N, K, act, f16_scale(amA), f16_scale(amB)};
    const int ntile = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const dim3 grid((ntile + 7) / 8 * 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto fn, const char* what) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) fn();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", what, ms * 50.f, 2.0 * M * N * K / (ms * 50e-6) / 1e12);
    };
    CK(hipFuncSetAttribute((const void*)gemm_p2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    CK(hipFuncSetAttribute((const void*)gemm_p2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE + 40960));
    time_it([&]() { hipLaunchKernelGGL(gemm_p2_kernel<2>, grid, dim3(256), 2 * STAGE, 0, p); }, "128 x 128, TWO workgroups per CU (64 KB LDS)");
    CK(hipFuncSetAttribute((const void*)gemm_p2_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    time_it([&]() { hipLaunchKernelGGL((gemm_p2_kernel<2, 1>), grid, dim3(256), 2 * STAGE, 0, p); }, "  + loads two steps ahead, C as 16-byte full-line stores");
    // the same kernel with its LDS request padded past half of the CU's 160 KB: ONE workgroup per CU - what co-residency itself buys
    time_it([&]() { hipLaunchKernelGGL(gemm_p2_kernel<1>, grid, dim3(256), 2 * STAGE + 40960, 0, p); }, "128 x 128, ONE workgroup per CU (LDS padded)");
    CK(hipGetLastError());
    // check sampled rows against fp64
    std::vector<float> hC((size_t)N);
    double worst = 0, scale = 0;
    for (int s = 0; s < 24; ++s) {
        const int m = (int)(((uint64_t)s * 2654435761u) % M);
        CK(hipMemcpy(hC.data(), C + (size_t)m * N, N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            double acc = hb[n];
            for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
            if (act == 1) acc = acc > 0 ? acc : expm1(acc);
            worst = fmax(worst, fabs(acc - hC[n])); scale = fmax(scale, fabs(acc));
        }
    }
    printf("max error against fp64 on 24 sampled rows: %.3e of scale %.3e (%s)\n", worst, scale, worst <= 2e-6 * scale + 1e-6 ? "ok" : "BAD");
    // the shipped kernel on the same operands
    void* h = dlopen(libp, RTLD_NOW);
    if (!h) { printf("no library at %s (%s)\n", libp, dlerror()); return 0; }
    typedef int (*amax_t)(const float*, int64_t, int64_t, int, int, int, void*, unsigned, void*, void*);
    typedef size_t (*wsb_t)(int, int, int, int);
    typedef int (*gemm_t)(const float*, int64_t, int64_t, int, const float*, int64_t, int64_t, int, const float*, int64_t, int, float*, int64_t, int64_t, void*,
                          int, int, int, int, int, const float*, const float*, void*, unsigned, void*);
    amax_t amax = (amax_t)dlsym(h, "resel_amax"); wsb_t wsb = (wsb_t)dlsym(h, "resel_gemm_f32_workspace_bytes"); gemm_t gemm = (gemm_t)dlsym(h, "resel_gemm_f32x");
    void *hA_, *hB_, *ws; CK(hipMalloc(&hA_, 1024)); CK(hipMalloc(&hB_, 1024)); CK(hipMemset(hA_, 0, 1024)); CK(hipMemset(hB_, 0, 1024));
    size_t nb = wsb(M, N, K, 1); CK(hipMalloc(&ws, nb ? nb : 16));
    amax(A, K, 0, M, K, 1, hA_, 1u, nullptr, nullptr); amax(B, K, 0, N, K, 1, hB_, 1u, nullptr, nullptr);
    CK(hipDeviceSynchronize());
    time_it([&]() { int rc = gemm(A, K, 0, 1, B, K, 0, 1, bias, 0, act, C2, N, 0, ws, M, N, K, 1, 2, (const float*)hA_, (const float*)hB_, nullptr, 0u, nullptr); if (rc) { printf("rc %d\n", rc); exit(1); } },
            "shipped: 256 x 128 producer / consumer edition");
    std::vector<float> c1((size_t)N), c2((size_t)N);
    double dmax = 0;
    for (int s = 0; s < 8; ++s) {
        const int m = (int)(((uint64_t)s * 40503u) % M);
        CK(hipMemcpy(c1.data(), C + (size_t)m * N, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2 + (size_t)m * N, N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) dmax = fmax(dmax, fabs((double)c1[n] - c2[n]));
    }
    printf("max difference to the shipped kernel on 8 sampled rows: %.3e\n", dmax);
    return 0;
}
