// What does the L2 -> CU path deliver for the GEMM's tile-load pattern?  Every block walks 256-row tiles of a [rows][K] fp32 matrix K step
// by K step; W = bytes of one row fetched per step (128 = the GEMM's BK of 32 floats; 256 / 512 = what a wider step would fetch), SH = how
// many blocks read the same tile at the same time (16 = the n-tiles of one m-tile; 1 = nobody shares: HBM).  IN_FLIGHT float4 per thread.
// build: hipcc --offload-arch=gfx950 -O3 l2_tile_bw.hip -o bin/l2_tile_bw      run: bin/l2_tile_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W, int NF>   // NF = a multiple of W / 16 (the instructions that cover the 256 rows of one step)
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, int64_t ld, int mtiles, int K, int SH, float* out) {
    constexpr int LPR = W / 16;                      // lanes per row
    constexpr int RPI = 256 / LPR;                   // rows per instruction of the block
    const int tid = threadIdx.x;
    const int grp = blockIdx.x / SH;                 // blocks of a group share their tiles
    const int ngrp = gridDim.x / SH;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = grp; t < mtiles; t += ngrp) {
        const float* base = A + (int64_t)t * 256 * ld + (tid / LPR) * ld + (tid % LPR) * 4;
        for (int k0 = 0; k0 < K; k0 += W / 4 * NF / (256 / RPI)) {
            float4 v[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) {           // NF instructions: rows (tid / LPR) + RPI * (i % (256 / RPI)), step k0 + (W / 4) * (i / (256 / RPI))
                const int ri = i % (256 / RPI), ki = i / (256 / RPI);
                v[i] = *reinterpret_cast<const float4*>(base + (int64_t)ri * RPI * ld + k0 + ki * (W / 4));
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
template <int W, int NF>
void run(const float* A, int64_t ld, int M, int K, int SH, float* out, const char* what) {
    const int mt = M / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<W, NF>), dim3(256), dim3(256), 0, 0, A, ld, mt, K, SH, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<W, NF>), dim3(256), dim3(256), 0, 0, A, ld, mt, K, SH, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)mt * 256 * K * 4 * SH;      // bytes delivered to CUs per launch
    printf("%-44s W=%3d B/row/step, %2d float4 in flight/thread, share %2d: %7.1f us  %6.2f TB/s to the CUs = %5.1f GB/s per CU\n", what, W, NF, SH,
           ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, bytes / (ms / reps * 1e-3) / 256 / 1e9);
}
int main() {
    const int M = 66816, K = 384;                    // 261 tiles of 256 rows
    float *A, *out;
    hipMalloc(&A, (size_t)M * 2048 * 4); hipMalloc(&out, 4096);
    hipMemset(A, 0, (size_t)M * 2048 * 4);
    for (int SH : {16, 1}) {
        run<128, 8>(A, K, M, K, SH, out, "K=384 rows of 1536 B");
        run<128, 16>(A, K, M, K, SH, out, "K=384 rows of 1536 B");
        run<256, 16>(A, K, M, K, SH, out, "K=384 rows of 1536 B");
        run<256, 32>(A, K, M, K, SH, out, "K=384 rows of 1536 B");
        run<512, 32>(A, K, M, K, SH, out, "K=384 rows of 1536 B");
        run<128, 16>(A, 2048, M, 2048, SH, out, "K=2048 rows of 8192 B");
        run<512, 32>(A, 2048, M, 2048, SH, out, "K=2048 rows of 8192 B");
    }
    return 0;
}
