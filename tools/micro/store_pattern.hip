// How fast does the chip accept the C-store pattern of the GEMM epilogue?  Persistent blocks (one per CU, NW waves) write a [M][N] fp32
// matrix in 256 x 128 block tiles with 16-byte stores per lane, in different per-instruction shapes:
//   mode 0: 8 rows x 128 B per wave-instruction (the shipped epilogue: 32 x 32 sub-tiles, rows ld apart)
//   mode 1: 4 rows x 256 B      mode 2: 2 rows x 512 B (a whole tile row per half-wave)      mode 3: 1 KB contiguous (fill-like, ignores tiles)
// Build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern ;  run: ./store_pattern [N] [waves per block] [nt]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ void k(float* C, int M, int N, int ntile_m, int ntile_n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const f4v v = {1.f, 2.f, 3.f, (float)lane};
    const int total = ntile_m * ntile_n;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int m0 = (t / ntile_n) * 256, n0 = (t % ntile_n) * 128;
        // the block tile = 256 x 128 floats = 8192 float4; a wave-instruction writes 64 float4
        for (int i = w; i < 128; i += nw) {              // 128 wave-instructions per tile
            int row, col;
            if (MODE == 0) {           // sub-tile (32 x 32): instruction i = (sub-tile st = i / 4, group g = i % 4): rows 8 g + lane / 8, 8 lanes x 16 B per row
                const int st = i >> 2, g = i & 3;
                row = (st >> 2) * 32 + 8 * g + (lane >> 3); col = (st & 3) * 32 + 4 * (lane & 7);
            } else if (MODE == 1) {    // 4 rows x 256 B
                const int rg = i >> 1, h = i & 1;        // 64 row groups of 4 rows, 2 halves of 64 columns
                row = 4 * rg + (lane >> 4); col = 64 * h + 4 * (lane & 15);
            } else if (MODE == 2) {    // 2 rows x 512 B
                row = 2 * i + (lane >> 5); col = 4 * (lane & 31);
            } else {                   // contiguous
                row = 0; col = 0;
            }
            float* q = MODE == 3 ? C + ((size_t)t * 128 + i) * 256 + 4 * lane : C + (size_t)(m0 + row) * N + n0 + col;
            if (MODE != 3 && m0 + row >= M) continue;
            if (NT) __builtin_nontemporal_store(v, (f4v*)q); else *(f4v*)q = v;
        }
    }
}
template <int MODE, bool NT>
void run(const char* name, float* C, int M, int N, int nw) {
    const int tm = (M + 255) / 256, tn = N / 128;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k<MODE, NT><<<256, nw * 64>>>(C, M, N, tm, tn);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<MODE, NT><<<256, nw * 64>>>(C, M, N, tm, tn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/block=%d nt=%d: %7.1f us  %.2f TB/s\n", name, nw, (int)NT, ms * 100.0, (double)M * N * 4 / (ms * 1e-4) / 1e12);
}
int main(int argc, char** argv) {
    const int M = 66752, N = argc > 1 ? atoi(argv[1]) : 2048;
    float* C; hipMalloc(&C, (size_t)(M + 256) * N * 4);
    for (int nw : {4, 8, 16}) {
        run<0, true>("8 rows x 128 B (shipped)", C, M, N, nw);
        run<0, false>("8 rows x 128 B (shipped)", C, M, N, nw);
        run<1, true>("4 rows x 256 B", C, M, N, nw);
        run<2, true>("2 rows x 512 B", C, M, N, nw);
        run<3, true>("1 KB contiguous", C, M, N, nw);
        run<3, false>("1 KB contiguous", C, M, N, nw);
    }
    return 0;
}
