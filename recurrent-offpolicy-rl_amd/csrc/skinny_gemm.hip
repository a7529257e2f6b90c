// C = W^T N over a very long reduction dimension: the weight-gradient GEMMs of the narrow Mamba projections
//   d(dt_proj.weight)[Di, R]    = ddelta[M, Di]^T  x_dbl[M, :R]         (R = 16)
//   d(x_proj.weight)[R+2N, Di]  = dx_dbl[M, R+2N]^T xc[M, Di]           (R + 2N = 80)
// with M = rows * T' = 66 752 tokens at config 2 (reference: autograd of the three F.linear calls in
// mamba_ssm/ops/selective_scan_interface_new.py:261-335).  The outputs are tiny (8 K / 41 K floats), the reduction is 66 752
// long: the library's kernels for these shapes reach 7 - 70 TFLOP/s (149 / 77 us); the operation is bound by reading the wide
// operand once (137 MB ~ 25 us at HBM speed).
//
// Layout trick: v_mfma_f32_32x32x2_f32 wants lane l to hold operand element [l % 32][k = l / 32].  Both operands are row-major
// [k][column], so a half-wave reads 32 consecutive columns of row k - and if every lane reads a float4 (4 consecutive columns)
// the four components are the A operands of FOUR output tiles whose rows are the columns 4 i + e (a permutation of the 128
// columns the wave owns).  One 16-byte load per lane feeds four MFMAs; no LDS, no transposes.  The reduction dimension is
// split over the grid; partial tiles go to a workspace in the FINAL layout and are summed in a fixed order (colsum_kernel),
// so the result is bitwise reproducible.
#include "resel_common.h"

namespace {
using namespace resel;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C row of accumulator register r of lane l in a 32x32 tile (column = l % 32)
__device__ __forceinline__ int acc_row(int r, int half) { return (r >> 2) * 8 + half * 4 + (r & 3); }

// grid (ksplit, ceil(Wd / 512)); 256 threads = 4 waves x 128 wide columns.  NT = number of 32-column tiles of the narrow operand.
template <int NT>
__global__ __launch_bounds__(256) void atb_kernel(const float* __restrict__ Wm, int64_t ldw, int Wd, const float* __restrict__ Nm,
                                                  int64_t ldn, int Nd, float* __restrict__ part, int64_t K, int rows_per,
                                                  int transposed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wb = blockIdx.y * 512 + wave * 128;
    if (wb >= Wd) return;
    const int64_t k0 = (int64_t)blockIdx.x * rows_per;
    const int64_t k1 = k0 + rows_per < K ? k0 + rows_per : K;
    f32x16 acc[4][NT];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][t][r] = 0.f;
    const int wc = wb + 4 * col;
    const bool w_ok = wc < Wd;                                   // Wd % 4 == 0: a float4 is inside or outside as a whole
    bool n_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) n_ok[t] = 32 * t + col < Nd;
    constexpr int UN = NT == 1 ? 8 : 4;                          // k-pairs in flight per iteration
    for (int64_t k = k0; k < k1; k += 2 * UN) {
        float4 a[UN];
        float b[UN][NT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t kr = k + 2 * u + half;
            const bool ok = kr < k1;
            a[u] = (ok && w_ok) ? ld4(Wm + kr * ldw + wc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < NT; ++t) b[u][t] = (ok && n_ok[t]) ? Nm[kr * ldn + 32 * t + col] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u][t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u][t], acc[1][t], 0, 0, 0);
                acc[2][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u][t], acc[2][t], 0, 0, 0);
                acc[3][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u][t], acc[3][t], 0, 0, 0);
            }
    }
    // partial slab of this k-split in the final layout: C[w][n] (ld Nd) or, transposed, C[n][w] (ld Wd)
    float* dst = part + (int64_t)blockIdx.x * Wd * Nd;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 32 * t + col;
            if (n >= Nd) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = wb + 4 * acc_row(r, half) + e;
                if (w < Wd) dst[transposed ? (int64_t)n * Wd + w : (int64_t)w * Nd + n] = acc[e][t][r];
            }
        }
}

inline int pick_ksplit(int64_t K, int Wd, int Nd) {
    // enough workgroups to pull HBM bandwidth, partial slabs bounded to ~16 MB
    int64_t ks = (int64_t)(16 << 20) / ((int64_t)Wd * Nd * 4);
    if (ks > 512) ks = 512;
    if (ks < 32) ks = 32;
    if (ks > (K + 1) / 2) ks = (K + 1) / 2;
    return (int)(ks < 1 ? 1 : ks);
}

}  // namespace

extern "C" {

size_t resel_atb_workspace_bytes(int64_t K, int Wd, int Nd) {
    return (size_t)pick_ksplit(K, Wd, Nd) * Wd * Nd * sizeof(float);
}

int resel_atb(const float* wide, int64_t ldw, int Wd, const float* narrow, int64_t ldn, int Nd, float* out, int transposed,
              void* workspace, int64_t K, resel_stream_t stream) {
    if (!wide || !narrow || !out || !workspace || K <= 0 || Wd <= 0 || Nd <= 0) return RESEL_EINVAL;
    if (Nd > 96 || (Wd & 3) || (ldw & 3) || !aligned16(wide)) return RESEL_EINVAL;
    const int ks = pick_ksplit(K, Wd, Nd);
    int rows_per = (int)((K + ks - 1) / ks);
    rows_per += rows_per & 1;                                    // k-pairs never straddle two splits
    const int nsplit = (int)((K + rows_per - 1) / rows_per);
    const dim3 grid(nsplit, (Wd + 511) / 512), block(256);
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    const int NT = (Nd + 31) / 32;
    switch (NT) {
        case 1: hipLaunchKernelGGL(atb_kernel<1>, grid, block, 0, s, wide, ldw, Wd, narrow, ldn, Nd, part, K, rows_per, transposed); break;
        case 2: hipLaunchKernelGGL(atb_kernel<2>, grid, block, 0, s, wide, ldw, Wd, narrow, ldn, Nd, part, K, rows_per, transposed); break;
        default: hipLaunchKernelGGL(atb_kernel<3>, grid, block, 0, s, wide, ldw, Wd, narrow, ldn, Nd, part, K, rows_per, transposed); break;
    }
    launch_colsum(part, (int64_t)Wd * Nd, nsplit, Wd * Nd, out, s);
    return launch_status();
}

}  // extern "C"
