// fp32 GEMM on the f32-input matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, 64 FLOP/clk/SIMD),
// with the tails of the fc / efc-E layers fused into the epilogue.
//
//     C[b][m][n] = epi( sum_k A[b](m, k) * B[b](n, k) )            b = ensemble member (blockIdx.z), strides per operand
//
// Each operand is read either as a [rows][K] matrix (k contiguous: activations x, weights W[out][in]) or as a [K][rows]
// matrix (row index contiguous: the TRANSPOSED use of an activation matrix in a weight gradient, or a weight matrix in an
// input gradient), so one kernel serves
//     forward   y[M, out]  = x[M, in] W[out, in]^T          A = x  [rows][K],  B = W  [rows][K]   (+ bias, + ELU)
//     dgrad     dx[M, in]  = dy[M, out] W[out, in]          A = dy [rows][K],  B = W  [K][rows]
//     wgrad     dW[out,in] = dy[M, out]^T x[M, in]          A = dy [K][rows],  B = x  [K][rows]   (K = M = 66 752: split over blocks)
// (reference: torch.nn.Linear in models/rnn_base.py:101-105 and EnsembleLinear.forward, models/ensemble_linear_model.py:36-49).
//
// Tiling: 128 x 128 block tile, 4 waves as 2 x 2, wave tile 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers), K step 32.
// A 32x32x2 MFMA takes one A and one B value per lane: lane (i = l & 31, h = l >> 5) supplies A(i, k_h) and B(j = i, k_h).  A
// group of four MFMAs covers 8 consecutive k with the assignment k = 8 g + 4 h + e (e = MFMA in the group) - the sum over k is
// order-free per output, and with it a lane's four A values are CONTIGUOUS in k: for a [rows][K] operand the LDS image keeps
// k contiguous (rows padded to 36 floats) and one ds_read_b128 feeds four MFMAs; for a [K][rows] operand the image is
// k-major and the four values are four ds_read_b32 of 32 consecutive floats.  At 64 cycles per MFMA either way is far
// below the LDS rate - no transposes anywhere, global loads are float4 along the operand's contiguous axis in both forms.
// Global -> LDS goes through registers one K step ahead (the loads of step s + 1 are in flight during the 64 MFMAs of step
// s), LDS is double-buffered, one barrier per step.  72 KB of LDS per block: two blocks (two waves per SIMD) per CU.
// Split-K (wgrad): blockIdx.y = K slice, partial tiles go to a slab [slice][M][N] summed in fixed order by a second kernel
// (deterministic, no atomics).  Block ids are XCD-aware: the n-tiles of one m-tile share an L2.
#include "resel_common.h"

namespace {
using namespace resel;

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BN = 128, BK = 32;        // block tile BM x 128, BM = 128 or 64 (template): see pick_bm
constexpr int LDK = BK + 4;           // [row][k] image: 36 floats per row (16-byte aligned rows, conflict-free ds_read_b128)
template <int ROWS> constexpr int ldr() { return ROWS + 4; }     // [k][row] image: ROWS + 4 floats per k

struct GemmParams {
    const float *A, *B, *bias;
    float* C;
    int64_t lda, ldb, ldc, sA, sB, sC, sBias;    // leading dimensions (floats) and per-batch strides
    int M, N, K, kslice;                          // kslice: K range per blockIdx.y (multiple of BK), = K without split-K
    int act;                                      // 0 none, 1 ELU
    int mt, nt;                                   // tile counts
};

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : fast_exp(x) - 1.f; }

// stage one 128 x 32 operand tile: registers <- global (float4 along the contiguous axis, zero beyond the edges)
template <bool KC, int ROWS>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int rows, int K, int r0, int k0, int kend, int tid,
                                          float4 (&r)[ROWS / 32]) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {                                  // P[row][k]: thread -> (row = tid / 8 + 32 i, k = 4 (tid % 8))
            const int row = r0 + (tid >> 3) + 32 * i, k = k0 + 4 * (tid & 7);
            if (row < rows && k < kend) r[i] = ld4(P + (int64_t)row * ld + k);
        } else {                                   // P[k][row]: ROWS / 4 threads per k
            constexpr int TPK = ROWS / 4, KPP = 256 / TPK;
            const int k = k0 + tid / TPK + KPP * i, row = r0 + 4 * (tid % TPK);
            if (k < kend && row < rows) r[i] = ld4(P + (int64_t)k * ld + row);
        }
    }
}
template <bool KC, int ROWS>
__device__ __forceinline__ void tile_store(float* __restrict__ S, int tid, const float4 (&r)[ROWS / 32]) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        if (KC) st4(S + ((tid >> 3) + 32 * i) * LDK + 4 * (tid & 7), r[i]);
        else {
            constexpr int TPK = ROWS / 4, KPP = 256 / TPK;
            st4(S + (tid / TPK + KPP * i) * ldr<ROWS>() + 4 * (tid % TPK), r[i]);
        }
    }
}
// the four operand values of lane (i, h) for k-group g of a 32-row sub-tile starting at row rb
template <bool KC, int ROWS>
__device__ __forceinline__ float4 frag(const float* __restrict__ S, int rb, int g, int i, int h) {
    if (KC) return ld4(S + (rb + i) * LDK + 8 * g + 4 * h);
    constexpr int LDR = ldr<ROWS>();
    const float* q = S + (8 * g + 4 * h) * LDR + rb + i;
    return make_float4(q[0], q[LDR], q[2 * LDR], q[3 * LDR]);
}

template <bool AKC, bool BKC, int BM>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
    constexpr int TM = BM / 64;                    // MFMA tiles per wave along m (wave tile 32 TM x 64)
    constexpr int ASZ = AKC ? BM * LDK : BK * ldr<BM>(), BSZ = BKC ? BN * LDK : BK * ldr<BN>();
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];
    // XCD-aware tile id: ids congruent mod 8 share an XCD; give each XCD whole m-tiles (their n-tiles reuse the A rows in L2)
    const int ntile = p.mt * p.nt;
    int bid = blockIdx.x;
    {
        const int q = ntile / 8, r = ntile % 8, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = bid / p.nt, tn = bid % p.nt;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.kslice, kend = min(p.K, kbeg + p.kslice);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * (32 * TM), wn = (w & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const float* A = p.A + (int64_t)blockIdx.z * p.sA;
    const float* B = p.B + (int64_t)blockIdx.z * p.sB;

    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    float4 ra[BM / 32], rb[BN / 32];
    tile_load<AKC, BM>(A, p.lda, p.M, p.K, m0, kbeg, kend, tid, ra);
    tile_load<BKC, BN>(B, p.ldb, p.N, p.K, n0, kbeg, kend, tid, rb);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        float* As = lds + buf * (ASZ + BSZ);
        float* Bs = As + ASZ;
        tile_store<AKC, BM>(As, tid, ra);
        tile_store<BKC, BN>(Bs, tid, rb);
        __syncthreads();
        if (k0 + BK < kend) {                       // next step's operands: in flight during this step's MFMAs
            tile_load<AKC, BM>(A, p.lda, p.M, p.K, m0, k0 + BK, kend, tid, ra);
            tile_load<BKC, BN>(B, p.ldb, p.N, p.K, n0, k0 + BK, kend, tid, rb);
        }
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 fa[TM], fb[2];
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[t] = frag<AKC, BM>(As, wm + 32 * t, g, li, lh);
#pragma unroll
            for (int t = 0; t < 2; ++t) fb[t] = frag<BKC, BN>(Bs, wn + 32 * t, g, li, lh);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
        buf ^= 1;                                   // the other buffer was last read before the barrier above
    }
    // epilogue: D layout col = lane & 31 (n), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (m)
    float* C = p.C + (int64_t)blockIdx.z * p.sC + (int64_t)blockIdx.y * gridDim.z * p.M * p.ldc;   // split-K: slab [slice][batch][M][N]
    const float* bias = p.bias ? p.bias + (int64_t)blockIdx.z * p.sBias : nullptr;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn + 32 * b + li;
        if (n >= p.N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < p.M) {
                    float v = acc[a][b][e] + bv;
                    if (p.act == 1) v = elu1(v);
                    C[(int64_t)m * p.ldc + n] = v;
                }
            }
    }
}

// out[i] = sum_s slab[s][i]  (fixed order; float4 per thread)
__global__ void splitk_sum_kernel(const float* __restrict__ slab, int nslice, int64_t n, float* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 acc = ld4(slab + i);
    for (int s = 1; s < nslice; ++s) {
        const float4 v = ld4(slab + (int64_t)s * n + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    st4(out + i, acc);
}

// 128-row tiles unless they would leave the 512 block slots (2 per CU) badly quantised: fewer than ~6 rounds of tiles
inline int pick_bm(int M, int N, int batch) {
    const long tiles = (long)((M + 127) / 128) * ((N + BN - 1) / BN) * batch;
    return tiles >= 3072 ? 128 : 64;
}
inline int pick_slices(int M, int N, int K, int batch) {
    const int BM = pick_bm(M, N, batch);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
    if (tiles >= 256 || K < 2048) return 1;
    int s = (512 + tiles - 1) / tiles;              // ~two blocks per CU
    const int maxs = K / 256 > 0 ? K / 256 : 1;     // at least 8 K steps per slice
    return s < maxs ? s : maxs;
}

}  // namespace

extern "C" size_t resel_gemm_f32_workspace_bytes(int M, int N, int K, int batch) {
    const int s = pick_slices(M, N, K, batch);
    return s > 1 ? (size_t)s * batch * M * N * sizeof(float) : 0;
}

extern "C" int resel_gemm_f32(const float* A, int64_t lda, int64_t strideA, int a_kcontig,
                              const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                              const float* bias, int64_t strideBias, int act,
                              float* C, int64_t ldc, int64_t strideC, void* workspace,
                              int M, int N, int K, int batch, resel_stream_t stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || act < 0 || act > 1) return RESEL_EINVAL;
    if (lda % 4 || ldb % 4 || strideA % 4 || strideB % 4 || !aligned16(A) || !aligned16(B)) return RESEL_EINVAL;
    // float4 loads run along the contiguous axis: its extent must be a multiple of 4 (K for [rows][K] operands, rows otherwise)
    if ((a_kcontig ? K : M) % 4 || (b_kcontig ? K : N) % 4) return RESEL_EINVAL;
    const int slices = pick_slices(M, N, K, batch);
    // split-K output: dense [batch][M][N] (the slabs are summed as flat arrays)
    if (slices > 1 && (bias || act || !workspace || ldc != N || (batch > 1 && strideC != (int64_t)M * N) || (N % 4) || !aligned16(C) ||
                       !aligned16(workspace))) return RESEL_EINVAL;
    const int BM = pick_bm(M, N, batch);
    GemmParams p{A, B, bias, slices > 1 ? (float*)workspace : C, lda, ldb, slices > 1 ? (int64_t)N : ldc, strideA, strideB,
                 slices > 1 ? (int64_t)M * N : strideC, strideBias, M, N, K, 0, act, (M + BM - 1) / BM, (N + BN - 1) / BN};
    p.kslice = slices > 1 ? ((K + slices - 1) / slices + BK - 1) / BK * BK : (K + BK - 1) / BK * BK;
    const int nsl = (K + p.kslice - 1) / p.kslice;
    dim3 grid(p.mt * p.nt, nsl, batch);
    hipStream_t s = (hipStream_t)stream;
#define RESEL_GEMM_LAUNCH(AK, BK_) \
    do { if (BM == 128) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, 128>), grid, dim3(256), 0, s, p); \
         else hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, 64>), grid, dim3(256), 0, s, p); } while (0)
    if (a_kcontig && b_kcontig) RESEL_GEMM_LAUNCH(true, true);
    else if (a_kcontig) RESEL_GEMM_LAUNCH(true, false);
    else if (b_kcontig) RESEL_GEMM_LAUNCH(false, true);
    else RESEL_GEMM_LAUNCH(false, false);
#undef RESEL_GEMM_LAUNCH
    if (slices > 1) {
        const int64_t n = (int64_t)batch * M * N;
        hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const float*)workspace, nsl, n, C);
    }
    return launch_status();
}
