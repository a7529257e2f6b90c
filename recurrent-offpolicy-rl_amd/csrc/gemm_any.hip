// The GEMM shapes the matrix-core editions do not take - so that NO product of the path is left to a vendor library:
//   * rows that are not 16-byte multiples (the 6-wide TD3 / discrete heads and their gradients, a rank-2 dt_proj, odd action counts),
//     reductions shorter than one matrix-instruction step, outputs narrower than a tile column          -> `any_tile_kernel`
//   * a handful of rows against a whole weight matrix: the T = 1 rollout step (reference models/rnn_base.py single-step branch,
//     smamba/mamba.py:257-305, flash-attn's decode step) - M <= 8 rows, weight read once, launch-latency class -> `rows_nt_kernel`
// Exact fp32 FMAs (at least as accurate as the split editions), optionally with the operands / the result rounded to bf16 the way the
// reference's bf16-autocast projections see them (`rnd` bits), same epilogues as resel_gemm_f32x (bias, ELU, softplus, accumulate),
// same magnitude publication.  Long reductions with few output tiles (narrow weight gradients over all tokens) are cut along K over
// grid.z; the partial tiles are summed in a fixed order by a second kernel (bitwise reproducible, no atomics).
#include "resel_common.h"
#include <algorithm>

namespace {
using namespace resel;

constexpr int RND_A = 1, RND_B = 2, RND_OUT = 4, OUT_BF16 = 8, A_BF16 = 16;

struct AnyParams {
    const void* A;
    const float *B, *bias;
    void* C;
    int64_t a_rs, a_cs, b_rs, b_cs, ldc, sA, sB, sC, sBias;
    int M, N, K, act;            // act: 0 none, 1 ELU, 2 C += product (+ bias), 3 softplus
    int kchunk, nz;              // K elements per grid.z slice; nz > 1: raw partial tiles go to `part` [nz][batch][M][N]
    float* part;
    AmaxOut amax;
};

__device__ __forceinline__ float rbf(float x) {                // round to nearest-even bf16, back as fp32
    typedef __bf16 bf1 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {x, 0.f};
    const bf1 b = __builtin_convertvector(v, bf1);
    return (float)b[0];
}
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : fast_exp(x) - 1.f; }

template <int RND>
__device__ __forceinline__ void store_out(const AnyParams& p, int z, int m, int n, float v, float& cmax) {
    if (p.bias) v += (RND & RND_B) ? rbf(p.bias[(int64_t)z * p.sBias + n]) : p.bias[(int64_t)z * p.sBias + n];
    if (p.act == 1) v = elu1(v);
    if (p.act == 3) v = softplus_nb(v);
    if (RND & OUT_BF16) {
        __bf16* c = reinterpret_cast<__bf16*>(p.C) + (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
        *c = (__bf16)v;
        cmax = fmaxf(cmax, __builtin_fabsf(v));
        return;
    }
    float* c = reinterpret_cast<float*>(p.C) + (int64_t)z * p.sC + (int64_t)m * p.ldc + n;
    if (p.act == 2) v += *c;
    if (RND & RND_OUT) v = rbf(v);
    *c = v;
    cmax = fmaxf(cmax, __builtin_fabsf(v));
}

// 64 x 64 output tile per workgroup, 256 threads x (4 x 4) outputs, 16-k steps through LDS.  grid (tiles_n, tiles_m, batch * nz).
template <int RND>
__global__ __launch_bounds__(256) void any_tile_kernel(AnyParams p) {
    __shared__ float As[16][68], Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int zz = blockIdx.z, z = zz / p.nz, ks = zz - z * p.nz;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* A = reinterpret_cast<const float*>(p.A) + (int64_t)z * p.sA;
    const float* B = p.B + (int64_t)z * p.sB;
    const int kbeg = ks * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const bool a_k = p.a_cs == 1, b_k = p.b_cs == 1;            // which axis of an operand is contiguous: threads run along it
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;
            {
                const int k = a_k ? (idx & 15) : (idx >> 6), m = a_k ? (idx >> 4) : (idx & 63);
                float v = 0.f;
                if (m0 + m < p.M && k0 + k < kend) v = A[(int64_t)(m0 + m) * p.a_rs + (int64_t)(k0 + k) * p.a_cs];
                As[k][m] = (RND & RND_A) ? rbf(v) : v;
            }
            {
                const int k = b_k ? (idx & 15) : (idx >> 6), n = b_k ? (idx >> 4) : (idx & 63);
                float v = 0.f;
                if (n0 + n < p.N && k0 + k < kend) v = B[(int64_t)(n0 + n) * p.b_rs + (int64_t)(k0 + k) * p.b_cs];
                Bs[k][n] = (RND & RND_B) ? rbf(v) : v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 a = ld4(&As[k][ty * 4]), b = ld4(&Bs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float cmax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            if (p.nz > 1) p.part[(((int64_t)ks * (gridDim.z / p.nz) + z) * p.M + m) * p.N + n] = acc[i][j];
            else store_out<RND>(p, z, m, n, acc[i][j], cmax);
        }
    }
    if (p.nz == 1) amax_publish_wave(cmax, p.amax);
}

// C = epilogue(sum over the nz K slices), slices added in index order.  grid (ceil(M N / 256), batch)
template <int RND>
__global__ __launch_bounds__(256) void any_reduce_kernel(AnyParams p, int batch) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int z = blockIdx.y;
    float cmax = 0.f;
    if (e < (int64_t)p.M * p.N) {
        float v = 0.f;
        for (int s = 0; s < p.nz; ++s) v += p.part[((int64_t)s * batch + z) * p.M * p.N + e];
        store_out<RND>(p, z, (int)(e / p.N), (int)(e % p.N), v, cmax);
    }
    amax_publish_wave(cmax, p.amax);
}

// M <= 8 rows x [N, K] weight (both K-contiguous, 16-byte aligned rows): one WAVE per output column - the 64 lanes split K in float4
// pieces, every row of x rides along (x is a few KB: L1 / L2 hits), one shuffle tree per row.  grid (ceil(N / 4), batch), 256 threads.
template <int RND>
__global__ __launch_bounds__(256) void rows_nt_kernel(AnyParams p) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + w, z = blockIdx.y;
    if (n >= p.N) return;                            // whole wave (no barriers in this kernel)
    const float* Brow = p.B + (int64_t)z * p.sB + (int64_t)n * p.b_rs;
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    for (int k = lane * 4; k < p.K; k += 256) {
        float4 wv = ld4(Brow + k);
        if (RND & RND_B) { wv.x = rbf(wv.x); wv.y = rbf(wv.y); wv.z = rbf(wv.z); wv.w = rbf(wv.w); }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m >= p.M) break;
            float4 xv;
            if (RND & A_BF16) {
                const __bf16* a = reinterpret_cast<const __bf16*>(p.A) + (int64_t)z * p.sA + (int64_t)m * p.a_rs + k;
                typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                const bf4 q = *reinterpret_cast<const bf4*>(a);
                xv = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
            } else {
                xv = ld4(reinterpret_cast<const float*>(p.A) + (int64_t)z * p.sA + (int64_t)m * p.a_rs + k);
                if (RND & RND_A) { xv.x = rbf(xv.x); xv.y = rbf(xv.y); xv.z = rbf(xv.z); xv.w = rbf(xv.w); }
            }
            acc[m] = __builtin_fmaf(xv.x, wv.x, acc[m]); acc[m] = __builtin_fmaf(xv.y, wv.y, acc[m]);
            acc[m] = __builtin_fmaf(xv.z, wv.z, acc[m]); acc[m] = __builtin_fmaf(xv.w, wv.w, acc[m]);
        }
    }
    float cmax = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (m >= p.M) break;
        const float s = wave_sum(acc[m]);
        if (lane == 0) store_out<RND>(p, z, m, n, s, cmax);
    }
    amax_publish_wave(cmax, p.amax);
}

inline int pick_nz(int M, int N, int K, int batch) {
    // cut K when the output tiles alone cannot occupy the chip and the reduction is long (narrow weight gradients over all tokens)
    const int64_t tiles = (int64_t)((M + 63) / 64) * ((N + 63) / 64) * batch;
    if (K < 2048 || tiles >= 256) return 1;
    int64_t nz = std::min<int64_t>(512 / tiles, K / 256);
    return (int)std::max<int64_t>(nz, 1);
}

template <int RND>
int launch_any(AnyParams p, int batch, bool rows, hipStream_t s) {
    if (rows) {
        hipLaunchKernelGGL(rows_nt_kernel<RND>, dim3((p.N + 3) / 4, batch), dim3(256), 0, s, p);
        return launch_status();
    }
    hipLaunchKernelGGL(any_tile_kernel<RND>, dim3((p.N + 63) / 64, (p.M + 63) / 64, batch * p.nz), dim3(256), 0, s, p);
    if (p.nz > 1)
        hipLaunchKernelGGL(any_reduce_kernel<RND>, dim3((unsigned)(((int64_t)p.M * p.N + 255) / 256), batch), dim3(256), 0, s, p, batch);
    return launch_status();
}

}  // namespace

namespace resel {

size_t gemm_any_workspace_bytes(int M, int N, int K, int batch) {
    const int nz = pick_nz(M, N, K, batch);
    return nz > 1 ? (size_t)nz * batch * M * N * sizeof(float) : 0;
}

// rows form: M <= 8, both operands K-contiguous with 16-byte (bf16 A: 8-byte) aligned rows, no accumulate
bool gemm_any_rows_ok(const void* A, int64_t lda, int64_t strideA, int a_kcontig, int a_bf16, const float* B, int64_t ldb, int64_t strideB,
                      int b_kcontig, int M, int K, int act) {
    const uintptr_t am = a_bf16 ? 7u : 15u;
    return M <= 8 && a_kcontig && b_kcontig && act != 2 && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0 &&
           !(reinterpret_cast<uintptr_t>(A) & am) && !(reinterpret_cast<uintptr_t>(B) & 15u);
}

// rnd: RND_A | RND_B | RND_OUT | OUT_BF16 | A_BF16 (bf16 A: rows form only)
int gemm_any_launch(const void* A, int64_t lda, int64_t strideA, int a_kcontig, const float* B, int64_t ldb, int64_t strideB, int b_kcontig,
                    const float* bias, int64_t strideBias, int act, void* C, int64_t ldc, int64_t strideC, void* workspace,
                    int M, int N, int K, int batch, int rnd, unsigned long long* amax_c, unsigned amax_epoch, hipStream_t s) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || act < 0 || act > 3 || lda <= 0 || ldb <= 0 || ldc <= 0) return RESEL_EINVAL;
    if ((rnd & OUT_BF16) && act == 2) return RESEL_EINVAL;
    const bool rows = gemm_any_rows_ok(A, lda, strideA, a_kcontig, rnd & A_BF16, B, ldb, strideB, b_kcontig, M, K, act);
    if ((rnd & A_BF16) && !rows) return RESEL_EINVAL;
    AnyParams p{A, B, bias, C, a_kcontig ? lda : 1, a_kcontig ? 1 : lda, b_kcontig ? ldb : 1, b_kcontig ? 1 : ldb, ldc, strideA, strideB, strideC,
                strideBias, M, N, K, act, K, 1, nullptr, AmaxOut{amax_c, amax_epoch}};
    if (!rows) {
        p.nz = pick_nz(M, N, K, batch);
        if (p.nz > 1) {
            if (!workspace) return RESEL_EINVAL;
            p.kchunk = ((K + p.nz - 1) / p.nz + 15) / 16 * 16;
            p.nz = (K + p.kchunk - 1) / p.kchunk;
            p.part = (float*)workspace;
        }
    }
    switch (rnd & 31) {
        case 0: return launch_any<0>(p, batch, rows, s);
        case RND_A | RND_B: return launch_any<RND_A | RND_B>(p, batch, rows, s);
        case RND_A | RND_B | RND_OUT: return launch_any<RND_A | RND_B | RND_OUT>(p, batch, rows, s);
        case RND_A | RND_B | OUT_BF16: return launch_any<RND_A | RND_B | OUT_BF16>(p, batch, rows, s);
        case RND_B | OUT_BF16 | A_BF16: return launch_any<RND_B | OUT_BF16 | A_BF16>(p, batch, rows, s);
        case RND_B | A_BF16: return launch_any<RND_B | A_BF16>(p, batch, rows, s);
        case RND_B | RND_OUT | A_BF16: return launch_any<RND_B | RND_OUT | A_BF16>(p, batch, rows, s);
        default: return RESEL_EINVAL;
    }
}

}  // namespace resel
