// Mixed-precision GEMM for the bf16 attention projections of the cgpt block (reference: flash-attn MHA with fused_bias_fc under
// the trainer's bf16 autocast, models/flash_attention/TransformerFlashAttention.py:67-70): every operand element is ROUNDED to
// bf16 (round to nearest even, as torch's .to(bfloat16)), the products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16, the
// bias (rounded to bf16 like the autocast parameter copy) is added in fp32 and the result leaves as bf16 or fp32.
//
//     C[m][n] = sum_k bf16(A(m, k)) * bf16(B(n, k)) + bf16(bias[n])
//
// Each operand is fp32 or bf16 in memory and [rows][K] or [K][rows] (as in gemm_f32.hip), so that the casts the reference's
// autocast inserts around F.linear - activations fp32 -> bf16, the fp32 master weights -> bf16 every call, bf16 gradients back to
// fp32 - happen on the way into LDS instead of as separate passes over HBM:
//     Wqkv forward      qkv  bf16 = x    fp32 [T][K]  . W fp32 [N][K]        out_proj forward   a    fp32 = ctx bf16 [T][K] . W fp32 [N][K]
//     Wqkv dgrad        dx   fp32 = dqkv bf16 [T][N]  . W fp32 [N][K]^T      out_proj dgrad     dctx bf16 = da  fp32 [T][N] . W fp32 [N][K]^T
//     Wqkv wgrad        dW   fp32 = dqkv bf16 [T][N]^T. x fp32 [T][K]        out_proj wgrad     dW   fp32 = da  fp32 [T][N]^T. ctx bf16 [T][K]
// Structure = csrc/gemm_bf3.hip with ONE plane: 256 x 128 block tile, 8 waves, K step 32, the same swizzled [row][32 k] bf16 LDS
// image, two stages, one barrier per step, persistent blocks, K slices of the tail round / of weight gradients summed by a fix-up
// kernel in a fixed order.  With 8 matrix instructions per wave and K step the kernel runs at the rate its operands arrive.
#include "resel_common.h"
#include <algorithm>

namespace {
using namespace resel;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int BM = 256, BN = 128, BK = 32;
constexpr int NTH = 512;
constexpr int GRID = 256;
constexpr int TILE = BM * BN;
constexpr int ROWB = 64;
constexpr int PLA = BM * ROWB, PLB = BN * ROWB;
constexpr int STAGE = PLA + PLB;                 // 24 576 bytes

struct Params {
    const void *A, *B;
    const float* bias;
    void* C;
    float* slab;
    int64_t lda, ldb, ldc;                       // in elements
    int M, N, K;
    int c_bf16;
    int mt, nt;
    int nfull, nsplit, nsl, kslice;
};

__device__ __forceinline__ void tile_origin(const Params& p, int t, int& m0, int& n0) {
    const int ntile = p.mt * p.nt;
    const int q = ntile / 8, r = ntile % 8, x = t & 7, j = t >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    m0 = (bid / p.nt) * BM;
    n0 = (bid % p.nt) * BN;
}
struct Item { int m0, n0, kbeg, kend, split; };
__device__ __forceinline__ Item decode(const Params& p, int it) {
    Item o;
    int t = it;
    o.kbeg = 0; o.kend = p.K; o.split = 0;
    if (it >= p.nfull) {
        const int idx = it - p.nfull, tr = idx / p.nsl, sl = idx - tr * p.nsl;
        t = p.nfull + tr;
        o.kbeg = sl * p.kslice; o.kend = min(p.K, o.kbeg + p.kslice); o.split = idx + 1;
    }
    tile_origin(p, t, o.m0, o.n0);
    return o;
}
__device__ __forceinline__ int plane_off(int row, int c) {       // as gemm_bf3.hip
    const int q = row >> 2;
    return ((row ^ (q & 1)) << 6) + ((c ^ (q & 3)) << 4);
}
__device__ __forceinline__ uint32_t pack_rne(float a, float b) {  // {bf16(b), bf16(a)}: a in the low half (v_cvt_pk_bf16_f32)
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }

// One operand's share of a thread in a K step (thread -> piece mapping as gemm_bf3.hip).  T = float: pieces are float4; T = __bf16:
// pieces are 8 bytes (four bf16 along the operand's contiguous axis).
template <bool KC, int ROWS, typename T>
struct Src {
    static constexpr bool F32 = sizeof(T) == 4;
    static constexpr int ES = sizeof(T);
    static constexpr int NPC = ROWS / 64;
    static constexpr int NR = KC ? NPC : (ROWS == 256 ? 4 : 2);
    const char* base;
    uint32_t off[NR];
    uint32_t loff;
    int64_t step;
    int kofs;
    float4 r[F32 ? NR : (NR + 1) / 2];             // bf16: two 8-byte pieces per float4 slot
    __device__ __forceinline__ uint2& piece(int i) { return reinterpret_cast<uint2*>(r)[i]; }
    __device__ __forceinline__ const uint2& piece(int i) const { return reinterpret_cast<const uint2*>(r)[i]; }
    __device__ __forceinline__ void init(const void* P0, int64_t ld, int rows, int r0, int k0, int tid) {
        const T* P = reinterpret_cast<const T*>(P0);
        if (KC) {
            kofs = 4 * (tid & 7);
            base = (const char*)(P + (int64_t)r0 * ld + k0);
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int rl = (tid >> 3) + 64 * i;
                off[i] = (uint32_t)(((r0 + rl < rows ? rl : 0) * ld + kofs) * ES);
            }
            step = BK * ES;
        } else {
            const int g = ROWS == 256 ? (tid & 7) + 8 * (tid >> 6) : (tid & 7) + 8 * (tid >> 7);
            kofs = ROWS == 256 ? 4 * ((tid >> 3) & 7) : 2 * ((tid >> 3) & 15);
            base = (const char*)(P + (int64_t)k0 * ld + r0);
            const int rl = 4 * g;
#pragma unroll
            for (int j = 0; j < NR; ++j) off[j] = (uint32_t)(((kofs + j) * ld + (r0 + rl < rows ? rl : 0)) * ES);
            step = (int64_t)BK * ld * ES;
        }
    }
    __device__ __forceinline__ void init_lds(int tid) {
        if (KC) {
            loff = plane_off(tid >> 3, (tid & 7) >> 1) + 8 * (tid & 1);
        } else if (ROWS == 256) {
            const int g = (tid & 7) + 8 * (tid >> 6), k4 = (tid >> 3) & 7;
            loff = plane_off(4 * g, k4 >> 1) + 8 * (k4 & 1);
        } else {
            const int g = (tid & 7) + 8 * (tid >> 7), k2 = (tid >> 3) & 15;
            loff = plane_off(4 * g, k2 >> 2) + 4 * (k2 & 3);
        }
    }
    __device__ __forceinline__ void load(int k0, int kend) {
        const bool full = k0 + BK <= kend;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const bool on = full || (k0 + kofs + (KC ? 0 : i) < kend);
            if (F32) {
                r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on) r[i] = *reinterpret_cast<const float4*>(base + off[i]);
            } else {
                piece(i) = make_uint2(0u, 0u);
                if (on) piece(i) = *reinterpret_cast<const uint2*>(base + off[i]);
            }
        }
        base += step;
    }
    // the four values of piece i along its contiguous axis as two packed bf16 words
    __device__ __forceinline__ uint2 packed(int i) const {
        if (F32) return make_uint2(pack_rne(r[i].x, r[i].y), pack_rne(r[i].z, r[i].w));
        return piece(i);
    }
    __device__ __forceinline__ void store(char* pl) const {
        if (KC) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) *reinterpret_cast<uint2*>(pl + loff + 4096 * i) = packed(i);
        } else if (ROWS == 256) {                  // 4 (k) x 4 (rows) patch -> row j gets its four k
            const uint2 q0 = packed(0), q1 = packed(1), q2 = packed(2), q3 = packed(3);   // q_k = {rows 0,1 | rows 2,3} of k row k
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;                // high / low halves
                const uint32_t w0 = (j < 2) ? __builtin_amdgcn_perm(q1.x, q0.x, sel) : __builtin_amdgcn_perm(q1.y, q0.y, sel);
                const uint32_t w1 = (j < 2) ? __builtin_amdgcn_perm(q3.x, q2.x, sel) : __builtin_amdgcn_perm(q3.y, q2.y, sel);
                *reinterpret_cast<uint2*>(pl + (loff ^ (j << 6))) = make_uint2(w0, w1);
            }
        } else {                                   // 2 (k) x 4 (rows) patch
            const uint2 q0 = packed(0), q1 = packed(1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
                const uint32_t w0 = (j < 2) ? __builtin_amdgcn_perm(q1.x, q0.x, sel) : __builtin_amdgcn_perm(q1.y, q0.y, sel);
                *reinterpret_cast<uint32_t*>(pl + (loff ^ (j << 6))) = w0;
            }
        }
    }
};

__device__ __forceinline__ bf16x8 lds16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <typename TA, typename TB, bool AKC, bool BKC>
__global__ __launch_bounds__(NTH, 2) void gemm_bf16_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const int total = p.nfull + p.nsplit * p.nsl;
    const int G = gridDim.x;
    if ((int)blockIdx.x >= total) return;

    Src<AKC, BM, TA> sa;
    Src<BKC, BN, TB> sb;
    sa.init_lds(tid);
    sb.init_lds(tid);
    int p_item = blockIdx.x, p_k0, p_kend;
    bool p_live = true;
    auto p_open = [&]() {
        const Item it = decode(p, p_item);
        sa.init(p.A, p.lda, p.M, it.m0, it.kbeg, tid);
        sb.init(p.B, p.ldb, p.N, it.n0, it.kbeg, tid);
        p_k0 = it.kbeg; p_kend = it.kend;
    };
    auto produce = [&]() {
        if (!p_live) return;
        sa.load(p_k0, p_kend);
        sb.load(p_k0, p_kend);
        p_k0 += BK;
        if (p_k0 >= p_kend) {
            p_item += G;
            if (p_item < total) p_open(); else p_live = false;
        }
    };
    const char* fa[2];
    const char* fb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fa[s] = lds + wm * ROWB + plane_off(li, 2 * s + lh);
        fb[s] = lds + PLA + wn * ROWB + plane_off(li, 2 * s + lh);
    }
    p_open();
    produce();
    sa.store(lds);
    sb.store(lds + PLA);
    produce();
    __syncthreads();
    int cur_st = 0;
    for (int c_item = blockIdx.x; c_item < total; c_item += G) {
        const Item cur = decode(p, c_item);
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        for (int c_k0 = cur.kbeg; c_k0 < cur.kend; c_k0 += BK) {
            const int so = cur_st * STAGE, sn = (cur_st ^ 1) * STAGE;
            bf16x8 fA[2][2], fB[2][2];                               // [slab][tile]
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fA[s][t] = lds16(fa[s] + so + t * 32 * ROWB);
                    fB[s][t] = lds16(fb[s] + so + t * 32 * ROWB);
                }
            sa.store(lds + sn);                                       // step s + 1 (zeros past the end of everything)
            sb.store(lds + sn + PLA);
            produce();                                                // global loads of step s + 2
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA[s][a], fB[s][b], acc[a][b], 0, 0, 0);
            __syncthreads();
            cur_st ^= 1;
        }
        // epilogue: D layout col = lane & 31 (n), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (m)
        if (cur.split) {
            float* o = p.slab + (int64_t)(cur.split - 1) * TILE + (wm + 4 * lh) * BN + wn + li;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[(32 * a + (e & 3) + 8 * (e >> 2)) * BN + 32 * b] = acc[a][b][e];
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = cur.n0 + wn + 32 * b + li;
                if (n >= p.N) continue;
                const float bv = p.bias ? bf16_round(p.bias[n]) : 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int mb = cur.m0 + wm + 32 * a + 4 * lh;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = mb + (e & 3) + 8 * (e >> 2);
                        if (m < p.M) {
                            const float v = acc[a][b][e] + bv;
                            if (p.c_bf16 == 1) reinterpret_cast<__bf16*>(p.C)[(int64_t)m * p.ldc + n] = (__bf16)v;
                            else reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n] = p.c_bf16 ? bf16_round(v) : v;
                        }
                    }
                }
            }
        }
    }
}

// C tile = sum over the K slices of a split tile (+ bias), fixed order; grid (TILE / 4 / 64, split tiles), block (64, 4)
__global__ __launch_bounds__(256) void gemm_bf16_fixup_kernel(Params p) {
    __shared__ float4 part[3][64];
    const int tr = blockIdx.y, q = threadIdx.y;
    const int e = blockIdx.x * 64 + threadIdx.x, ml = e >> 5, nl = 4 * (e & 31);
    const float* s = p.slab + (int64_t)tr * p.nsl * TILE + ml * BN + nl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = q; i < p.nsl; i += 4) {
        const float4 u = ld4(s + (int64_t)i * TILE);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (q) part[q - 1][threadIdx.x] = v;
    __syncthreads();
    if (q) return;
    const float4 g1 = part[0][threadIdx.x], g2 = part[1][threadIdx.x], g3 = part[2][threadIdx.x];
    const float o[4] = {(v.x + g1.x) + (g2.x + g3.x), (v.y + g1.y) + (g2.y + g3.y), (v.z + g1.z) + (g2.z + g3.z), (v.w + g1.w) + (g2.w + g3.w)};
    int m0, n0;
    tile_origin(p, p.nfull + tr, m0, n0);
    const int m = m0 + ml, n = n0 + nl;
    if (m >= p.M) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (n + j >= p.N) break;
        const float x = o[j] + (p.bias ? bf16_round(p.bias[n + j]) : 0.f);
        if (p.c_bf16 == 1) reinterpret_cast<__bf16*>(p.C)[(int64_t)m * p.ldc + n + j] = (__bf16)x;
        else reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n + j] = p.c_bf16 ? bf16_round(x) : x;
    }
}

struct Plan { int nfull, nsplit, nsl, kslice; };
inline Plan make_plan(int M, int N, int K) {
    const long nbt = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int ksteps = (K + BK - 1) / BK;
    Plan pl{(int)nbt, 0, 1, ksteps * BK};
    const int r = (int)(nbt % GRID);
    if (r == 0 || r > GRID / 2 || ksteps < 4) return pl;
    int s = std::min(GRID / r, ksteps / 2);
    const int per = (ksteps + s - 1) / s;
    s = (ksteps + per - 1) / per;
    if (s < 2) return pl;
    pl.nfull = (int)(nbt - r); pl.nsplit = r; pl.nsl = s; pl.kslice = per * BK;
    return pl;
}

template <typename TA, typename TB>
void launch_layout(const Params& p, dim3 grid, int akc, int bkc, hipStream_t s) {
    if (akc && bkc) hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true, true>), grid, dim3(NTH), 0, s, p);
    else if (akc) hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, true, false>), grid, dim3(NTH), 0, s, p);
    else if (bkc) hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, false, true>), grid, dim3(NTH), 0, s, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, false, false>), grid, dim3(NTH), 0, s, p);
}

}  // namespace

extern "C" size_t resel_gemm_bf16_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const Plan pl = make_plan(M, N, K);
    return std::max((size_t)pl.nsplit * pl.nsl * TILE * sizeof(float), gemm_any_workspace_bytes(M, N, K, 1));
}

extern "C" int resel_gemm_bf16(const void* A, int64_t lda, int a_kcontig, int a_bf16, const void* B, int64_t ldb, int b_kcontig, int b_bf16,
                               const float* bias, void* C, int64_t ldc, int c_bf16, void* workspace, int M, int N, int K,
                               resel_stream_t stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc <= 0) return RESEL_EINVAL;
    // a thread fetches four consecutive elements of an operand's contiguous axis: extents, leading dimensions and base addresses
    // must keep those 16-byte (fp32) / 8-byte (bf16) pieces aligned.  Shapes that do not, and the M <= 8 rows of a decode step against
    // a whole weight matrix, go to gemm_any.hip with the same rounding points (operands and bias rounded to bf16, fp32 accumulation,
    // result rounded as `c_bf16` says) - fp32 operands only there, except a bf16 A in the rows form (the attention output of a decode step)
    const bool mfma_ok = !((a_kcontig ? K : M) % 4 || (b_kcontig ? K : N) % 4 || lda % 4 || ldb % 4 ||
                           (reinterpret_cast<uintptr_t>(A) & (a_bf16 ? 7u : 15u)) || (reinterpret_cast<uintptr_t>(B) & (b_bf16 ? 7u : 15u)) ||
                           lda >= (int64_t)1 << 22 || ldb >= (int64_t)1 << 22);
    const bool rows = !b_bf16 && gemm_any_rows_ok(A, lda, 0, a_kcontig, a_bf16, (const float*)B, ldb, 0, b_kcontig, M, K, 0);
    if (!mfma_ok || rows) {
        if (b_bf16 || (a_bf16 && !rows)) return RESEL_EINVAL;
        const int rnd = (a_bf16 ? 16 : 1) | 2 | (c_bf16 == 1 ? 8 : (c_bf16 == 2 ? 4 : 0));
        return gemm_any_launch(A, lda, 0, a_kcontig, (const float*)B, ldb, 0, b_kcontig, bias, 0, 0, C, ldc, 0, workspace, M, N, K, 1, rnd, nullptr, 0u,
                               (hipStream_t)stream);
    }
    const Plan pl = make_plan(M, N, K);
    if (pl.nsplit && (!workspace || !aligned16(workspace))) return RESEL_EINVAL;
    Params p{A, B, bias, C, (float*)workspace, lda, ldb, ldc, M, N, K, c_bf16 == 2 ? 2 : (c_bf16 ? 1 : 0),
             (M + BM - 1) / BM, (N + BN - 1) / BN, pl.nfull, pl.nsplit, pl.nsl, pl.kslice};
    const int64_t total = (int64_t)pl.nfull + (int64_t)pl.nsplit * pl.nsl;
    dim3 grid((unsigned)std::min<int64_t>(total, GRID));
    hipStream_t s = (hipStream_t)stream;
    if (a_bf16 && b_bf16) launch_layout<__bf16, __bf16>(p, grid, a_kcontig, b_kcontig, s);
    else if (a_bf16) launch_layout<__bf16, float>(p, grid, a_kcontig, b_kcontig, s);
    else if (b_bf16) launch_layout<float, __bf16>(p, grid, a_kcontig, b_kcontig, s);
    else launch_layout<float, float>(p, grid, a_kcontig, b_kcontig, s);
    if (pl.nsplit) hipLaunchKernelGGL(gemm_bf16_fixup_kernel, dim3(TILE / 4 / 64, pl.nsplit), dim3(64, 4), 0, s, p);
    return launch_status();
}

// ---- bias gradient of a bf16 projection: out[n] = sum_m gy[m][n] in fp32 (the `.sum(0)` of LinearBf16's backward; ATen's reduction
// takes 29 us for [32 832, 256..1024] bf16, 10 x the time of the bytes).  A thread owns 8 consecutive columns (one 16-byte load per
// row), a block a slab of rows; per-block partials are summed in a fixed order by colsum_kernel (no atomics, bitwise reproducible).
namespace {
constexpr int CS_ROWS = 64;                        // rows per block (256: 129 blocks at configs[2], half the chip, 42 us for 50 MB; 64: 513 blocks)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const uint16_t* __restrict__ x, int64_t ld, int M, int N, float* __restrict__ part) {
    __shared__ float s_acc[256][9];
    const int tpr = N >> 3;                        // threads per row
    const int rpi = 256 / tpr;                     // rows per iteration
    const int tid = threadIdx.x, cg = tid % tpr, rl = tid / tpr;
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < rpi) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += rpi) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + (int64_t)r * ld + 8 * cg);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[2 * j] += __uint_as_float(w[j] << 16);
                a[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s_acc[tid][j] = a[j];
    __syncthreads();
    if (tid < tpr) {                               // row group 0 of every column group sums the others in a fixed order
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = 0.f;
            for (int g = 0; g < rpi; ++g) t += s_acc[g * tpr + tid][j];
            part[(int64_t)blockIdx.x * N + 8 * tid + j] = t;
        }
    }
}
}  // namespace

extern "C" size_t resel_colsum_bf16_workspace_bytes(int M, int N) {
    return M > 0 && N > 0 ? (size_t)((M + CS_ROWS - 1) / CS_ROWS) * N * sizeof(float) : 0;
}

extern "C" int resel_colsum_bf16(const uint16_t* x, int64_t ld, int M, int N, float* out, void* workspace, resel_stream_t stream) {
    if (!x || !out || !workspace || M <= 0 || N <= 0 || N % 8 || N > 2048 || ld % 8 || ((uintptr_t)x & 15)) return RESEL_EINVAL;
    const int nblk = (M + CS_ROWS - 1) / CS_ROWS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(nblk), dim3(256), 0, s, x, ld, M, N, (float*)workspace);
    resel::launch_colsum((const float*)workspace, N, nblk, N, out, s);
    return resel::launch_status();
}
